"""Import the upstream MMSSL reference in THIS container (CPU only) — TEST INFRASTRUCTURE.

Used only by oracle/gen_golden.py and by the optional `-m "not gpu"` cross-check tests
that skip when /root/reference is absent (it never exists on the GPU box). Nothing of
the reference travels: only the .npz vectors written by gen_golden.py are committed.

Shims (none touches reference files):
  1. sys.argv is set before import: parse_args() runs at import in 4 modules
     (reference Models.py:15, main.py:34, utility/load_data.py:8, utility/batch_test.py:13).
  2. `.cuda()` becomes identity (no GPU here; hard-coded at Models.py:46-47,123 and
     main.py:59-60,71-72,112,...).
  3. dgl / visdom / torch.utils.tensorboard: empty stub modules (dead-code imports,
     main.py:8,13,31).
  4. numpy 2.x removed np.asfarray (utility/metrics.py:50,75).
"""
import importlib
import os
import sys
import types

REF_ROOT = "/root/reference/MMSSL"


def available():
    return os.path.isdir(REF_ROOT)


def load(dataset_dir_parent, dataset, extra_argv=()):
    """Returns the imported reference `main` module (which re-exports Trainer, MMSSL,
    data_generator, test_torch, ...). `dataset_dir_parent` must end with '/'."""
    import numpy as np
    import torch

    assert dataset_dir_parent.endswith("/")
    for m in [k for k in sys.modules if k == "main" or k == "Models" or k.startswith("utility")]:
        del sys.modules[m]
    sys.argv = ["main.py", "--dataset", dataset, "--data_path", dataset_dir_parent, "--debug"] + list(extra_argv)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    for name in ("dgl", "visdom"):
        sys.modules.setdefault(name, types.ModuleType(name))
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = object
    sys.modules.setdefault("torch.utils.tensorboard", tb)
    if not hasattr(np, "asfarray"):
        np.asfarray = lambda a, dtype=float: np.asarray(a, dtype=dtype)
    import multiprocessing
    # batch_test.py:11 uses cpu_count()//5 workers; Pool(0) raises on <5-core hosts.
    if multiprocessing.cpu_count() < 5:
        multiprocessing.cpu_count = lambda: 5
    return importlib.import_module("main")
