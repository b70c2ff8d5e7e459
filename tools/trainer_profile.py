"""cProfile of a few Trainer batches (host-side hot spots of the whole loop). Same setup as trainer_bench.py."""
import cProfile
import os
import pstats
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import synth_data  # noqa: E402
from mmssl_amd import synth  # noqa: E402
from mmssl_amd.config import configure  # noqa: E402
from mmssl_amd.utility import batch_test  # noqa: E402
from mmssl_amd import main as M  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "baby"
U, I, E, dv, dt = synth.SHAPES[wl]
root = tempfile.mkdtemp(prefix="mmssl_prof_")
synth_data.write_dataset(root, wl, U, I, E, dv, dt, seed=1)
configure(["--data_path", root + "/", "--dataset", wl, "--weight_size", "[64,64,64]", "--verbose", "0"])
M.set_seed(2022)
dg = batch_test.init_data()
tr = M.Trainer({"n_users": dg.n_users, "n_items": dg.n_items})


def batches(n, start):
    for idx in range(start, start + n):
        tr.model.train()
        users, pos, neg = dg.sample()
        tr._discriminator_step(users)
        tr._generator_step(idx, users, pos, neg)
    torch.cuda.synchronize()


batches(3, 0)
pr = cProfile.Profile()
pr.enable()
batches(6, 3)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
