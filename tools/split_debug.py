"""Which part of SplitHotPath's forward-only capture upsets hipStreamEndCapture? Each variant runs in a child process."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, tempfile, faulthandler
faulthandler.enable()
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests")); sys.path.insert(0, os.path.join(%r, "oracle"))
import torch
import helpers as H
from mmssl_amd import config, ops
from mmssl_amd.utility import batch_test
variant = sys.argv[1]
tmp = tempfile.mkdtemp()
root = H.write_dataset_dir(tmp)
config.configure([], data_path=root, dataset="tiny", batch_size=48, drop_rate=0.2, debug=True)
batch_test.init_data()
from mmssl_amd.main import Trainer, set_seed
set_seed(2022)
tr = Trainer(data_config={})
dg = tr.data_generator
os.environ["MMSSL_TRAINER_GRAPH"] = "0"
for idx in range(int(os.environ.get("EAGER_BATCHES", "3"))):
    tr.model.train(); u, p, n = dg.sample(); tr.train_batch(idx, u, p, n)
torch.cuda.synchronize()
assert tr._steady_state()
from mmssl_amd.hotpath import SplitHotPath
from mmssl_amd.config import args
cap = SplitHotPath(tr.model, tr._graphs(), tr.optimizer_D, 48, tr.decay, [1, 1, 1, args.cl_rate, args.cl_rate], 1e-7)
s = cap.stream
m = tr.model
def fwd_model_only():
    return m(*cap.graphs)
def fwd_full():
    return cap._forward()
if variant.startswith("bis"):
    kind = variant.split("-")[1]
    cap.extra_grads = None
    with torch.cuda.stream(s):
        for _ in range(2):
            outs, terms, total = cap._forward()
            if cap.extra_grads is None:
                cap.extra_grads = [torch.zeros_like(outs[k]) for k in (2, 3, 4, 5)]
            if kind == "noopt":
                cap.optimizer.zero_grad(set_to_none=True)
                torch.autograd.backward([total] + [outs[k] for k in (2, 3, 4, 5)], [cap._one] + list(cap.extra_grads))
            elif kind == "optonly":
                cap.optimizer.step()
            elif kind == "oneroot":
                cap.optimizer.zero_grad(set_to_none=True)
                total.backward(gradient=cap._one)
                cap.optimizer.step()
            elif kind == "keepgrads":
                cap.optimizer.zero_grad(set_to_none=False)
                torch.autograd.backward([total] + [outs[k] for k in (2, 3, 4, 5)], [cap._one] + list(cap.extra_grads))
                cap.optimizer.step()
            elif kind == "full":
                cap._backward(outs, total)
    torch.cuda.synchronize()
    del outs, terms, total
    print("warm-up done", kind, flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        o = cap._forward()
    torch.cuda.synchronize()
    print("OK", variant, flush=True)
    sys.exit(0)
if variant.startswith("cap"):
    wu = int(variant.split("-")[1])
    print("cap.capture warmup", wu, flush=True)
    ok = cap.capture(warmup=wu)
    print("capture returned", ok, getattr(cap, "capture_error", None), flush=True)
    if ok:
        import numpy as np
        b3 = torch.stack([torch.from_numpy(np.asarray(x, dtype=np.int64)).cuda() for x in dg.sample()])
        cap.forward(b3); cap.backward(None); torch.cuda.synchronize()
        print("OK", variant, float(cap.loss), flush=True)
    sys.exit(0)
fn = {"model": fwd_model_only, "full": fwd_full}[variant.split("-")[0]]
with torch.cuda.stream(s):
    for _ in range(2):
        o = fn()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
mode = "thread_local" if "tl" in variant else "global"
print("capturing", variant, flush=True)
with torch.cuda.graph(g, stream=s, capture_error_mode=mode):
    o = fn()
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print("OK", variant, flush=True)
''' % (ROOT, ROOT, ROOT)
for env, variant in (({"EAGER_BATCHES": "3"}, "cap-2"), ({"EAGER_BATCHES": "2"}, "cap-2")):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", CHILD, variant], env=e, capture_output=True, text=True, timeout=300)
    tail = [l for l in (r.stdout + r.stderr).splitlines() if l.strip() and "amdgpu.ids" not in l][-6:]
    print("=== %s %s -> rc %d" % (variant, env, r.returncode))
    for l in tail:
        print("    ", l[:200])
