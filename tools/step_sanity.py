"""Loss trajectories of the captured Baby step under the step object's options: ring vs set_batch, fused AdamW vs one
launch. All variants start from the same parameters, dropout generator state and batches: they must agree to rounding."""
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mmssl_amd import ops  # noqa: E402

a = types.SimpleNamespace(workload=sys.argv[1] if len(sys.argv) > 1 else "baby", d=64, gcn_layers=3, batch=1024, no_fuse_adam=False)
dev = torch.device("cuda", 0)
out = {}
for name, ring, fuse in (("ring+fused", True, True), ("set_batch+fused", False, True), ("ring+plain", True, False),
                         ("set_batch+plain", False, False)):
    a.no_fuse_adam = not fuse
    torch.manual_seed(2022)
    step, raw, mats, plans = bench.build_single_gpu(a, dev)
    ops.seed_dropout(7, dev)
    batches = [torch.stack([torch.from_numpy(x).to(dev) for x in b]) for b in bench.make_batches(raw, 8, a.batch, seed=2022)]
    step.set_batch(batches[0])
    if ring:
        step.set_batch_ring(torch.stack(batches))
    assert step.capture(warmup=2), getattr(step, "capture_error", "")
    losses = []
    n0 = int(step.optimizer.step_counter(0, dev)[0])
    for i in range(120):
        if not ring:
            step.set_batch(batches[(n0 + i) % 8])
        step.run()
        if i % 10 == 9:
            torch.cuda.synchronize()
            losses.append(round(float(step.loss), 5))
    out[name] = losses
    del step
print(json.dumps(out, indent=1))
