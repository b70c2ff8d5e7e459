"""Where does a replayed step's time go? Host enqueue time of N replays vs their total time, for the step as benched
(streams forked) and on one stream, split-precision and fp32-MFMA projection. python tools/step_probe.py"""
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def build(proj, overlap):
    from mmssl_amd import ops
    from mmssl_amd.hotpath import HotPathStep
    ops.PROJ_SPLIT = proj == "split"
    a = types.SimpleNamespace(workload="baby", d=64, gcn_layers=3, batch=1024, graph="uniform", xcd_bands=0, no_fuse_adam=False,
                              dense_fuse=False)
    step, raw, mats, plans = bench.build_single_gpu(a, torch.device("cuda"))
    if not overlap:
        step = HotPathStep(step.model, step.graphs, a.batch, decay=1e-5, overlap=False)
    batches = [torch.stack([torch.from_numpy(x).cuda() for x in b]) for b in bench.make_batches(raw, 8, a.batch, seed=2022)]
    step.set_batch_ring(torch.stack(batches))
    assert step.capture()
    return step


def main():
    out = []
    for proj in ("split", "f32"):
        for overlap in (True, False):
            step = build(proj, overlap)
            for _ in range(100):
                step.run()
            torch.cuda.synchronize()
            n = 500
            t0 = time.perf_counter()
            for _ in range(n):
                step.run()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            rec = {"proj": proj, "overlap": overlap, "host_enqueue_us_per_step": round((t1 - t0) / n * 1e6, 1),
                   "total_us_per_step": round((t2 - t0) / n * 1e6, 1)}
            # one replay at a time (no queueing behind the previous one): replay latency
            ts = []
            for _ in range(50):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                step.run()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            rec["single_replay_us_median"] = round(sorted(ts)[len(ts) // 2] * 1e6, 1)
            print(json.dumps(rec), flush=True)
            out.append(rec)
            del step
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
