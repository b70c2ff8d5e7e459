"""Is the captured hot-path step GPU-bound or bound by the host cost of hipGraphLaunch?
Prints wall ms/step over many replays, the host time of one replay() call, and the device time of
a single isolated replay (HIP events on the step's stream)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    a = argparse.Namespace(workload="baby", d=64, gcn_layers=3, batch=1024)
    dev = torch.device("cuda", 0)
    step, raw, mats, plans = bench.build_single_gpu(a, dev)
    b = bench.make_batches(raw, 1, a.batch, seed=2022)[0]
    step.set_batch(torch.stack([torch.from_numpy(x).to(dev) for x in b]))
    assert step.capture()
    g, s = step._graph, step.stream
    for _ in range(20):
        step.run()
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        step.run()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("wall %.4f ms/step; host enqueue only %.4f ms/step" % (t_all * 1e3 / n, t_host * 1e3 / n))
    dts = []
    with torch.cuda.stream(s):
        for _ in range(20):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            g.replay()
            e1.record(s)
            torch.cuda.synchronize()
            dts.append(e0.elapsed_time(e1))
    dts.sort()
    print("isolated replay, device time: min %.4f median %.4f ms" % (dts[0], dts[len(dts) // 2]))
    with torch.cuda.stream(s):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(50):
            g.replay()
        e1.record(s)
        torch.cuda.synchronize()
        print("50 back-to-back replays, device time per replay: %.4f ms" % (e0.elapsed_time(e1) / 50))


    print("graph nodes: see DEBUG_HIP_GRAPH_DOT_PRINT; env:", {k: v for k, v in os.environ.items() if "GRAPH" in k or k.startswith("MMSSL_")})


if __name__ == "__main__":
    main()
