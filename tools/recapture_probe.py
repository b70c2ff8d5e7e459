"""Is the step time a property of the process (memory layout, clocks) or of the captured graph (how the hipGraph executor
mapped branches to queues)? One process, the Baby hot-path step captured several times, each capture timed."""
import os, sys, time, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

a = types.SimpleNamespace(workload="baby", d=64, gcn_layers=3, batch=1024)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
step, raw, mats, plans = bench.build_single_gpu(a, dev)
batches = [(torch.stack([torch.from_numpy(x).to(dev) for x in b]),) for b in bench.make_batches(raw, 8, a.batch, seed=2022)]


def run(n):
    for i in range(n):
        step.set_batch(*batches[i % len(batches)])
        step.run()


for rep in range(int(os.environ.get("REPS", "6"))):
    ok = step.capture()
    run(300)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        run(500)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 500 * 1e3)
    print("capture %d ok=%s: %s ms/step" % (rep, ok, " ".join("%.4f" % t for t in ts)), flush=True)
