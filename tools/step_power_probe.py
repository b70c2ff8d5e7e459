"""Is the whole hot-path step held back by the board's power management? The captured Baby step is replayed for a few
seconds while rocm-smi power / sclk are sampled, once with the real (random) feature matrices and once with the feature
matrices zeroed: same kernels, same launches, same bytes moved, far fewer toggling bits in the two projection GEMMs."""
import json, os, subprocess, sys, threading, time, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
SMI = "/opt/rocm/bin/rocm-smi"


def smi():
    try:
        d0 = list(json.loads(subprocess.run([SMI, "--showpower", "--showclocks", "--json"], capture_output=True, text=True).stdout).values())[0]
    except Exception:
        return None
    out = {}
    for k, v in d0.items():
        kl = k.lower()
        if "sclk clock speed" in kl:
            out["sclk"] = float(str(v).strip("()Mhz ").replace("Mhz", ""))
        elif "power" in kl:
            try:
                out["power"] = float(v)
            except Exception:
                pass
    return out


a = types.SimpleNamespace(workload="baby", d=64, gcn_layers=3, batch=1024)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
step, raw, mats, plans = bench.build_single_gpu(a, dev)
batches = [(torch.stack([torch.from_numpy(x).to(dev) for x in b]),) for b in bench.make_batches(raw, 8, a.batch, seed=2022)]
assert step.capture()


def run(n):
    for i in range(n):
        step.set_batch(*batches[i % len(batches)])
        step.run()


def measure(tag, secs=4.0):
    run(300)
    torch.cuda.synchronize()
    samples, stop = [], [False]

    def sampler():
        time.sleep(0.4)
        while not stop[0]:
            s = smi()
            if s:
                samples.append(s)
            time.sleep(0.25)
    th = threading.Thread(target=sampler); th.start()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < secs:
        run(200)
        n += 200
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stop[0] = True; th.join()
    pw = [s["power"] for s in samples if "power" in s]
    ck = [s["sclk"] for s in samples if "sclk" in s]
    print("%-34s %.4f ms/step | power avg %.0f W max %.0f W | sclk avg %.0f MHz min %.0f (%d samples)" % (
        tag, dt / n * 1e3, sum(pw) / max(len(pw), 1), max(pw or [0]), sum(ck) / max(len(ck), 1), min(ck or [0]), len(samples)), flush=True)


m = step.model
measure("features random (as benchmarked)")
with torch.no_grad():
    img0, txt0 = m.image_feats.clone(), m.text_feats.clone()
    m.image_feats.zero_(); m.text_feats.zero_()
measure("features zeroed")
with torch.no_grad():
    m.image_feats.copy_(img0); m.text_feats.copy_(txt0)
measure("features random again")
