"""Is the projection GEMM's time DATA-dependent? Same kernels, same shapes, same memory traffic and instruction stream;
only the operand VALUES change (zeros toggle far fewer bits in the fp32 multipliers and on the data paths = less power).
A kernel limited by issue, latency or bandwidth takes the same time for any data; one held back by the chip's power
management runs faster on zeros. rocm-smi power / sclk are sampled while each variant loops."""
import json, os, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmssl_amd import ops, _lib
SMI = "/opt/rocm/bin/rocm-smi"
M, K, d = 18357, 4096, 64


def smi():
    try:
        d0 = list(json.loads(subprocess.run([SMI, "--showpower", "--showclocks", "--json"], capture_output=True, text=True).stdout).values())[0]
    except Exception:
        return {}
    return {k.split("(")[0].strip()[:22]: v for k, v in d0.items() if "sclk clock speed" in k.lower() or "power" in k.lower()}


def run(name, fn, secs=2.0):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    samples, stop = [], [False]

    def sampler():
        time.sleep(0.5)
        while not stop[0]:
            samples.append(smi())
            time.sleep(0.3)
    th = threading.Thread(target=sampler); th.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n, t0 = 0, time.time()
    e0.record()
    while time.time() - t0 < secs:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize(); stop[0] = True; th.join()
    print("%-44s %7.1f us   %s" % (name, e0.elapsed_time(e1) * 1e3 / n, samples[-1] if samples else ""), flush=True)


gen = torch.Generator(device="cuda").manual_seed(0)
W = torch.randn(d, K, device="cuda", generator=gen) * 0.02
b = torch.zeros(d, device="cuda")
variants = {
    "F random, W random": (torch.randn(M, K, device="cuda", generator=gen), W),
    "F zeros,  W zeros": (torch.zeros(M, K, device="cuda"), torch.zeros_like(W)),
    "F random, W zeros": (torch.randn(M, K, device="cuda", generator=gen), torch.zeros_like(W)),
    "F ones,   W ones": (torch.ones(M, K, device="cuda"), torch.ones_like(W)),
    "F small ints (exact), W random": (torch.randint(-3, 4, (M, K), device="cuda", generator=gen).float(), W),
}
with torch.no_grad():
    for name, (F_, Wv) in variants.items():
        run("fwd  " + name, lambda: ops._linear_raw(F_, Wv, b, None, 1.0))
    gY = torch.randn(M, d, device="cuda", generator=gen)
    for name, (F_, Wv) in list(variants.items())[:2]:
        g = gY if "random" in name else torch.zeros_like(gY)
        run("wgrad " + name, lambda: ops._linear_wgrad_raw(g, None, 1.0, F_, Wv))
    big = torch.randn(64 << 20, device="cuda"); big2 = torch.empty_like(big)
    run("copy 256 MiB random", lambda: big2.copy_(big))
    big.zero_()
    run("copy 256 MiB zeros", lambda: big2.copy_(big))
    x = torch.randn(4096, 4096, device="cuda")
    run("rocBLAS sgemm 4096^3 random", lambda: torch.mm(x, x))
    x.zero_()
    run("rocBLAS sgemm 4096^3 zeros", lambda: torch.mm(x, x))
