"""Stage-wise hipGraph capture probe for the hot-path step (each stage in its own process)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

STAGES = ["fwd", "fwd_loss", "fwd_bwd", "full"]


def run(stage):
    import numpy as np, torch, scipy.sparse as sp
    from mmssl_amd import config, synth, ops
    from mmssl_amd.graph import GraphPlan
    from mmssl_amd.hotpath import HotPathStep
    from mmssl_amd.Models import MMSSL
    U, I, E, dv, dt = synth.SHAPES["tiktok"]
    config.configure([], weight_size="[64, 64, 64]", batch_size=1024, drop_rate=0.2)
    raw = synth.interaction_matrix(U, I, E)
    ui, iu = synth.normalised_pair(raw)
    P = (GraphPlan(ui), GraphPlan(iu))
    e1, e2 = GraphPlan(sp.csr_matrix((U, I), dtype=np.float32)), GraphPlan(sp.csr_matrix((I, U), dtype=np.float32))
    graphs = (P[0], P[1], e1, e2, e1, e2)
    model = MMSSL(U, I, 64, [64] * 3, [0.1] * 3, np.random.randn(I, dv).astype(np.float32),
                  np.random.randn(I, dt).astype(np.float32)).cuda().train()
    step = HotPathStep(model, graphs, 1024)
    step.set_batch(torch.randperm(U)[:1024].cuda(), torch.randint(0, I, (1024,)).cuda(), torch.randint(0, I, (1024,)).cuda())

    def body():
        if stage == "fwd":
            with torch.no_grad():
                return model(*graphs)[0].sum()
        if stage == "fwd_loss":
            with torch.no_grad():
                return step.losses()[0]
        if stage == "fwd_bwd":
            for p in model.parameters():
                p.grad = None
            t, _ = step.losses()
            t.backward()
            return t.detach()
        return step.step()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            body()
    torch.cuda.synchronize()
    print(stage, "warmup ok", flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        out = body()
    torch.cuda.synchronize()
    print(stage, "capture ok", flush=True)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print(stage, "replay ok", float(out), flush=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(stage, "graph replay: %.1f us/iter" % (e0.elapsed_time(e1) * 1e3 / 50), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for st in STAGES:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), st], capture_output=True, text=True)
            print("== stage %s rc=%d" % (st, r.returncode))
            print(r.stdout[-1500:])
            err = [l for l in r.stderr.splitlines() if "Warning" not in l and "amdgpu.ids" not in l]
            print("\n".join(err[-25:]))
