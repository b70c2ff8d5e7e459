"""Does running the projection GEMMs concurrently with the (independent) GCN SpMM chain pay inside a hipGraph?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmssl_amd import graph, ops, synth
U, I, E, dv, dt = synth.SHAPES["baby"]
raw = synth.interaction_matrix(U, I, E); ui, iu = synth.normalised_pair(raw)
P = (graph.GraphPlan(ui), graph.GraphPlan(iu))
d = 64
Fi = torch.randn(I, dv, device="cuda"); Ft = torch.randn(I, dt, device="cuda")
Wi = torch.randn(d, dv, device="cuda") * .02; Wt = torch.randn(d, dt, device="cuda") * .02
Xu = torch.randn(U, d, device="cuda"); Xi = torch.randn(I, d, device="cuda")

def gemms():
    return ops.linear(Fi, Wi), ops.linear(Ft, Wt)
def gcn():
    u, i = Xu, Xi
    for l in range(3):
        u = ops.spmm(P[0], i); i = ops.spmm(P[1], u)
    return u, i
def modal(x1, x2):
    a = ops.spmm(P[1], ops.spmm(P[0], x1)); b = ops.spmm(P[1], ops.spmm(P[0], x2))
    return a, b

def bench(fn, tag):
    with torch.no_grad():
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3): fn(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(5): fn(s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g.replay(); e0.record()
        for _ in range(20): g.replay()
        e1.record(); torch.cuda.synchronize()
        print("%-44s %.1f us" % (tag, e0.elapsed_time(e1) * 10))

side = torch.cuda.Stream(); side2 = torch.cuda.Stream()
def seq(s):
    x1, x2 = gemms(); gcn(); modal(x1, x2)
def par(s):
    side.wait_stream(s)
    with torch.cuda.stream(side):
        gcn()
    x1, x2 = gemms(); modal(x1, x2)
    s.wait_stream(side)
def par3(s):
    side.wait_stream(s); side2.wait_stream(s)
    with torch.cuda.stream(side):
        gcn()
    with torch.cuda.stream(side2):
        x2 = ops.linear(Ft, Wt); b = ops.spmm(P[1], ops.spmm(P[0], x2))
    x1 = ops.linear(Fi, Wi); a = ops.spmm(P[1], ops.spmm(P[0], x1))
    s.wait_stream(side); s.wait_stream(side2)
bench(lambda s: gemms(), "2 GEMMs only")
bench(lambda s: gcn(), "GCN 6 SpMM only")
bench(seq, "sequential: GEMMs + GCN + modal")
bench(par, "2 streams: (GEMMs+modal) || GCN")
bench(par3, "3 streams: img-chain || txt-chain || GCN")
