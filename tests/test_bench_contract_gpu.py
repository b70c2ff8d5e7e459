"""bench.py prints ONE JSON line with the driver's contract fields (task statement, section 4):
run here on the small parity-test shape so that the whole GPU suite stays fast."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
                        "--workload", "tiktok"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - d["config"]["edge_layers_per_step"] / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
    rf = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert "step_frac" in rf and 0 < rf["step_frac"] < rf["frac"] * 1.5
    gf = d["gcn_forward"]
    assert gf["edge_layers"] == 3 * 2 * d["config"]["n_edges"] and gf["us"] > 0
    assert abs(gf["edge_layers_per_s"] - gf["edge_layers"] / gf["us"] * 1e6) <= 1e-3 * gf["edge_layers_per_s"]
    lc = d["loss_check"]
    assert lc["rel_err"] <= lc["tolerance"] == 1e-4, lc          # north_star: fp32 loss within 1e-4 relative
    cb = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    assert 0 < d["host_enqueue_us"] <= d["ms_per_step"] * 1e3 * 1.05          # the host's share of a step


def test_bench_sharded_path_one_rank_reports_comm():
    """The row-sharded code path (RCCL collectives, interleaved chains, captured step) on ONE rank: the JSON line
    keeps the contract and carries the communication record."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
                        "--workload", "tiktok", "--force-dist", "--scaling", "strong"], capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and d["value"] > 0
    c = d["comm"]
    # 3-layer GCN: 6 gathers / 6 reduce-scatters (the packed modal chain's two each ride along as grouped pairs) + 2
    # all-reduces (batch rows; the dense-gradient bucket, which carries the regulariser share): 14 launches per step
    assert c["by_kind"]["all_gather"][0] == 6 and c["by_kind"]["reduce_scatter"][0] == 6 and c["by_kind"]["all_reduce"][0] == 2
    assert c["collectives_per_step"] == 14
    assert c["comm_only_ms"] > 0 and c["bytes_per_step"] > 0


def _n_rank_line(extra, timeout=1500):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--workload", "tiktok", "--share-gpu"] + extra, capture_output=True, text=True, timeout=timeout,
                       cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_n_ranks_flow_on_one_gpu_including_scaling_stress():
    """The N > 1 bench flow end to end on a one-GPU box (`--share-gpu`: every rank on GPU 0): self-launch of the ranks,
    per-rank generation of the weak-scaling graph (item degrees summed across ranks), the start-up self-test that selects
    the PEER EXCHANGE (kernels over IPC-mapped windows), the item-side sharded step captured in a hipGraph, barrier +
    max-over-ranks timing, ONE JSON line that proves itself: `rccl_ranks` (world size, backend, every rank's device UUID),
    `loss_vs_n1` (the job's first-step loss against the same job whole on rank 0, 1e-4), `comm` (0 torch.distributed
    collectives per step) and the `scaling_stress` record (configs[4]'s share x N next to the same share on one rank,
    timed in this invocation). Timings mean nothing here; the fields and the records' arithmetic are checked."""
    d = _n_rank_line([])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3 and d["value"] > 0
    assert d["config"]["n_users"] == 2 * 9319 and d["config"]["launch"] == "hipGraph replay"
    c = d["comm"]
    assert c["scheme"] == "item-side" and c["collectives_per_step"] >= 14 and c["bytes_per_step"] > 0
    assert c["transport"].startswith("peer") and c["torch_distributed_collectives_per_step"] == 0, c
    assert "4 rounds through the same windows" in c["selftest"], c       # the start-up check of the visibility protocol
    assert c["by_kind"]["peer_gather"][0] >= 6 and c["by_kind"]["peer_reduce"][0] >= 6 and c["by_kind"]["all_gather"][0] == 0
    assert c["peer"]["exchange_launches_per_step"] > 0 and c["peer"]["windows"] >= c["peer"]["exchange_call_sites_per_step"]
    rr = d["rccl_ranks"]
    assert rr["world_size"] == 2 and len(rr["device_uuids"]) == 2 and rr["distinct_devices"] == 1      # --share-gpu
    lv = d["loss_vs_n1"]
    assert lv["ok"] and lv["rel_err"] <= lv["tolerance"] == 1e-4, lv
    st = d["scaling_stress"]
    assert "error" not in st, st
    assert st["ms_per_step"] > 0 and st["comm"]["scheme"] == "item-side" and st["comm"]["column_chunks"] >= 1
    assert abs(st["edge_layers_per_s"] - 2 * 250_000_000 / (st["ms_per_step"] * 1e-3)) <= 1e-3 * st["edge_layers_per_s"]
    assert st["one_rank"]["edge_layers_per_s"] > 0 and abs(st["per_rank_ratio"] - st["edge_layers_per_s"] / 2 / st["one_rank"]["edge_layers_per_s"]) < 1e-3
    assert st["n1_ms"] > 0 and abs(st["weak_efficiency_vs_n1_ms"] - st["n1_ms"] / st["ms_per_step"]) < 1e-3


def test_bench_n_ranks_flow_over_the_collective_transport():
    """The same flow with `--transport collective` (the A/B: gloo moving device tensors on a shared GPU, RCCL on a real
    node): eager launches, the exchanges counted as torch.distributed collectives, the same loss check."""
    d = _n_rank_line(["--transport", "collective", "--no-stress"])
    c = d["comm"]
    assert d["config"]["launch"] == "eager" and c["transport"].startswith("collective")
    assert c["torch_distributed_collectives_per_step"] >= 14 and c["by_kind"]["all_gather"][0] >= 6
    assert d["loss_vs_n1"]["ok"], d["loss_vs_n1"]


def test_bench_falls_back_to_collectives_when_the_peer_exchange_fails_after_its_self_test():
    """`--peer-sabotage`: the self-test passes, then the last rank stops signalling its reduce-scatter partials. The other
    rank's waits give up (0.5 s, later waits do not spin again), the AGREED health check raises on every rank together, the
    job drops the peer exchange, rebuilds the step over torch.distributed's collectives and still prints its line - with
    the reason - instead of hanging or dying: what the driver's SCALE run needs if the exchange misbehaves on real links."""
    d = _n_rank_line(["--no-stress", "--peer-sabotage"])
    c = d["comm"]
    assert c["transport"].startswith("collective") and "peer_rejected" in c, c
    assert "timed out" in c["peer_rejected"] or "first eager steps" in c["peer_rejected"], c
    assert d["config"]["launch"] == "eager" and c["torch_distributed_collectives_per_step"] >= 14
    assert d["loss_vs_n1"]["ok"], d["loss_vs_n1"]
    assert d["value"] > 0
