"""`python bench.py --gpus N` WITHOUT a launcher starts its own ranks (torch.distributed.run, one process per GPU): the
launch path - self-launch, rendezvous on 127.0.0.1, barrier / max-over-ranks skeleton, ONE JSON line from rank 0, exit
code - checked on CPU with gloo through bench.py's --launch-check dry run; and under the driver's explicit launcher."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_lines(out):
    return [json.loads(ln) for ln in out.splitlines() if ln.startswith("{")]


@pytest.mark.parametrize("n", [2, 3])
def test_bench_self_launches_its_ranks(n):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1",
                        "--launch-check", "--backend", "gloo"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    recs = _json_lines(r.stdout)
    assert len(recs) == 1 and recs[0]["launch_check"] is True and recs[0]["n_gpus"] == n, r.stdout[-1000:]
    assert recs[0]["rank_sum"] == n * (n + 1) / 2


def test_bench_under_the_drivers_launcher_and_world_mismatch():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
            "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")]
    r = subprocess.run(base + ["--gpus", "2", "--launch-check", "--backend", "gloo"], capture_output=True, text=True,
                       timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    recs = _json_lines(r.stdout)
    assert len(recs) == 1 and recs[0]["n_gpus"] == 2
    # --gpus disagreeing with the launcher's world size is an error, not a silent 2-rank run labelled 4
    r = subprocess.run(base + ["--gpus", "4", "--launch-check", "--backend", "gloo"], capture_output=True, text=True,
                       timeout=300, cwd=ROOT)
    assert r.returncode != 0 and not _json_lines(r.stdout)


def test_stress_watchdog_prints_the_headline_line_and_leaves_cleanly():
    """A scaling_stress run that hangs (a collective one rank never joins) must not cost the N > 1 line: after the limit rank
    0 prints the headline with the time-out recorded and every rank exits 0."""
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "out = {'metric': 'm', 'value': 1.0}\n"
            "bench.stress_watchdog(1, out, int(sys.argv[1]))\n"
            "time.sleep(60)\nprint('not reached'); sys.exit(5)\n" % ROOT)
    for rank in (0, 1):
        r = subprocess.run([sys.executable, "-c", code, str(rank)], capture_output=True, text=True, timeout=120, cwd=ROOT)
        assert r.returncode == 0 and "not reached" not in r.stdout, (r.returncode, r.stdout, r.stderr[-500:])
        recs = _json_lines(r.stdout)
        if rank == 0:
            assert len(recs) == 1 and recs[0]["value"] == 1.0 and "timed out" in recs[0]["scaling_stress"]["error"]
        else:
            assert recs == []
    # cancelled in time: nothing happens
    code2 = ("import sys, time; sys.path.insert(0, %r); import bench\n"
             "t = bench.stress_watchdog(1, {}, 0); t.cancel(); time.sleep(2); print('alive')\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 0 and r.stdout.strip() == "alive"
