"""The peer exchange (csrc/peer.hip, mmssl_amd/peer.py): row shards move between the ranks through IPC-mapped device
windows written and read by kernels, ordered by epoch flags - no collective library in the data path. Validated with
several PROCESSES sharing this GPU (cross-process IPC handles work on one device; what a one-GPU box cannot show is the
cross-device coherence of the flags, which follows the HSA memory model: system-scope release / acquire)."""
import os
import subprocess
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.gpu


def _run(world, tmp_path, script="_peer_worker.py", extra=()):
    import test_dist_cpu as T
    port = T._free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, script), str(r), str(world), str(port)] + list(extra) +
                              [str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-2500:] for o in outs)
    return [torch.load(os.path.join(str(tmp_path), "r%d.pt" % r)) for r in range(world)]


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_peer_primitives_between_processes_sharing_the_gpu(tmp_path, world):
    """all-gather by push (contiguous and row-pitched shards), reduce-scatter by pull with the fixed rank order (copied in
    and produced in place), the small all-reduce: four eager steps, then the sequence captured in a hipGraph and replayed
    four times with new inputs - every result equal to the BITS each rank computes for itself."""
    recs = _run(world, tmp_path)
    for o in recs:
        assert o["stats"]["world"] == world and o["stats"]["call_sites"] == 5 and o["stats"]["windows"] == 5


def test_a_wait_that_cannot_be_satisfied_gives_up_and_is_reported(tmp_path):
    """Rank 0 of two waits on a channel rank 1 never signals: the wait kernel returns after its timeout (the device is
    not hung), the context's error word names the missing peer and PeerTransport.check() raises."""
    recs = _run(2, tmp_path, extra=("timeout",))
    assert recs[0]["timed_out"] and "0x2" in recs[0]["msg"], recs[0]
    assert not recs[1]["timed_out"]


def test_a_set_up_failure_on_one_rank_is_raised_by_every_rank(tmp_path):
    """The window export fails on rank 2 of three (the library call returns an error there): all three ranks raise the
    same MmsslError naming rank 2 from the same host-side exchange - none is left waiting in it - and a collective issued
    afterwards still works (what bench.choose_transport relies on when it falls back to RCCL)."""
    recs = _run(3, tmp_path, extra=("setupfail",))
    for r in recs:
        assert r["raised"] and "rank 2" in r["msg"] and r["sum"] == 6.0, r
