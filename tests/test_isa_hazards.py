"""The gfx950 ISA of the kernels that wait for memory by hand (csrc/linear.hip until round 6, csrc/projection.hip's
LDS-DMA pieces) must never touch a register whose load is still in flight: the round-5 intermittent weight gradient was a
`v_mov_b64` the register allocator put at a loop back-edge in front of a hand-written `s_waitcnt` (tools/vmcnt_check.py).
CPU tests: hipcc cross-compiles the listings here."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import vmcnt_check as V  # noqa: E402

CSRC = os.path.join(ROOT, "mmssl_amd", "csrc")

# the shape of the round-5 bug, reduced: an eight-deep ring would be long, a two-deep one has the same structure -
# slot 0 is reloaded into a fresh register and copied into the loop-carried one BEFORE the wait that covers it
BUGGY = """
kern:
	global_load_dwordx4 v[2:5], v20, s[0:1]
	global_load_dwordx4 v[6:9], v21, s[0:1]
.LBB0_1:
	s_waitcnt vmcnt(1)
	v_mfma_f32_16x16x4_f32 v[40:43], v2, v3, v[40:43]
	global_load_dwordx4 v[30:33], v20, s[0:1]
	s_waitcnt vmcnt(1)
	v_mfma_f32_16x16x4_f32 v[40:43], v6, v7, v[40:43]
	global_load_dwordx4 v[6:9], v21, s[0:1]
	v_mov_b64_e32 v[2:3], v[30:31]
	v_mov_b64_e32 v[4:5], v[32:33]
	s_cbranch_scc1 .LBB0_1
	s_waitcnt vmcnt(0)
	s_endpgm
.Lfunc_end0:
"""
FIXED = BUGGY.replace("\tv_mov_b64_e32 v[2:3], v[30:31]", "\ts_waitcnt vmcnt(1)\n\tv_mov_b64_e32 v[2:3], v[30:31]")
# a dead prefetch's register handed to the epilogue before the ring is drained (the second hazard of the same kernel)
REUSED = """
kern:
	global_load_dwordx4 v[2:5], v20, s[0:1]
	v_and_b32_e32 v2, 63, v0
	s_waitcnt vmcnt(0)
	s_endpgm
.Lfunc_end0:
"""
# a wait that lives in another block than the load (the checker follows the control-flow graph, not the listing order)
BRANCHY = """
kern:
	s_branch .LBB0_2
.LBB0_1:
	s_waitcnt vmcnt(0)
	v_add_f32_e32 v9, v2, v3
	s_endpgm
.LBB0_2:
	global_load_dwordx4 v[2:5], v20, s[0:1]
	s_branch .LBB0_1
.Lfunc_end0:
"""


def _viol(text):
    f = V.functions(text)
    assert list(f) == ["kern"]
    return V.check_function(f["kern"])


def test_checker_flags_a_copy_of_an_in_flight_register_at_the_back_edge():
    v = _viol(BUGGY)
    assert [x[1] for x in v] == ["v_mov_b64_e32 v[2:3], v[30:31]", "v_mov_b64_e32 v[4:5], v[32:33]"], v
    assert _viol(FIXED) == []


def test_checker_flags_reuse_of_a_dead_prefetch_register_and_follows_branches():
    v = _viol(REUSED)
    assert len(v) == 1 and v[0][2] == "v2"
    assert _viol(BRANCHY) == []
    # loads retire in order: a second load into the same register is not a hazard, its address operands are
    assert _viol("kern:\n\tglobal_load_dword v4, v[16:17], off\n\tglobal_load_dword v4, v[16:17], off offset:4\n"
                 "\ts_waitcnt vmcnt(0)\n\ts_endpgm\n") == []
    assert len(_viol("kern:\n\tglobal_load_dwordx2 v[16:17], v[2:3], off\n\tglobal_load_dword v4, v[16:17], off\n"
                     "\ts_waitcnt vmcnt(0)\n\ts_endpgm\n")) == 1


UNITS = ("graph.hip", "infonce.hip", "linear.hip", "peer.hip", "projection.hip")


@pytest.fixture(scope="module")
def listings():
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(4) as ex:
        return dict(zip(UNITS, ex.map(lambda f: V.compile_listing(os.path.join(CSRC, f)), UNITS)))


@pytest.mark.parametrize("unit", UNITS)
def test_no_kernel_touches_a_register_whose_load_is_in_flight(listings, unit):
    bad = {}
    for name, items in V.functions(listings[unit]).items():
        v = V.check_function(items)
        if v:
            bad[name] = v[:4]
    assert not bad, bad


def test_hand_waited_units_are_the_ones_checked():
    """Every unit with an inline-asm load / DMA or a hand-written vmcnt is in the parametrised list above."""
    hand = set()
    for f in os.listdir(CSRC):
        if f.endswith(".hip"):
            text = open(os.path.join(CSRC, f)).read()
            if re.search(r"vm_wait_n<|glds16|asm\s+volatile\(\s*\"(global|buffer)_load|s_waitcnt vmcnt", text):
                hand.add(f)
    assert hand <= set(UNITS), hand


def test_projx_memory_queue_has_one_shape(listings):
    """projx_sk_kernel waits for its LDS-DMA pieces with CONSTANT counts (vmcnt(10) per step, vmcnt(16) after the
    prologue) that assume: per slice two DMA pieces, then four 16-byte register loads, on every path. Pin that in the
    ISA: in program order the kernel's vector-memory reads are (4 loads, 2 DMA) x 3 + 4 loads in the prologue and
    (2 DMA, 4 loads) in each of the four steps of the loop body, and the only hand-written waits are those counts."""
    f = V.functions(listings["projection.hip"])
    name = [k for k in f if "projx_sk_kernel" in k]
    assert len(name) == 1
    text = listings["projection.hip"]
    body = text[text.index(name[0] + ":"):]
    body = body[:body.index(".Lfunc_end")]
    assert "scratch_" not in body                       # no spills anywhere in the kernel
    hand = re.findall(r";;#ASMSTART\s*\n\s*s_waitcnt vmcnt\((\d+)\)", body)
    # the pipeline ends at the hand-written vmcnt(0); behind it (round 6) are the in-kernel epilogues, whose loads are
    # ordinary compiler-waited ones
    drain = re.search(r";;#ASMSTART\s*\n\s*s_waitcnt vmcnt\(0\)", body)
    assert drain is not None
    ops = []
    for line in body[:drain.start()].splitlines():
        ins = line.strip()
        if not ins or ins.startswith((";", ".")):
            continue
        op = ins.split()[0]
        if op == "global_load_lds_dwordx4":
            ops.append("D")
        elif op.startswith("global_load_dwordx4"):
            ops.append("L")
        elif op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
            ops.append("?")           # any other vector-memory read (a spill reload, a stray load) breaks the counts
    seq = "".join(ops)
    assert seq == "LLLLDD" * 3 + "LLLL" + "DDLLLL" * 4, seq
    assert hand == ["16", "10", "10", "10", "10", "0"], hand
