"""Static check of hand-counted `s_waitcnt vmcnt` protocols in the gfx950 ISA of csrc/*.hip.

Why: some kernels issue their global loads from inline asm (so that neither the IR passes nor the machine scheduler
can sink them) and wait for them with hand-written `s_waitcnt vmcnt(N)`. To the compiler the destination of such a load
is an ordinary value that exists as soon as the asm statement has executed, so register allocation may COPY it (phi
elimination at a loop back-edge, a spill to an AGPR, a rematerialised address computation that reuses the register)
before the hand-written wait - a read of a register whose load is still in flight. The copy then carries whatever the
register held before (the operand of eight steps earlier), but only when the load is late: alone the kernel is
bit-stable, beside memory-bound kernels it is intermittently wrong. That was the round-5 "text-projection weight
gradient 6e-3 off in one run of three or four" (wgrad10_kernel: `v_mov_b64 v[94:95], v[130:131]` at the back-edge).

Model (what LLVM's SIInsertWaitcnts assumes for gfx9: vector memory operations retire in issue order):
  * a FIFO of outstanding vector-memory operations; a load carries its destination VGPRs, a store / LDS-DMA load none;
  * `s_waitcnt vmcnt(N)` retires the oldest entries until N are left;
  * any instruction that names a VGPR which is the destination of a still-queued load is a violation;
  * a `; vmcnt-landed v[a:b] ...` comment left by an asm statement is the author's assertion that the loads of those
    registers (and every older operation) have retired: for waits picked at run time among several immediates, which a
    walk of the control-flow graph cannot correlate with the branch that issued the loads.
Control flow: every path of the function's control-flow graph is followed, a block once per distinct queue state (a
software pipeline reaches its steady state after one trip, so loops converge).

    python tools/vmcnt_check.py [file.hip ...]      (default: every csrc/*.hip that has an inline-asm load)
exit status 1 and one line per violation; used by tests/test_isa_hazards.py.
"""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "mmssl_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-S", "--cuda-device-only"]

_REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
_LOAD = re.compile(r"^(global|buffer|flat|scratch)_load_")
_STORE = re.compile(r"^(global|buffer|flat|scratch)_(store|atomic)")
_VMCNT = re.compile(r"vmcnt\((\d+)\)")


def _regs(text):
    out = set()
    for m in _REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def functions(asm_text):
    """{kernel name: [(line_no, label or None, instruction text)]} for every function in a hipcc -S listing."""
    funcs, cur, name = {}, None, None
    for no, raw in enumerate(asm_text.splitlines(), 1):
        m = re.search(r";\s*vmcnt-landed\s+(.*)$", raw)
        if m and cur is not None:                       # the author's assertion (see csrc/projection.hip tie4)
            cur.append((no, None, "vmcnt_landed " + m.group(1).strip()))
            continue
        line = raw.split(";")[0].rstrip()
        s = line.strip()
        if not s:
            continue
        m = re.match(r"^([A-Za-z_][\w.$]*):", line)
        if m and not m.group(1).startswith(".L"):
            name, cur = m.group(1), []
            funcs[name] = cur
            continue
        if cur is None:
            continue
        m = re.match(r"^(\.L[\w.$]+):", line)
        if m:
            cur.append((no, m.group(1), None))
            continue
        if s.startswith("."):
            if s.startswith(".Lfunc_end") or s.startswith(".section") or s.startswith(".amdhsa_kernel"):
                cur = None
            continue
        cur.append((no, None, s))
    return {k: v for k, v in funcs.items() if any(i and i.startswith("s_endpgm") for _, _, i in v)}


def check_function(items, max_states=256):
    """[(line_no, instruction, register, load line_no)] violations of one function: every path through its control-flow
    graph, a block revisited once per distinct queue state (a software pipeline reaches its steady state after one
    trip, so loops converge; `max_states` per block bounds pathological cases)."""
    n = len(items)
    labels = {lab: k for k, (_, lab, _) in enumerate(items) if lab}
    viol, seen = [], set()

    def step(queue, no, ins):
        op = ins.split()[0]
        if op == "s_waitcnt":
            m = _VMCNT.search(ins)
            if m:
                return queue[max(0, len(queue) - int(m.group(1))):]
            if re.fullmatch(r"s_waitcnt\s+(0|0x0)", ins):
                return ()
            return queue
        if op == "vmcnt_landed":
            regs = _regs(ins)
            last = max([q for q, (dest, _) in enumerate(queue) if dest & regs], default=-1)
            return queue[last + 1:]
        is_load = bool(_LOAD.match(op))
        lds = is_load and ("_lds_" in op or ins.rstrip().endswith(" lds"))
        # a load INTO a register another load still targets is fine (loads retire in order: the later data wins);
        # its address operands are read at issue like any other source
        touched = _regs(ins.split(",", 1)[1] if (is_load and not lds and "," in ins) else ins)
        for dest, lno in queue:
            hit = touched & dest
            if hit and (no, lno) not in seen:
                seen.add((no, lno))
                viol.append((no, ins, "v%d" % min(hit), lno))
        if is_load:
            queue = queue + ((frozenset() if lds else frozenset(_regs(ins.split(",")[0])), no),)
        elif _STORE.match(op):
            queue = queue + ((frozenset(), no),)
        return queue[-64:]           # the counter saturates at 63: older operations have retired

    visited = {}
    work = [(0, ())]
    while work:
        k, queue = work.pop()
        st = visited.setdefault(k, set())
        if queue in st or len(st) >= max_states:
            continue
        st.add(queue)
        while k < n:
            no, lab, ins = items[k]
            if lab is not None and k in visited and k != 0:
                pass
            if ins is None:          # a label inside the run: a block boundary (another path may enter here)
                if queue in visited.setdefault(-k - 1, set()):
                    break
                visited[-k - 1].add(queue)
                k += 1
                continue
            queue = step(queue, no, ins)
            op = ins.split()[0]
            if op == "s_endpgm" or op.startswith("s_setpc") or op.startswith("s_trap"):
                break
            m = re.match(r"(s_cbranch_\w+|s_branch)\s+(\.L[\w.$]+)", ins)
            if m:
                tgt = labels.get(m.group(2))
                if tgt is not None:
                    work.append((tgt, queue))
                if m.group(1) == "s_branch":
                    break
            k += 1
    viol.sort()
    return viol


def compile_listing(src):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        r = subprocess.run([HIPCC] + FLAGS + ["-o", out, src], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc -S failed for %s:\n%s" % (src, r.stderr))
        with open(out) as f:
            return f.read()


def check_file(src):
    """{kernel: violations} for the kernels of one .hip source (compiled here for gfx950)."""
    res = {}
    for name, items in functions(compile_listing(src)).items():
        v = check_function(items)
        if v:
            res[name] = v
    return res


def asm_load_sources():
    out = []
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".hpp")):
            with open(os.path.join(CSRC, f)) as fh:
                if re.search(r'asm\s+volatile\(\s*"(global|buffer)_load', fh.read()):
                    out.append(f)
    # headers are checked through the units that include them
    hips = [f for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    need = set(f for f in out if f.endswith(".hip"))
    for h in (f for f in out if f.endswith(".hpp")):
        for f in hips:
            with open(os.path.join(CSRC, f)) as fh:
                if h in fh.read():
                    need.add(f)
    return sorted(need)


def main(argv):
    files = argv or [os.path.join(CSRC, f) for f in asm_load_sources()]
    bad = 0
    for src in files:
        res = check_file(src)
        print("%s: %s" % (os.path.basename(src), "clean" if not res else "%d kernel(s) with in-flight register reads" % len(res)))
        for name, viol in res.items():
            for no, ins, reg, lno in viol:
                bad += 1
                print("  %s\n    line %d: `%s` touches %s, destination of the load at line %d still in flight" % (
                    name, no, ins, reg, lno))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
