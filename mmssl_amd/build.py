"""Builds libmmssl_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m mmssl_amd.build [--force]

Each csrc/*.hip is compiled to an object (in parallel) and linked into
mmssl_amd/libmmssl_hip.so. The .so is git-ignored but travels to the GPU box with gpurun.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libmmssl_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-munsafe-fp-atomics"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    hdrs.append(os.path.join(HERE, "..", "include", "mmssl_hip.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, force, hdr_m):
    obj = os.path.join(OBJ, src[:-4] + ".o")
    s = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(s), hdr_m):
        return obj, False
    cmd = [HIPCC] + FLAGS + ["-c", s, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hdr_m = _deps_mtime()
    with ThreadPoolExecutor(max_workers=8) as ex:
        res = list(ex.map(lambda s: _compile(s, force, hdr_m), _sources()))
    objs = [o for o, _ in res]
    build.last = {"compiled": [os.path.basename(o) for o, c in res if c],
                  "reused": [os.path.basename(o) for o, c in res if not c], "linked": False}
    if force or any(c for _, c in res) or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        build.last["linked"] = True
        if verbose:
            print("built %s: %d unit(s) compiled for gfx950, %d object(s) reused" % (
                LIB, len(build.last["compiled"]), len(build.last["reused"])))
    elif verbose:
        print("up to date: %s (%d objects newer than their sources and headers; --force recompiles)" % (LIB, len(objs)))
    return LIB


build.last = None


if __name__ == "__main__":
    build(force="--force" in sys.argv)
