"""Builds vectordb_amd/lib/libepsilla_gfx950.so from vectordb_amd/csrc with hipcc for gfx950 (in-tree, so the
.so travels with the repo snapshot to the GPU box).  `python -m vectordb_amd.build [--force]`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lib")
OBJ = os.path.join(OUT, "obj")
LIB = os.path.join(OUT, "libepsilla_gfx950.so")
SOURCES = ["index.cpp", "shard_group.cpp", "exchange.cpp", "flat_kernels.hip", "traverse.hip", "mfma_filter.hip", "graph_build.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-I" + os.path.join(os.path.dirname(HERE), "include")]


# lab only: extra -D switches for kernel experiments (e.g. EPS_BUILD_DEFS="-DEPS_TRV_U=8"); the product build sets none
FLAGS += os.environ.get("EPS_BUILD_DEFS", "").split()


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(SRC, f) for f in os.listdir(SRC) if f.endswith((".hpp", ".h"))]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "epsilla_gfx950.h"))
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(SRC, s)
        obj = os.path.join(OBJ, s.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            cmd = [hipcc] + FLAGS + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", src, "-o", obj]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-6000:]))
        if verbose and r.stderr.strip():
            print(r.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"])   # (-ldl: exchange.cpp resolves RCCL at run time)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
