"""Builds dynamicpdb_amd/csrc/libdfold_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m dynamicpdb_amd.build_ext [--force]

The .so is built IN-TREE (it is git-ignored but travels to the GPU box with the snapshot).
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(HERE, "..", "include")
LIB = os.path.join(CSRC, "libdfold_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result",
         # the 2x5-tile GEMM epilogue must unroll completely (160 accumulator registers indexed statically), else the
         # accumulators are demoted to scratch
         "-mllvm", "-pragma-unroll-threshold=100000", "-mllvm", "-unroll-threshold=2000"]


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest():
    h = hashlib.sha256()
    for f in sources() + sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + ["../../include/dfold_hip.h"]:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode())
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    stamp = os.path.join(CSRC, ".build_stamp")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)

    def cc(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [HIPCC] + FLAGS + ["-I", INCLUDE, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(cc, sources()))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    with open(stamp, "w") as fh:
        fh.write(dig)
    if verbose:
        print(f"built {LIB} from {len(objs)} sources")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
