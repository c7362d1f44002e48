#!/usr/bin/env python3
"""Build libipdm.so (gfx950 only) with hipcc, in-tree.  No torch dependency: the library is a plain
C-ABI shared object (include/ipdm.h) that the host side loads with ctypes.

    python inverseproblemwithdiffusionmodel_amd/csrc/build.py [--force] [--verbose]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
INCLUDE = os.path.join(REPO, "include")
OUT = os.path.join(os.path.dirname(HERE), "libipdm.so")
OBJ_DIR = os.path.join(HERE, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", f"-I{INCLUDE}", f"-I{HERE}"]


def sources():
    return sorted(os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = [os.path.join(INCLUDE, "ipdm.h")] + [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    jobs = []
    objs = []
    for src in sources():
        obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            flags = list(FLAGS)
            if os.path.basename(src).startswith("conv"):
                # the MFMA convolution is an fma chain by construction; let its prologue VALU math fuse too
                flags[flags.index("-ffp-contract=off")] = "-ffp-contract=fast"
            flags += os.environ.get("IPDM_EXTRA_HIPCC_FLAGS", "").split()     # diagnostic builds (e.g. -DIPDM_WBX3_TRACE)
            jobs.append([HIPCC, *flags, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)
    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(OUT, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", OUT])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv))
