"""Build libddx.so (HIP kernels for gfx950 + host C++) in-tree.

hipcc cross-compiles gfx950 code objects without a GPU, so this runs in the build container; the
resulting ``doubletdetection_amd/libddx.so`` travels to the GPU box with the repository snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libddx.so")

# per-file flags: the kNN screen compares MFMA results right away -- keep the accumulators in VGPRs (no v_accvgpr_read)
EXTRA_FLAGS = {"k_knn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}
HIP_SOURCES = ["ddx_api.hip", "k_sparse.hip", "k_pca.hip", "k_bitplane.hip", "k_knn.hip", "k_prologue.hip", "k_louvain.hip"]
CXX_SOURCES = ["louvain.cpp", "hostmath.cpp"]
HEADERS = ["ddx_internal.h", "ddx_prims.h", os.path.join("..", "..", "include", "ddx.h")]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    for src in HIP_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result"]
            cmd += EXTRA_FLAGS.get(src, [])
            cmd += os.environ.get("DDX_EXTRA_HIPCC_FLAGS", "").split()     # experiments (e.g. -DDDX_LDS_PANEL_ROWS=592)
            cmd += ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
    for src in CXX_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-pthread", "-Wall", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("compile failed: " + " ".join(cmd))
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-pthread", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
