"""Builds libfitslam_frontier.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so travels to the
GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(CSRC, "libfitslam_frontier.so")
SOURCES = ["fs_raymarch.hip", "fs_fim.hip", "fs_rank.hip", "fs_sort.hip", "fs_gridops.hip", "fs_keyframes.hip", "fs_capi.hip"]
HEADERS = ["fs_internal.h", os.path.join("..", "..", "include", "fitslam_frontier.h")]

# -ffp-contract=off: the ray set-up (fp64) and the landmark transform (fp32) must round exactly like
# the specification; fused multiply-adds appear only where the code calls fma explicitly.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
               "-Wall", "-Wno-unused-function", "-Wno-unused-result"]
if os.environ.get("FS_FIM_BOUNDS"):       # development: range-checked global accesses in the FIM kernels (counter 30)
    HIPCC_FLAGS.append("-DFS_FIM_BOUNDS")
if os.environ.get("FS_FIM_STAMPS"):       # development: per-phase cycle counters of the FIM worker in counters 16..24
    HIPCC_FLAGS.append("-DFS_FIM_STAMPS")


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built (there is no CPU fallback)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    cc = hipcc()
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        src_path = os.path.join(CSRC, src)
        hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src_path), hdr_t):
            cmd = [cc, *HIPCC_FLAGS, "-c", src_path, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
