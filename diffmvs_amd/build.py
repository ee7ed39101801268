"""Build libdmvs_hip.so (gfx950 code object + C ABI) in-tree with hipcc.

    python -m diffmvs_amd.build [--force] [--save-temps]

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the
resulting .so is git-ignored but travels to the GPU box with the source snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libdmvs_hip.so")
SOURCES = ["conv2d_k33.hip", "conv2d_k55.hip", "conv2d_k77.hip", "conv2d_k15.hip", "conv2d.hip", "stem.hip", "conv3d.hip", "warp_quad.hip", "warp_bwd.hip", "warp_bwd_win.hip", "warp_init_bwd_win.hip", "misc.hip", "optim.hip", "norm.hip", "fusion.hip"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "dmvs.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force: bool = False, save_temps: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
             "-I", os.path.join(ROOT, "include"), "-I", CSRC,
             "-munsafe-fp-atomics",          # hardware global_atomic_add_f32/f64 (gradient scatter, GroupNorm stats)
             "-Wall", "-Wno-unused-function"]
    objdir = os.path.join(ROOT, "build", "temps" if save_temps else "obj")
    os.makedirs(objdir, exist_ok=True)
    if save_temps:
        flags += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]

    # one hipcc per translation unit, in parallel (the tiled conv2d kernel is split over four translation units for this: as one file it was 6 of the 7 minutes), then one link
    def compile_one(src):
        obj = os.path.join(objdir, src + ".o")
        cmd = [hipcc] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[diffmvs_amd.build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=objdir if save_temps else ROOT)
        return obj

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
    if verbose:
        print("[diffmvs_amd.build]", " ".join(link), flush=True)
    subprocess.run(link, check=True, cwd=ROOT)
    return LIB


if __name__ == "__main__":
    build_hip(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv)
    print(LIB)
