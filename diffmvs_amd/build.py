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
SOURCES = ["conv2d.hip", "stem.hip", "conv3d.hip", "warp.hip", "warp_win.hip", "warp_init_win.hip", "warp_quad.hip", "warp_bwd.hip", "warp_bwd_win.hip", "warp_init_bwd_win.hip", "misc.hip", "optim.hip", "norm.hip", "fusion.hip"]


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
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I", os.path.join(ROOT, "include"), "-I", CSRC,
           "-munsafe-fp-atomics",          # hardware global_atomic_add_f32/f64 (gradient scatter, GroupNorm stats)
           "-Wall", "-Wno-unused-function"]
    if save_temps:
        tmp = os.path.join(ROOT, "build", "temps")
        os.makedirs(tmp, exist_ok=True)
        cmd += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print("[diffmvs_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=(os.path.join(ROOT, "build", "temps") if save_temps else ROOT))
    return LIB


if __name__ == "__main__":
    build_hip(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv)
    print(LIB)
