"""Build libdmvs_hip.so (gfx950 code object + C ABI) -- and libdmvs_probe.so, the bench-only measurement probes of
include/dmvs_probe.h -- in-tree with hipcc.

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
PROBE_LIB = os.path.join(PKG, "libdmvs_probe.so")      # measurement probes (bench.py's untimed roofline legs); never loaded by the path
PROBE_SOURCES = ["probe/getcost_probe.hip", "probe/random_line_gather.hip"]
SOURCES = ["conv2d_k33.hip", "conv2d_k55.hip", "conv2d_k77.hip", "conv2d_k15.hip", "conv2d.hip", "stem.hip", "conv3d.hip", "warp_quad.hip", "warp_bwd.hip", "warp_bwd_win.hip", "warp_init_bwd_win.hip", "misc.hip", "mask_upsample.hip", "optim.hip", "norm.hip", "fusion.hip"]


def _deps():
    out = [os.path.join(ROOT, "include", "dmvs.h"), os.path.join(ROOT, "include", "dmvs_probe.h")]
    for base, _, files in os.walk(CSRC):
        out += [os.path.join(base, f) for f in files]
    return out


def _stale() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(PROBE_LIB):
        return True
    t = min(os.path.getmtime(LIB), os.path.getmtime(PROBE_LIB))
    return any(os.path.getmtime(d) > t for d in _deps())


def build_hip(force: bool = False, save_temps: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    import time
    t_start = time.time()          # the libraries are stamped with the START of the build: a source edited while it ran makes them stale
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
    # an object is reused when it is newer than its source and than every header (no finer dependency tracking: a header change rebuilds all)
    hdr_t = max(os.path.getmtime(d) for d in _deps() if d.endswith(".h"))

    def compile_one(src):
        obj = os.path.join(objdir, src.replace("/", "_") + ".o")
        if not force and not save_temps and os.path.exists(obj) and os.path.getmtime(obj) > max(hdr_t, os.path.getmtime(os.path.join(CSRC, src))):
            return obj
        cmd = [hipcc] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[diffmvs_amd.build]", " ".join(cmd), flush=True)
        t0 = time.time()
        subprocess.run(cmd, check=True, cwd=objdir if save_temps else ROOT)
        os.utime(obj, (t0, t0))
        return obj

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as pool:
        objs = list(pool.map(compile_one, SOURCES + PROBE_SOURCES))
    for lib, lobjs in ((LIB, objs[:len(SOURCES)]), (PROBE_LIB, objs[len(SOURCES):])):
        link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + lobjs + ["-o", lib]
        if verbose:
            print("[diffmvs_amd.build]", " ".join(link), flush=True)
        subprocess.run(link, check=True, cwd=ROOT)
        os.utime(lib, (t_start, t_start))
    return LIB


if __name__ == "__main__":
    build_hip(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv)
    print(LIB)
