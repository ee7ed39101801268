"""Build a macro variant of the GPU library next to the default one, for A/B runs of compile-time experiments:

    python tools/build_variant.py nopipe -DDMVS_QUAD_PIPE=0 --only warp_quad.hip   # -> tools/calib/libdmvs_hip_nopipe.so
    python tools/build_variant.py kyrolled -DDMVS_CONV_KY_ROLLED
    python tools/build_variant.py gcexp1 -DDMVS_GC_EXP=1 --only warp_quad.hip     # one translation unit, the rest from build/obj
    gpurun -- 'CONV_2D_ONLY=1 python tools/conv_bench.py > a.jsonl; CONV_2D_ONLY=1 CONV_LIB=tools/calib/libdmvs_hip_kyrolled.so python tools/conv_bench.py > b.jsonl'

Runs in the build container (hipcc cross-compiles gfx950); the .so is git-ignored and travels to the GPU box with the snapshot.
The whole library is rebuilt with the extra flags (objects under build/variant_<name>/), the default library is not touched."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffmvs_amd.build import CSRC, SOURCES  # noqa: E402


def main():
    name, extra = sys.argv[1], sys.argv[2:]
    only = None
    if "--only" in extra:       # --only warp_quad.hip: recompile that translation unit alone, link it with the default build's objects
        i = extra.index("--only")
        only = extra[i + 1].split(",")
        extra = extra[:i] + extra[i + 2:]
    objdir = os.path.join(ROOT, "build", "variant_" + name)
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-munsafe-fp-atomics",
             "-Wno-unused-function"] + extra

    def one(src):
        if only is not None and src not in only:
            return os.path.join(ROOT, "build", "obj", src + ".o")
        obj = os.path.join(objdir, src + ".o")
        subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-c", os.path.join(CSRC, src), "-o", obj], check=True, cwd=ROOT, stderr=subprocess.DEVNULL)
        return obj

    with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as pool:
        objs = list(pool.map(one, SOURCES))
    out = os.path.join(ROOT, "tools", "calib", "libdmvs_hip_%s.so" % name)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out], check=True, cwd=ROOT)
    print(out)


if __name__ == "__main__":
    main()
