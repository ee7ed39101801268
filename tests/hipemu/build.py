"""TEST-ONLY: compile diffmvs_amd/csrc/*.hip for the host with g++ against the fiber-based
HIP emulation header (tests/hipemu/hip/hip_runtime.h) -> tests/hipemu/libdmvs_emu.so.
Used by the CPU tests to execute the kernel sources without a GPU; never by the product."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "diffmvs_amd", "csrc")
LIB = os.path.join(HERE, "libdmvs_emu.so")


def build_emu(force=False, defines=(), tag=""):
    """defines / tag: a macro variant of the kernel sources (tools/build_variant.py builds the same variants for the GPU) as
    libdmvs_emu_<tag>.so beside the default build"""
    LIB = os.path.join(HERE, "libdmvs_emu%s.so" % ("_" + tag if tag else ""))
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [ os.path.join(ROOT, "include", "dmvs.h"),
                   os.path.join(HERE, "hip", "hip_runtime.h")]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(HERE, os.path.basename(s) + (".emu_%s.o" % tag if tag else ".emu.o"))
        objs.append(o)
        procs.append(subprocess.Popen(
            ["g++", "-O2", "-std=c++17", "-fPIC", "-x", "c++", "-c", s, "-o", o,
             "-I", HERE, "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-Wno-unknown-pragmas", "-Wno-attributes"] + ["-D" + d for d in defines]))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipemu compile failed")
    subprocess.run(["g++", "-shared", "-o", LIB] + objs + ["-pthread"], check=True)
    return LIB


if __name__ == "__main__":
    print(build_emu(force=True))
