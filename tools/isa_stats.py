"""Per-kernel ISA statistics of one csrc/*.hip file (no GPU needed: hipcc cross-compiles gfx950):

    python tools/isa_stats.py diffmvs_amd/csrc/conv2d_k33.hip [kernel-name-substring] [--diff old.json] [--save new.json]

For every kernel: instructions, MFMAs, LDS-DMA instructions, VGPRs, occupancy (waves/SIMD), SGPR-spill lanes read /
written (v_readlane / v_writelane), scratch bytes, LDS bytes, and `vmcnt(0)` waits.  The convolution kernels are
sensitive to scalar-register pressure and to waits the compiler inserts between an LDS-DMA and the first ds_read
(DESIGN.md section 4), so these numbers are compared across commits before a change goes to the GPU."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_stats(hip_file):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                        "-I", os.path.join(ROOT, "diffmvs_amd", "csrc"), "-S", "--cuda-device-only", "-o", out, hip_file],
                       check=True, stderr=subprocess.DEVNULL)
        src = open(out).read().split("\n")
    stats, k = {}, 0
    while k < len(src):
        m = re.match(r"^(_Z\S+):", src[k])
        if m and "@" + m.group(1) in src[k]:
            j = k
            while j < len(src) and not src[j].startswith(".Lfunc_end"):
                j += 1
            ins = [x.strip() for x in src[k:j] if x.startswith("\t") and not x.strip().startswith((".", ";"))]
            meta = " ".join(src[j:j + 200])

            def field(name):
                mm = re.search(r"; %s: (\d+)" % name, meta)
                return int(mm.group(1)) if mm else None
            if field("NumVgprs") is not None:
                name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
                name = re.sub(r"\(.*", "", name.replace("void ", "").replace("(anonymous namespace)::", ""))
                stats[name] = {"instr": len(ins), "mfma": sum("v_mfma" in x for x in ins),
                               "lds_dma": sum("global_load_lds" in x for x in ins), "vgpr": field("NumVgprs"),
                               "occupancy": field("Occupancy"), "sgpr_spill_rd": sum(x.startswith("v_readlane") for x in ins),
                               "sgpr_spill_wr": sum(x.startswith("v_writelane") for x in ins), "scratch": field("ScratchSize"),
                               "lds_bytes": (re.search(r"; LDSByteSize: (\d+)", meta) or [None, None])[1],
                               "vmcnt0": sum("vmcnt(0)" in x for x in ins)}
            k = j
        k += 1
    return stats


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    stats = kernel_stats(args[0])
    pat = args[1] if len(args) > 1 else ""
    old = json.load(open(sys.argv[sys.argv.index("--diff") + 1])) if "--diff" in sys.argv else None
    for name in sorted(stats):
        if pat in name:
            line = "%-62s " % name[:int(os.environ.get("ISA_NAME_WIDTH", "62"))] + " ".join("%s=%s" % kv for kv in stats[name].items())
            if old and name in old:
                ch = {k: (old[name][k], v) for k, v in stats[name].items() if old[name].get(k) != v}
                line += "   CHANGED " + str(ch) if ch else "   (same)"
            print(line)
    if "--save" in sys.argv:
        json.dump(stats, open(sys.argv[sys.argv.index("--save") + 1], "w"), indent=0)


if __name__ == "__main__":
    main()
