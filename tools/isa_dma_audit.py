"""Static audit of the built gfx950 code objects: every loop that issues an LDS-DMA (global_load_lds_*) and synchronises with s_barrier
must wait for the DMA (s_waitcnt with vmcnt(0)) INSIDE the loop -- the round-4 non-determinism was a tile-walking loop whose only
vmcnt(0) sat in front of the loop (hipcc does not emit one for a workgroup-scope __syncthreads() on gfx950, and does not track the
LDS-DMA -> ds_read dependency here).

    python tools/isa_dma_audit.py            # disassembles build/obj/*.o (python -m diffmvs_amd.build first); exit code 1 on a finding

Heuristic, conservative: a "loop" is the address range of a backward branch; reports loops that contain a DMA and a barrier but no vmcnt(0).
Also reports kernels with a DMA but no vmcnt(0) at all.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(obj, tmp):
    name = os.path.basename(obj)
    fat, co = os.path.join(tmp, name + ".fat"), os.path.join(tmp, name + ".co")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
    subprocess.run([LLVM + "/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co,
                    "--unbundle"], check=True)
    return subprocess.run([LLVM + "/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout


INSN = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
LABEL = re.compile(r"^([0-9a-f]+) <(.+)>:")


def audit(text, src):
    findings, kernel, insns = [], None, []

    def flush():
        if kernel is None or not insns:
            return
        addr_idx = {a: i for i, (a, _, _) in enumerate(insns)}
        dma = [i for i, (_, op, _) in enumerate(insns) if op.startswith("global_load_lds") or (op.startswith("buffer_load") and " lds" in insns[i][2])]
        if not dma:
            return
        waits = [i for i, (_, op, args) in enumerate(insns) if op == "s_waitcnt" and "vmcnt(0)" in args]
        bars = [i for i, (_, op, _) in enumerate(insns) if op == "s_barrier"]
        if not waits:
            findings.append((src, kernel, "LDS-DMA but no s_waitcnt vmcnt(0) anywhere"))
            return
        loops = []
        for i, (a, op, args) in enumerate(insns):
            if not op.startswith("s_cbranch") and op != "s_branch":
                continue
            m2 = re.match(r"(\d+)", args)      # the simm16 operand: target = address + 4 + 4 * simm16
            if not m2:
                continue
            simm = int(m2.group(1))
            if simm >= 0x8000:
                simm -= 0x10000
            tgt = a + 4 + 4 * simm
            if tgt > a or tgt not in addr_idx:
                continue
            lo, hi = addr_idx[tgt], i
            has_wait = any(lo <= w <= hi for w in waits)
            loops.append((lo, hi, has_wait, any(lo <= d <= hi for d in dma) and any(lo <= b <= hi for b in bars)))
        for lo, hi, has_wait, relevant in loops:
            if not relevant or has_wait:
                continue
            # (the structurizer's flow blocks show up as backward branches NESTED in the real tile loop: a range is reported only when no
            # enclosing backward range of the kernel waits either)
            if any(l2 <= lo <= h2 and w2 for l2, h2, w2, _ in loops):      # (their out-of-line tails may lie beyond the loop's own back edge)
                continue
            findings.append((src, kernel, "loop 0x%x..0x%x issues LDS-DMA and s_barrier without a vmcnt(0) inside" % (insns[lo][0], insns[hi][0])))

    for line in text.splitlines():
        m = LABEL.match(line)
        if m:
            flush()
            kernel, insns = m.group(2), []
            continue
        m = INSN.match(line)
        if m and kernel is not None:
            insns.append((int(m.group(3), 16), m.group(1), m.group(2)))
    flush()
    return findings


def main():
    objdir = os.path.join(ROOT, "build", "obj")
    objs = sorted(os.path.join(objdir, f) for f in os.listdir(objdir) if f.endswith(".o"))
    allf = []
    with tempfile.TemporaryDirectory() as tmp:
        for o in objs:
            allf += audit(disassemble(o, tmp), os.path.basename(o))
    for src, k, what in allf:
        print(f"{src}: {k[:110]}: {what}")
    print(f"{len(allf)} finding(s) in {len(objs)} code objects")
    return 1 if allf else 0


if __name__ == "__main__":
    sys.exit(main())
