"""Static audit of the built gfx950 code objects: every loop that issues an LDS-DMA (global_load_lds_*) and synchronises with s_barrier
must wait for the DMA (s_waitcnt with vmcnt(0)) INSIDE the loop -- the round-4 non-determinism was a tile-walking loop whose only
vmcnt(0) sat in front of the loop (hipcc does not emit one for a workgroup-scope __syncthreads() on gfx950, and does not track the
LDS-DMA -> ds_read dependency here).

    python tools/isa_dma_audit.py            # disassembles build/obj/*.o (python -m diffmvs_amd.build first); exit code 1 on a finding

Heuristic, conservative: a "loop" is the address range of a backward branch.  A loop that contains a DMA and a barrier passes when at least
one of its barriers PUBLISHES the DMA: walking backwards from that barrier through the loop body in program (address) order, wrapping at the
loop top, a `s_waitcnt vmcnt(0)` comes before any LDS-DMA instruction.  (Round 5 accepted any vmcnt(0) inside the range -- also one that
sits after the barrier or serves an unrelated register dependency, the very accident dmvs_common.h describes; barriers that deliberately
leave a prefetch in flight, DMVS_LDS_BARRIER, are fine as long as the loop has one publishing barrier.)  Also reports kernels with a DMA but
no vmcnt(0) at all.  $HIPCC's directory / $DMVS_LLVM_BIN override the tool locations.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("DMVS_LLVM_BIN") or "/opt/rocm/lib/llvm/bin"


def tools_present():
    return all(os.path.exists(os.path.join(LLVM, t)) for t in ("clang-offload-bundler", "llvm-objdump"))


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
            in_dma, in_bar, in_wait = [d for d in dma if lo <= d <= hi], [b for b in bars if lo <= b <= hi], [w for w in waits if lo <= w <= hi]
            published = False
            for b in in_bar:      # nearest preceding DMA-or-wait of this barrier, cyclically inside [lo, hi]
                n = hi - lo + 1
                for step in range(1, n + 1):
                    j = lo + (b - lo - step) % n
                    if j in in_wait:
                        published = True
                        break
                    if j in in_dma:
                        break
                if published:
                    break
            loops.append((lo, hi, published, bool(in_dma) and bool(in_bar)))
        for lo, hi, has_wait, relevant in loops:
            if not relevant or has_wait:
                continue
            # (the structurizer's flow blocks show up as backward branches NESTED in the real tile loop: a range is reported only when no
            # enclosing backward range of the kernel waits either)
            if any(l2 <= lo <= h2 and w2 for l2, h2, w2, _ in loops):      # (their out-of-line tails may lie beyond the loop's own back edge)
                continue
            findings.append((src, kernel, "loop 0x%x..0x%x issues LDS-DMA and s_barrier, but no barrier in it is preceded by a vmcnt(0) that follows the DMA" % (insns[lo][0], insns[hi][0])))

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
