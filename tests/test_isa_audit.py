"""Static check of the built gfx950 code objects (no GPU needed): every loop that stages through LDS-DMA and synchronises with a
barrier waits for the DMA inside the loop (tools/isa_dma_audit.py; the round-4 stem kernel did not, and was intermittently wrong
on hardware while every emulation test passed)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_dma_audit as A  # noqa: E402

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
needs_rocm = pytest.mark.skipif(not (os.path.exists(HIPCC) and A.tools_present()),
                                reason="needs hipcc and the ROCm llvm tools (clang-offload-bundler, llvm-objdump)")


@needs_rocm
def test_every_lds_dma_loop_waits_for_its_dma():
    from diffmvs_amd.build import build_hip
    objdir = os.path.join(ROOT, "build", "obj")
    build_hip(force=not (os.path.isdir(objdir) and any(f.endswith(".o") for f in os.listdir(objdir))), verbose=False)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_dma_audit.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 finding(s)" in r.stdout


@needs_rocm
def test_the_audit_flags_a_loop_without_the_wait(tmp_path):
    """the checker itself: the same tile loop compiled with the DMA barrier degraded to a plain __syncthreads() is reported"""
    obj = tmp_path / "stem_nowait.o"
    csrc = os.path.join(ROOT, "diffmvs_amd", "csrc")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"), "-I", csrc,
                    "-DDMVS_DMA_BARRIER()=__syncthreads()", "-c", os.path.join(csrc, "stem.hip"), "-o", str(obj)], check=True, capture_output=True)
    findings = A.audit(A.disassemble(str(obj), str(tmp_path)), "stem_nowait.o")
    assert findings and all("featurenet_stem_kernel" in k for _, k, _ in findings), findings


def test_the_audit_wants_the_wait_between_the_dma_and_the_barrier():
    """the rule itself, on hand-written listings: a vmcnt(0) that sits AFTER the barrier (or in front of the DMA) does not publish the
    DMA's data, although it lies inside the loop's address range"""
    def listing(body):
        lines = ["0000000000001000 <k>:"]
        for i, (op, args) in enumerate(body):
            lines.append("\t%s %s // %012X: 00000000" % (op, args, 0x1000 + 4 * i))
        # backward branch to the top: simm16 = (target - (addr + 4)) / 4
        n = len(body)
        lines.append("\ts_cbranch_scc1 %d // %012X: 00000000" % ((-(n + 1)) & 0xFFFF, 0x1000 + 4 * n))
        return "\n".join(lines) + "\n"
    dma, bar, wait, work = ("global_load_lds_dwordx4", "v[0:1], off"), ("s_barrier", ""), ("s_waitcnt", "vmcnt(0)"), ("v_mfma_f32_16x16x4_f32", "a, b, c")
    good = listing([wait, bar, dma, work])                 # the DMA of trip i is waited for at the top of trip i + 1, then published
    late = listing([bar, wait, dma, work])                 # the wait sits behind the barrier: ds_reads after the barrier race the DMA
    early = listing([wait, dma, bar, work])                # the wait precedes the DMA it should cover
    assert A.audit(good, "x") == []
    for bad in (late, early):
        f = A.audit(bad, "x")
        assert len(f) == 1 and "no barrier in it is preceded" in f[0][2], f
