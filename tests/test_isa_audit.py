"""Static check of the built gfx950 code objects (no GPU needed): every loop that stages through LDS-DMA and synchronises with a
barrier waits for the DMA inside the loop (tools/isa_dma_audit.py; the round-4 stem kernel did not, and was intermittently wrong
on hardware while every emulation test passed)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_lds_dma_loop_waits_for_its_dma():
    from diffmvs_amd.build import build_hip
    objdir = os.path.join(ROOT, "build", "obj")
    build_hip(force=not (os.path.isdir(objdir) and any(f.endswith(".o") for f in os.listdir(objdir))), verbose=False)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_dma_audit.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 finding(s)" in r.stdout


def test_the_audit_flags_a_loop_without_the_wait(tmp_path):
    """the checker itself: the same tile loop compiled with the DMA barrier degraded to a plain __syncthreads() is reported"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_dma_audit as A
    obj = tmp_path / "stem_nowait.o"
    csrc = os.path.join(ROOT, "diffmvs_amd", "csrc")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"), "-I", csrc,
                    "-DDMVS_DMA_BARRIER()=__syncthreads()", "-c", os.path.join(csrc, "stem.hip"), "-o", str(obj)], check=True, capture_output=True)
    findings = A.audit(A.disassemble(str(obj), str(tmp_path)), "stem_nowait.o")
    assert findings and all("featurenet_stem_kernel" in k for _, k, _ in findings), findings
