"""CPU: the C-ABI boundary.  Every function include/dmvs.h declares must be bound in
diffmvs_amd/_lib.py and exported by the built gfx950 library (no compute calls here: no GPU)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from diffmvs_amd import _lib


def header_functions():
    src = open(os.path.join(ROOT, "include", "dmvs.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(dmvs_\w+)\s*\(", src)))


def test_header_and_binding_agree():
    assert header_functions() == sorted(_lib.SIGNATURES)


def test_library_builds_and_exports_every_symbol():
    from diffmvs_amd.build import build_hip
    path = build_hip()
    dll = ctypes.CDLL(path)
    for name in header_functions():
        assert hasattr(dll, name), name
    assert dll.dmvs_abi_version() == _lib.ABI_VERSION
    _lib.Lib(path)


def test_descriptor_structs_match_header_field_order():
    src = open(os.path.join(ROOT, "include", "dmvs.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    for cname, cls in (("dmvs_conv2d_desc", _lib.Conv2dDesc), ("dmvs_conv3d_desc", _lib.Conv3dDesc),
                       ("dmvs_getcost_desc", _lib.GetCostDesc)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), src, flags=re.S).group(1)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            typ, names = decl.rsplit(" ", 1)[0], decl
            names = re.sub(r"^(const\s+)?(float|double|int32_t|void)\s*\*?", "", decl)
            for n in names.split(","):
                fields.append(n.replace("*", "").strip())
        got = [f[0].rstrip("_") for f in cls._fields_]
        assert got == fields, (cname, got, fields)


def test_abi_version_pins_the_descriptor_sizes():
    """a change of a descriptor's size or of an entry point's argument list must come with a DMVS_ABI_VERSION bump (ADVICE round 3:
    `arith` was appended under version 1): the sizes and arities of version 3 are pinned here, header and binding alike (version 3 = version 2 +
    dmvs_mask_upsample4_f32 and dmvs_getcost_desc.tune)"""
    src = open(os.path.join(ROOT, "include", "dmvs.h")).read()
    assert int(re.search(r"#define DMVS_ABI_VERSION (\d+)", src).group(1)) == _lib.ABI_VERSION == 4
    assert (ctypes.sizeof(_lib.Conv2dDesc), ctypes.sizeof(_lib.Conv3dDesc), ctypes.sizeof(_lib.GetCostDesc)) == (216, 104, 160)
    assert len(_lib.SIGNATURES["dmvs_featurenet_stem_f32"]) == 13 and len(_lib.SIGNATURES["dmvs_warp_corr_init_quad_f32"]) == 18
    assert "dmvs_conv3x3_pair16_f32" not in _lib.SIGNATURES and len(_lib.SIGNATURES["dmvs_mask_upsample4_f32"]) == 15


def test_library_reads_no_environment_variable():
    """include/dmvs.h: no global state, no environment variables -- knobs are `tune` arguments, read from the environment (if at
    all) by the Python layer"""
    csrc = os.path.join(ROOT, "diffmvs_amd", "csrc")
    for base, _, files in os.walk(csrc):          # (csrc/probe: the bench-only probe library obeys the same rule)
        for f in files:
            assert "getenv" not in open(os.path.join(base, f)).read(), f


def test_probe_library_exports_its_header():
    """include/dmvs_probe.h <-> diffmvs_amd/libdmvs_probe.so: the bench-only measurement kernels (GetCost ceiling probe, random line gather).
    Built by the same build step; never loaded by the depth-estimation path (nothing under diffmvs_amd/ or models/ names it)."""
    from diffmvs_amd.build import PROBE_LIB, build_hip
    build_hip()
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "dmvs_probe.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\bint\s+(dmvs_probe_\w+)\s*\(", src)))
    assert names == ["dmvs_probe_abi_version", "dmvs_probe_getcost_loads_f32", "dmvs_probe_getcost_pair_loads_f32", "dmvs_probe_random_line_gather"]
    dll = ctypes.CDLL(PROBE_LIB)
    for n in names:
        assert hasattr(dll, n), n
    assert dll.dmvs_probe_abi_version() == 2
    # argument validation happens before any launch: NULL descriptor / table -> DMVS_EINVAL
    dll.dmvs_probe_getcost_loads_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    assert dll.dmvs_probe_getcost_loads_f32(None, None) == -22
    dll.dmvs_probe_getcost_pair_loads_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    assert dll.dmvs_probe_getcost_pair_loads_f32(None, None) == -22
    dll.dmvs_probe_random_line_gather.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                   ctypes.c_uint32, ctypes.c_void_p]
    assert dll.dmvs_probe_random_line_gather(None, 1024, 16, 8, 0, 0, 1, None) == -22
    assert dll.dmvs_probe_random_line_gather(ctypes.c_void_p(4096), 1024, 1024, 8, 0, 0, 1, None) == -22      # "once": more requests than lines
    for pkg in ("diffmvs_amd", "models"):
        for base, _, files in os.walk(os.path.join(ROOT, pkg)):
            for f in files:
                if f.endswith(".py") and f != "build.py":
                    assert "libdmvs_probe" not in open(os.path.join(base, f)).read(), f


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(_lib.DmvsError):
        _lib.Lib(str(tmp_path / "nope.so"))


def test_invalid_descriptors_are_rejected_before_any_launch():
    """Argument validation happens on the host before the first HIP call, so it can be exercised without a GPU: NULL
    operands, unsupported shapes and sizes beyond the kernels' 32-bit addressing all return DMVS_EINVAL (-22).  The
    pointers below are never dereferenced."""
    from diffmvs_amd.build import build_hip
    lib = _lib.Lib(build_hip())
    p = ctypes.c_void_p(4096)

    def conv2d(**kw):
        d = _lib.Conv2dDesc(in0=p, weight=p, out=p, B=1, c0=16, c1=0, Hin=64, Win=64, Hout=64, Wout=64, cout=16, cout_pad=16,
                            kh=3, kw=3, stride=1, pad_h=1, pad_w=1, in_mode=0, act=0, res_mode=0, res_after_act=0,
                            out_layout=0, out_cstride=16, out_coffset=0, gn_groups=0, post_scale=1.0)
        for k, v in kw.items():
            setattr(d, k, v)
        return lib.dll.dmvs_conv2d_f32(ctypes.byref(d), None)

    assert conv2d(in0=None) == -22
    assert conv2d(out=None) == -22
    assert conv2d(weight=ctypes.c_void_p(4100)) == -22             # the weight slab is staged in 16-byte pieces
    assert conv2d(cout_pad=12) == -22                              # not a multiple of 8
    assert conv2d(Hout=63) == -22                                  # inconsistent with Hin / pad / stride
    assert conv2d(c1=8) == -22                                     # concat without a second tensor
    assert conv2d(kh=4, kw=4, pad_h=1, pad_w=1, Hout=63, Wout=63) == -22      # no 4x4 instantiation
    assert conv2d(Hin=8192, Win=8192, Hout=8192, Wout=8192) == -22            # plane >= 2^24 pixels
    assert conv2d(c0=4096, Hin=1024, Win=1024, Hout=1024, Wout=1024) == -22   # item >= 2^31 elements

    def conv3d(**kw):
        d = _lib.Conv3dDesc(in_=p, weight=p, out=p, B=1, cin=8, cout=8, cout_pad=8, Din=8, Hin=8, Win=8, Dout=8, Hout=8, Wout=8,
                            stride=1, transposed=0, act=0)
        for k, v in kw.items():
            setattr(d, k, v)
        return lib.dll.dmvs_conv3d_f32(ctypes.byref(d), None)

    assert conv3d(weight=None) == -22
    assert conv3d(weight=ctypes.c_void_p(4100)) == -22
    assert conv3d(stride=3) == -22
    assert conv3d(Dout=7) == -22
    assert conv3d(transposed=1, stride=1) == -22
    assert conv3d(cin=64, Din=512, Hin=512, Win=512, Dout=512, Hout=512, Wout=512) == -22     # item >= 2^31 elements

    def getcost(**kw):
        d = _lib.GetCostDesc(ref=p, src=p, rt=p, inv_depth=p, view_w=p, disp_min=p, disp_max=p, out_cost=p, out_samples=p,
                             worklist=p, B=1, S=2, C=32, G=4, n=6, H=32, W=32, vw_shift=1, cost_cstride=24, cost_coffset=0,
                             samp_cstride=6, samp_coffset=0, interval=0.04, min_radius=0.125, max_radius=8.0)
        for k, v in kw.items():
            setattr(d, k, v)
        return lib.dll.dmvs_getcost_quad_f32(ctypes.byref(d), None)

    assert getcost(n=5) == -22                                     # the reference uses 4 or 6 hypotheses
    assert getcost(C=24) == -22
    assert getcost(G=3) == -22
    assert getcost(ref=None) == -22
    assert getcost(feat_dtype=4) == -22                            # DMVS_DTYPE_* 0..3
    assert getcost(H=8192, W=8192) == -22                          # plane >= 2^24 texels: 24-bit row multiplies
    assert lib.dll.dmvs_conv2d_f32(None, None) == -22 and lib.dll.dmvs_conv3d_f32(None, None) == -22


def test_integration_md_stub_matches_the_binding():
    """the ctypes stub INTEGRATION.md shows a maintainer must describe the same struct as diffmvs_amd/_lib.py"""
    src = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"class GetCostDesc\(C.Structure\):.*?\n\n", src, flags=re.S).group(0)
    ns = {"C": ctypes}
    exec(code, ns)
    assert [n for n, _ in ns["GetCostDesc"]._fields_] == [n for n, _ in _lib.GetCostDesc._fields_]
    assert ctypes.sizeof(ns["GetCostDesc"]) == ctypes.sizeof(_lib.GetCostDesc)
