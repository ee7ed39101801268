"""ctypes binding of the C ABI declared in include/dmvs.h.

The product path loads exactly one thing: diffmvs_amd/libdmvs_hip.so, the gfx950 code
object built by diffmvs_amd.build.  If it is missing, or no HIP device is present, loading
fails loudly -- there is no CPU or PyTorch fallback behind this module.
"""
from __future__ import annotations

import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
# DMVS_LIB: another build of the same sources for A/B runs; default: the in-tree product library
HIP_LIB_PATH = os.environ.get("DMVS_LIB") or os.path.join(PKG, "libdmvs_hip.so")

ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH, ACT_SILU = range(5)
IN_PLAIN, IN_UPSAMPLE2, IN_UNSHUFFLE2, IN_ZEROINSERT2 = range(4)
LAYOUT_NCHW, LAYOUT_NHWC, LAYOUT_NHWC_BF16, LAYOUT_NHWC_F16 = 0, 1, 2, 3
DTYPE_F32, DTYPE_BF16, DTYPE_F16, DTYPE_F32_PLAIN = 0, 1, 2, 3
ARITH_F32, ARITH_BF16, ARITH_SPLIT = 0, 1, 2
EW_DEPTH_TO_DISP, EW_DISP_TO_DEPTH = 0, 1

_P = C.c_void_p
_I = C.c_int32
_F = C.c_float


class Conv2dDesc(C.Structure):
    _fields_ = [
        ("in0", _P), ("in1", _P), ("mul0", _P), ("weight", _P), ("scale", _P), ("shift", _P),
        ("residual", _P), ("gru_z", _P), ("gru_h", _P), ("out", _P), ("gn_stats", _P), ("out_mul", _P),
        ("B", _I), ("c0", _I), ("c1", _I), ("Hin", _I), ("Win", _I), ("Hout", _I), ("Wout", _I),
        ("cout", _I), ("cout_pad", _I), ("kh", _I), ("kw", _I), ("stride", _I), ("pad_h", _I), ("pad_w", _I),
        ("in_mode", _I), ("act", _I), ("res_mode", _I), ("res_after_act", _I),
        ("out_layout", _I), ("out_cstride", _I), ("out_coffset", _I), ("gn_groups", _I), ("post_scale", _F),
        ("gate_cstride", _I), ("arith", _I), ("tune", _I), ("out_mul_c0", _I), ("in0_cstride", _I), ("weight_split", _P),
    ]


class Conv3dDesc(C.Structure):
    _fields_ = [
        ("in_", _P), ("weight", _P), ("scale", _P), ("shift", _P), ("residual", _P), ("out", _P),
        ("B", _I), ("cin", _I), ("cout", _I), ("cout_pad", _I),
        ("Din", _I), ("Hin", _I), ("Win", _I), ("Dout", _I), ("Hout", _I), ("Wout", _I),
        ("stride", _I), ("transposed", _I), ("act", _I), ("tune", _I),
    ]


class GetCostDesc(C.Structure):
    _fields_ = [
        ("ref", _P), ("src", _P), ("rt", _P), ("inv_depth", _P), ("confidence", _P), ("view_w", _P),
        ("disp_min", _P), ("disp_max", _P), ("out_cost", _P), ("out_samples", _P), ("worklist", _P),
        ("B", _I), ("S", _I), ("C", _I), ("G", _I), ("n", _I), ("H", _I), ("W", _I), ("vw_shift", _I),
        ("cost_cstride", _I), ("cost_coffset", _I), ("samp_cstride", _I), ("samp_coffset", _I),
        ("interval", _F), ("min_radius", _F), ("max_radius", _F), ("feat_dtype", _I), ("tune", _I),
    ]


# name -> argtypes; every function returns int (0 = ok).  Must list every symbol of dmvs.h:
# tests/test_abi.py checks header <-> this table <-> the built library.
SIGNATURES = {
    "dmvs_abi_version": [],
    "dmvs_conv2d_f32": [C.POINTER(Conv2dDesc), _P],
    "dmvs_featurenet_stem_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "dmvs_conv2d_wgrad_workspace_f32": [C.POINTER(Conv2dDesc), C.POINTER(C.c_int64)],
    "dmvs_conv2d_wgrad_f32": [C.POINTER(Conv2dDesc), _P, _P, _P, _P, C.c_int64, _P],
    "dmvs_conv3d_f32": [C.POINTER(Conv3dDesc), _P],
    "dmvs_conv3d_wgrad_workspace_f32": [C.POINTER(Conv3dDesc), C.POINTER(C.c_int64)],
    "dmvs_conv3d_wgrad_f32": [C.POINTER(Conv3dDesc), _P, _P, _P, _P, C.c_int64, _P],
    "dmvs_compose_proj_f32": [_P, _P, _I, _I, _P],
    "dmvs_warp_volume_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "dmvs_getcost_quad_f32": [C.POINTER(GetCostDesc), _P],
    "dmvs_warp_corr_init_quad_f32": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "dmvs_warp_corr_init_bwd_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "dmvs_getcost_bwd_f32": [C.POINTER(GetCostDesc), _P, _P, _P, _P],
    "dmvs_view_aggregate_bwd_f32": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "dmvs_view_aggregate_f32": [_P, _P, _P, _I, _I, _I, _I, _P],
    "dmvs_sigmoid_max_d_f32": [_P, _P, _I, _I, _I, _P],
    "dmvs_depth_regress_f32": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "dmvs_convex_upsample_f32": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "dmvs_mask_upsample4_f32": [_P, _P, _P, _F, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dmvs_groupnorm_silu_f32": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "dmvs_groupnorm_apply_f32": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "dmvs_delta_update_f32": [_P, _P, _P, _F, _P, _P, _P, _I, _I, _I, _I, _P],
    "dmvs_depth_convert_f32": [_P, _P, _P, _P, _I, _I, _I, _P],
    "dmvs_act_slice_f32": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "dmvs_upsample_nearest_f32": [_P, _P, _I, _I, _I, _I, _P],
    "dmvs_nchw_to_nhwc_f32": [_P, _P, _I, _I, _I, _P],
    "dmvs_groupnorm_silu_bwd_workspace_f32": [_I, _I, _I, C.POINTER(C.c_int64)],
    "dmvs_groupnorm_silu_bwd_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int64, _I, _I, _I, _I, _F, _P],
    "dmvs_batchnorm_workspace_f32": [_I, _I, _I, _I, C.POINTER(C.c_int64)],
    "dmvs_batchnorm_train_fwd_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int64, _I, _I, _I, _I, _I, _F, _F, _I, _P],
    "dmvs_batchnorm_train_bwd_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int64, _I, _I, _I, _I, _I, _I, _P],
    "dmvs_geo_consistency_f32": [_P, _P, _P, _P, _P, _I, _I, _F, _F, _P, _P, _I, _I, _I, _I, _I, _P],
    "dmvs_sumsq_f32": [_P, C.c_int64, _P, _P],
    "dmvs_adamw_step_f32": [_P, _P, _P, _P, C.c_int64, _F, _F, _F, _F, _F, _I, _F, _P, _F, _P],
}
ABI_VERSION = 4
# dmvs.h: DMVS_TUNE_* (dmvs_conv2d_desc.tune, dmvs_featurenet_stem_f32), DMVS_TUNE3D_* (dmvs_conv3d_desc.tune), DMVS_TUNE_SWEEP_GLOBAL
TUNE_NO_WALK, TUNE_PIECES4, TUNE_NO_LEAN, TUNE_1X1_TILED, TUNE_NO_TALL, TUNE_TALL = 0x4, 0x8, 0x100, 0x200, 0x400, 0x800
TUNE3D_PIECES4, TUNE3D_S2_DIRECT, TUNE3D_NO_PAIR = 0x1, 0x2, 0x4
TUNE_SWEEP_GLOBAL = 0x1
TUNE_WGRAD_ACCUMULATE = 0x1000
TUNE_SPLIT_ALL = 0x20000
TUNE_STEM_EXACT = 0x40000


def tune_xcd_group(n: int) -> int:      # DMVS_TUNE_XCD_GROUP(n)
    return (int(n) & 7) << 14


def tune3d_xcd_group(n: int) -> int:      # DMVS_TUNE3D_XCD_GROUP(n)
    return (int(n) & 7) << 4
TUNE_BWD_INTERLEAVED, BWD_GATHER_INTERLEAVED = 0x1, 2


def tune_tile_wx(n: int) -> int:
    return n & 3


def tune_tile_mt(n: int) -> int:
    return (n & 7) << 4


class DmvsError(RuntimeError):
    pass


class Lib:
    """A loaded C-ABI library; every call raises DmvsError on a non-zero return."""

    def __init__(self, path: str):
        if not os.path.exists(path):
            raise DmvsError(f"{path} not found: build it with `python -m diffmvs_amd.build`")
        self.path = path
        self.dll = C.CDLL(path)
        self.dll.dmvs_abi_version.restype = C.c_int
        v = self.dll.dmvs_abi_version()          # FIRST: a library of another ABI version may lack / re-type the symbols below
        if v != ABI_VERSION:
            raise DmvsError(f"{path}: ABI version {v}, binding expects {ABI_VERSION} (rebuild: python -m diffmvs_amd.build)")
        for name, argtypes in SIGNATURES.items():
            fn = getattr(self.dll, name)     # AttributeError if the library lacks a declared symbol
            fn.argtypes = argtypes
            fn.restype = C.c_int

    def call(self, name: str, *args):
        rc = getattr(self.dll, name)(*args)
        if rc != 0:
            raise DmvsError(f"{name} failed with code {rc}" + (" (DMVS_EINVAL: unsupported descriptor)" if rc == -22 else " (hipError_t)"))


_hip_lib = None


def hip_lib() -> Lib:
    """The product library.  No fallback: raises if the .so is missing."""
    global _hip_lib
    if _hip_lib is None:
        _hip_lib = Lib(HIP_LIB_PATH)
    return _hip_lib
