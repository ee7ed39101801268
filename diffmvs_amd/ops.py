"""Tensor-level wrappers around the C ABI (include/dmvs.h).

torch is used here for device memory and the stream handle only; every arithmetic
operation on the hot path is a kernel of libdmvs_hip.so.  `Ops` is bound to one library and
one device; the product constructs it with `Ops.for_device(cuda_device)` which loads the
gfx950 library and refuses anything else.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from ._lib import (ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SILU, ACT_TANH, ARITH_BF16, ARITH_F32, ARITH_SPLIT, DTYPE_BF16, DTYPE_F16, DTYPE_F32, IN_PLAIN,  # noqa: F401
                   IN_UNSHUFFLE2, IN_UPSAMPLE2, IN_ZEROINSERT2, LAYOUT_NCHW, LAYOUT_NHWC, LAYOUT_NHWC_BF16, LAYOUT_NHWC_F16)

# matrix arithmetic of the multi-tap 2-D convolutions: "fp32" (exact, the reference's) | "bf16" (bf16 products, fp32 accumulation)
CONV_ARITH = {"fp32": ARITH_F32, "bf16": ARITH_BF16, "split": ARITH_SPLIT}
# reduced-precision FEATURE storage (BASELINE.json's bf16 / fp16 configurations): torch dtype <-> C-ABI codes
FEATURE_DTYPES = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}
_DTYPE_CODE = {torch.float32: DTYPE_F32, torch.bfloat16: DTYPE_BF16, torch.float16: DTYPE_F16}
_NHWC_LAYOUT = {torch.float32: LAYOUT_NHWC, torch.bfloat16: LAYOUT_NHWC_BF16, torch.float16: LAYOUT_NHWC_F16}


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


@dataclass
class PackedConv:
    """Weights in kernel layout [cin][k...][cout_pad] + folded per-channel epilogue."""
    weight: torch.Tensor
    scale: Optional[torch.Tensor]
    shift: Optional[torch.Tensor]
    cin: int
    cout: int
    cout_pad: int
    k: tuple
    stride: int = 1
    pad: tuple = (0, 0)
    transposed: bool = False
    wsplit: Optional[torch.Tensor] = None      # (conv2d, ARITH_SPLIT) the weights as bf16 triples in the matrix cores' operand order: split_weights()


def split_weights(pc: "PackedConv") -> torch.Tensor:
    """dmvs_conv2d_desc.weight_split (include/dmvs.h): [ceil(cin/8)][ceil(T/4)][3 planes hi, mid, lo][4][cout_pad][8] bf16, element
    (c, g, p, q, co, j) = part p of the weight of input channel 8c + j, tap 4g + q, output channel co; hi = bf16(w), mid = bf16(w - hi),
    lo = bf16(w - hi - mid), round to nearest even; zero beyond cin / the taps.  Built once per packed layer, cached on it."""
    if pc.wsplit is None:
        kh, kw = pc.k
        T, cp = kh * kw, pc.cout_pad
        NG, NC = (T + 3) // 4, (pc.cin + 7) // 8
        w = torch.zeros(NC * 8, NG * 4, cp, dtype=torch.float32, device=pc.weight.device)
        w[:pc.cin, :T] = pc.weight.reshape(pc.cin, T, cp)
        hi = w.to(torch.bfloat16)
        r1 = w - hi.float()
        mid = r1.to(torch.bfloat16)
        lo = (r1 - mid.float()).to(torch.bfloat16)
        planes = torch.stack([hi, mid, lo]).view(3, NC, 8, NG, 4, cp)          # [p, c, j, g, q, co]
        pc.wsplit = planes.permute(1, 3, 0, 4, 5, 2).contiguous()              # [c, g, p, q, co, j]
    return pc.wsplit


def _pad_cout(cout: int) -> int:
    return 8 if cout <= 8 else _round_up(cout, 16)


def fold_bn(bn: dict, eps: float = 1e-5):
    """eval BatchNorm -> (scale, shift).  reference models/module.py:46,90 (torch default eps)."""
    scale = bn["weight"] * torch.rsqrt(bn["running_var"] + eps)
    shift = bn["bias"] - bn["running_mean"] * scale
    return scale, shift


def pack_conv2d(w: torch.Tensor, bias=None, bn: Optional[dict] = None, stride=1, pad=(0, 0),
                standardize=False) -> PackedConv:
    """w: [cout, cin, kh, kw] (torch layout).  standardize: WeightStandardizedConv2d,
    reference models/update.py:86-94 (fp32 eps 1e-5) applied once at pack time."""
    w = w.detach().float()
    if standardize:
        mean = w.mean(dim=(1, 2, 3), keepdim=True)
        var = w.var(dim=(1, 2, 3), unbiased=False, keepdim=True)
        w = (w - mean) * torch.rsqrt(var + 1e-5)
    cout, cin, kh, kw = w.shape
    cp = _pad_cout(cout)
    if cp == cout:                  # the usual case: no padding columns, one permuting copy
        wk = w.permute(1, 2, 3, 0).contiguous()
    else:
        wk = torch.zeros(cin, kh, kw, cp, dtype=torch.float32, device=w.device)
        wk[..., :cout] = w.permute(1, 2, 3, 0)
    scale = shift = None
    if bn is not None:
        scale, shift = fold_bn(bn)
        if bias is not None:
            shift = shift + bias.detach().float() * scale
    elif bias is not None:
        shift = bias.detach().float()
    pad = (pad, pad) if isinstance(pad, int) else tuple(pad)
    return PackedConv(wk.contiguous(), None if scale is None else scale.contiguous().float(),
                      None if shift is None else shift.contiguous().float(), cin, cout, cp, (kh, kw), stride, pad)


def pack_conv3d(w: torch.Tensor, bias=None, bn: Optional[dict] = None, stride=1, transposed=False) -> PackedConv:
    """w: Conv3d [cout,cin,3,3,3] or ConvTranspose3d [cin,cout,3,3,3] (reference module.py:88,130)."""
    w = w.detach().float()
    if transposed:
        cin, cout = w.shape[0], w.shape[1]
        wk_src = w.permute(0, 2, 3, 4, 1)          # [cin,kd,kh,kw,cout]
    else:
        cout, cin = w.shape[0], w.shape[1]
        wk_src = w.permute(1, 2, 3, 4, 0)
    cp = _pad_cout(cout)
    if cp == cout:
        wk = wk_src.reshape(cin, 27, cout).contiguous()
    else:
        wk = torch.zeros(cin, 27, cp, dtype=torch.float32, device=w.device)
        wk[..., :cout] = wk_src.reshape(cin, 27, cout)
    scale = shift = None
    if bn is not None:
        scale, shift = fold_bn(bn)
    elif bias is not None:
        shift = bias.detach().float()
    return PackedConv(wk.contiguous(), None if scale is None else scale.contiguous(),
                      None if shift is None else shift.contiguous(), cin, cout, cp, (3, 3, 3), stride, (1, 1),
                      transposed)


def g4_channels(C: int, dtype=torch.float32) -> torch.Tensor:
    """Channel order of the GROUP-INTERLEAVED channel-last feature layout "NHWC-g4" the quad-per-pixel warp kernels read
    (include/dmvs.h): position p of a texel holds channel  c(p) = ((p // 4) % 4) * (C // 4) + (p // 16) * 4 + p % 4,
    i.e. x_g4 = x_nhwc[..., g4_channels(C)].  The engine folds this permutation into the weights of FeatureNet's output
    convolutions; tests and the module-level GetCost use it on plain tensors.  16-bit features are read in plain NHWC
    order (a group's channels already form one 8 / 16 / 24-byte run per lane): identity."""
    p = torch.arange(C)
    if dtype != torch.float32:
        return p
    return ((p // 4) % 4) * (C // 4) + (p // 16) * 4 + p % 4


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


MAX_WINDOW_VIEWS = 16      # include/dmvs.h: DMVS_GETCOST_MAX_WINDOW_VIEWS

# Kernels that write parameters / buffers through raw pointers (dmvs_adamw_step_f32, the running statistics of
# dmvs_batchnorm_train_fwd_f32) do not bump torch's per-tensor version counters, so every cache of packed weights
# (models.module.HipModule.packed, CasDiffMVS.engine) also keys on this process-wide generation.
_weights_generation = [0]


def weights_generation() -> int:
    return _weights_generation[0]


def bump_weights_generation() -> None:
    _weights_generation[0] += 1


BWD_INTERLEAVED_DEFAULT = 1      # round-6 A/B (profiles/r6_bwd_interleave_ab.json): getcost_bwd 33.3 -> 9.9 ms per cfg4 step, 31.3 -> 40.3 samples/s


def _env_tune():
    """A/B knobs of the kernels' `tune` arguments (include/dmvs.h), read ONCE per binding in the Python layer -- the C library
    itself reads no environment variable -- and Ops.for_device keeps ONE binding per (library, device) for the life of the process: changing
    a variable after the first forward has no effect (A/B tools pass tune= per call, mutate ops.tune, or construct a fresh Ops).
    All default to 0 = the library's measured-best path:
      DMVS_CONV_WX=1|2, DMVS_CONV_MT=1|2|4, DMVS_CONV_WALK=0, DMVS_CONV_LEAN=0, DMVS_CONV_V16=0, DMVS_CONV1X1_PX4=0, DMVS_CONV_TALL=0|1, DMVS_CONV_XCD=1..7   (dmvs_conv2d_desc.tune)
      DMVS_CONV3D_V16=0, DMVS_CONV3D_S2=direct, DMVS_CONV3D_PAIR=0, DMVS_CONV3D_XCD=1..4   (dmvs_conv3d_desc.tune)      DMVS_STEM_V16=0, DMVS_STEM_XCD=1..4, DMVS_STEM_EXACT=1      DMVS_PLANE_SWEEP=quad"""
    e = os.environ.get
    t2 = _lib.tune_tile_wx(int(e("DMVS_CONV_WX", "0"))) | _lib.tune_tile_mt(int(e("DMVS_CONV_MT", "0")))
    t2 |= _lib.TUNE_NO_WALK if e("DMVS_CONV_WALK") == "0" else 0
    t2 |= _lib.TUNE_PIECES4 if e("DMVS_CONV_V16") == "0" else 0
    t2 |= _lib.TUNE_NO_LEAN if e("DMVS_CONV_LEAN") == "0" else 0
    t2 |= _lib.TUNE_1X1_TILED if e("DMVS_CONV1X1_PX4") == "0" else 0
    t2 |= {"0": _lib.TUNE_NO_TALL, "1": _lib.TUNE_TALL}.get(e("DMVS_CONV_TALL"), 0)
    t2 |= _lib.TUNE_SPLIT_ALL if e("DMVS_CONV_SPLIT_ALL") == "1" else 0
    t2 |= _lib.tune_xcd_group(int(e("DMVS_CONV_XCD", "0")))      # round 6: which tiles share an XCD's L2 (include/dmvs.h DMVS_TUNE_XCD_GROUP)
    t3 = (_lib.TUNE3D_PIECES4 if e("DMVS_CONV3D_V16") == "0" else 0) | (_lib.TUNE3D_S2_DIRECT if e("DMVS_CONV3D_S2") == "direct" else 0)
    t3 |= _lib.TUNE3D_NO_PAIR if e("DMVS_CONV3D_PAIR") == "0" else 0
    t3 |= _lib.tune3d_xcd_group(int(e("DMVS_CONV3D_XCD", "0")))
    return {"conv2d": t2, "conv3d": t3, "stem": (_lib.TUNE_PIECES4 if e("DMVS_STEM_V16") == "0" else 0) | _lib.tune_xcd_group(int(e("DMVS_STEM_XCD", "0"))) | (_lib.TUNE_STEM_EXACT if e("DMVS_STEM_EXACT") == "1" else 0),
            "sweep": _lib.TUNE_SWEEP_GLOBAL if e("DMVS_PLANE_SWEEP") == "quad" else 0,
            # training: GetCost backward through the per-pixel gather kernel only (no tile pre-pass / LDS-window worklist): DMVS_GETCOST_BWD=gather
            "bwd_gather": e("DMVS_GETCOST_BWD") == "gather",
            # training: channel -> lane mapping of the per-pixel scatter kernels (include/dmvs.h DMVS_TUNE_BWD_INTERLEAVED): DMVS_BWD_INTERLEAVED=0|1
            "bwd_il": e("DMVS_BWD_INTERLEAVED", "%d" % BWD_INTERLEAVED_DEFAULT) == "1"}


class Ops:
    _bindings = {}      # (library path, device) -> the one Ops of the process for it (for_device)

    def __init__(self, lib: _lib.Lib, device):
        self.lib = lib
        self.device = torch.device(device)
        self.tune = _env_tune()
        # optional per-entry-point HIP-event timing on the launch stream (bench.py roofline leg):
        # {"dmvs_getcost_f32": [(start_event, end_event), ...]}
        self.timers = None
        self.conv_arith = ARITH_F32      # matrix arithmetic of the multi-tap 2-D convolutions (with_conv_arith)
        # debugging aid (tools/determinism.py, tests): a float every freshly allocated kernel output is filled with before the launch, so
        # that a kernel that leaves part of its output unwritten -- or reads scratch it never wrote -- shows up as NaN instead of whatever the
        # allocator handed back.  None (the product default): plain torch.empty
        self.debug_fill = None
        # measurement hook (bench.py's untimed probe legs): called as hook(descriptor, tensors) right BEFORE every GetCost launch, on the
        # launch stream.  None in the product.
        self.getcost_hook = None

    def with_conv_arith(self, arith: int) -> "Ops":
        """a binding of the same library whose conv2d() computes in `arith` (ARITH_F32 | ARITH_BF16) by default"""
        import copy
        o = copy.copy(self)
        o.conv_arith = arith
        return o

    def _call(self, name, *args):
        if self.timers is not None and name in self.timers:
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            self.lib.call(name, *args)
            en.record()
            self.timers[name].append((st, en))
        else:
            self.lib.call(name, *args)

    @classmethod
    def for_device(cls, device) -> "Ops":
        device = torch.device(device)
        if device.type != "cuda":
            raise _lib.DmvsError(
                f"diffmvs_amd runs on MI355X only (got device '{device}'); there is no CPU path. "
                "Move the model and inputs to a HIP device.")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        lib = _lib.hip_lib()
        # ONE binding per (library, device): the engine, the training forward and the Trainer then share its per-binding state
        # (timers, conv_arith, the GetCost path probe) instead of each building a throw-away Ops per call
        key = (lib.path, device.index)
        o = cls._bindings.get(key)
        if o is None:
            o = cls._bindings[key] = cls(lib, device)
        return o

    # ------------------------------------------------------------------ plumbing
    def stream(self):
        if self.device.type == "cuda":
            return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return None

    def empty(self, *shape, dtype=torch.float32):
        t = torch.empty(*shape, dtype=dtype, device=self.device)
        if self.debug_fill is not None and t.is_floating_point():      # tools/determinism.py: every kernel output starts as NaN / garbage
            t.fill_(self.debug_fill)
        return t

    def empty_like(self, x):
        t = torch.empty_like(x)
        if self.debug_fill is not None and t.is_floating_point():
            t.fill_(self.debug_fill)
        return t

    def _chk_feat(self, *ts):
        dt = ts[0].dtype
        for t in ts:
            if t.dtype not in _DTYPE_CODE or t.dtype != dt or not t.is_contiguous() or t.device != self.device:
                raise _lib.DmvsError(f"expected contiguous fp32 / bf16 / fp16 feature tensors of one dtype on {self.device}, got "
                                     f"{t.dtype} contiguous={t.is_contiguous()} on {t.device}")
        return _DTYPE_CODE[dt]

    def _chk(self, *ts):
        for t in ts:
            if t is None:
                continue
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
                raise _lib.DmvsError(f"expected contiguous fp32 tensor on {self.device}, got {t.dtype} "
                                     f"contiguous={t.is_contiguous()} on {t.device}")

    # ------------------------------------------------------------------ conv2d
    def conv2d(self, pc: PackedConv, x0, x1=None, *, mul0=None, in_mode=IN_PLAIN, act=ACT_NONE, residual=None,
               res_mode=IN_PLAIN, res_after_act=False, post_scale=1.0, gru_z=None, gru_h=None, out=None,
               out_layout=LAYOUT_NCHW, out_cstride=None, out_coffset=0, gn_stats=None, gn_groups=4, out_dtype=torch.float32,
               gate_cstride=0, arith=None, tune=None, out_mul=None, out_mul_c0=0, in0_cstride=0):
        """gn_stats: zeroed float64 [B*gn_groups*2] tensor that receives the GroupNorm statistics of
        the (pre-activation) output, for a following groupnorm_apply().  out_dtype (channel-last outputs only): bf16 / fp16
        feature storage, rounded to nearest even in the epilogue.  arith: ARITH_F32 | ARITH_BF16 (default: this binding's
        conv_arith) -- bf16 rounds inputs and weights as they enter the matrix cores (fp32 accumulation, fp32 tensors); layers
        with one tap and channel-last outputs always compute in fp32."""
        if in0_cstride:       # x0 is a channel slice of a contiguous [B,in0_cstride,H,W] tensor (the r * h half of the merged gate conv's output)
            if (x0.dtype != torch.float32 or x0.device != self.device or x0.stride(1) != x0.shape[2] * x0.shape[3] or
                    x0.stride(0) != in0_cstride * x0.shape[2] * x0.shape[3] or x0.stride(3) != 1 or in_mode != IN_PLAIN):
                raise _lib.DmvsError("in0_cstride: x0 must be a channel slice of a contiguous [B,in0_cstride,H,W] tensor, plain input mode")
        self._chk(out_mul)
        if gate_cstride:      # mul0 / gru_z are channel slices of one [B,gate_cstride,H,W] tensor (merged z|r gate convolution)
            self._chk(*(() if in0_cstride else (x0,)), x1, residual, gru_h)
            for t in (mul0, gru_z):
                if t is not None and (t.dtype != torch.float32 or t.device != self.device or t.stride(1) != t.shape[2] * t.shape[3] or
                                      t.stride(0) != gate_cstride * t.shape[2] * t.shape[3] or t.stride(3) != 1):
                    raise _lib.DmvsError("gate_cstride: mul0 / gru_z must be channel slices of a contiguous [B,gate_cstride,H,W] tensor")
        else:
            self._chk(*(() if in0_cstride else (x0,)), x1, mul0, residual, gru_z, gru_h)
        if out_dtype != torch.float32:
            if out_layout != LAYOUT_NHWC or out is not None:
                raise _lib.DmvsError("16-bit outputs are channel-last feature tensors allocated by conv2d")
            out_layout = _NHWC_LAYOUT[out_dtype]
        else:
            self._chk(out)
        if gn_stats is not None and (gn_stats.dtype != torch.float64 or gn_stats.numel() < x0.shape[0] * gn_groups * 2):
            raise _lib.DmvsError("gn_stats must be float64 with B*groups*2 elements")
        B = x0.shape[0]
        if in_mode == IN_PLAIN:
            c0, Hin, Win = x0.shape[1], x0.shape[2], x0.shape[3]
        elif in_mode in (IN_UPSAMPLE2, IN_ZEROINSERT2):
            c0, Hin, Win = x0.shape[1], x0.shape[2] * 2, x0.shape[3] * 2
        else:
            c0, Hin, Win = x0.shape[1] * 4, x0.shape[2] // 2, x0.shape[3] // 2
        c1 = 0 if x1 is None else x1.shape[1]
        assert c0 + c1 == pc.cin, (c0, c1, pc.cin)
        kh, kw = pc.k
        Hout = (Hin + 2 * pc.pad[0] - kh) // pc.stride + 1
        Wout = (Win + 2 * pc.pad[1] - kw) // pc.stride + 1
        if out_cstride is None:
            out_cstride = pc.cout
        if out is None:
            shape = (B, out_cstride, Hout, Wout) if out_layout == LAYOUT_NCHW else (B, Hout, Wout, out_cstride)
            out = self.empty(*shape, dtype=out_dtype)
        arith = self.conv_arith if arith is None else arith
        wsplit = split_weights(pc) if (arith == ARITH_SPLIT and kh * kw > 1 and out_layout in (LAYOUT_NCHW, LAYOUT_NHWC)) else None
        d = _lib.Conv2dDesc(
            weight_split=_ptr(wsplit),
            in0=_ptr(x0), in1=_ptr(x1), mul0=_ptr(mul0), weight=_ptr(pc.weight), scale=_ptr(pc.scale),
            shift=_ptr(pc.shift), residual=_ptr(residual), gru_z=_ptr(gru_z), gru_h=_ptr(gru_h), out=_ptr(out),
            gn_stats=_ptr(gn_stats), gn_groups=(gn_groups if gn_stats is not None else 0), B=B, c0=c0, c1=c1, Hin=Hin, Win=Win, Hout=Hout, Wout=Wout, cout=pc.cout, cout_pad=pc.cout_pad,
            kh=kh, kw=kw, stride=pc.stride, pad_h=pc.pad[0], pad_w=pc.pad[1], in_mode=in_mode, act=act,
            res_mode=res_mode, res_after_act=int(res_after_act), out_layout=out_layout, out_cstride=out_cstride,
            out_coffset=out_coffset, post_scale=post_scale, gate_cstride=gate_cstride,
            arith=arith, tune=(self.tune["conv2d"] if tune is None else tune),
            out_mul=_ptr(out_mul), out_mul_c0=out_mul_c0, in0_cstride=in0_cstride)
        self._call("dmvs_conv2d_f32", C.byref(d), self.stream())
        if self.timers is not None and "dmvs_conv2d_f32" in self.timers:      # bench: MFMA roofline over every conv launch
            self.timers.setdefault("_conv2d_flops", []).append(2.0 * B * Hout * Wout * pc.cout * pc.cin * kh * kw)
            self.timers.setdefault("_conv2d_shape", []).append((B, pc.cin, pc.cout, kh, kw, pc.stride, Hout, Wout, in_mode, int(mul0 is not None)))
            # algorithmic bytes: every operand tensor read once, the output written once (SURVEY 8d's convention)
            nb = sum(t.numel() * t.element_size() for t in (x0, x1, mul0, residual, gru_z, gru_h, out_mul) if t is not None)
            self.timers.setdefault("_conv2d_bytes", []).append(nb + B * Hout * Wout * pc.cout * out.element_size())
        return out

    def featurenet_stem(self, pc0: PackedConv, pc1: PackedConv, x, tune=None):
        """relu(bn(conv0.1(relu(bn(conv0.0(x)))))) of FeatureNet in one kernel: x [N,3,H,W] -> [N,8,H,W].
        x may be a LIST of V tensors [B,3,H,W] (the views of a batch): one launch per view into consecutive slices of one
        [V*B,8,H,W] output -- the view stack is never concatenated"""
        if isinstance(x, (list, tuple)):
            self._chk(*x)
            Bv, cin, H, W = x[0].shape
            y = self.empty(len(x) * Bv, 8, H, W)
            for v, xv in enumerate(x):
                self._stem_launch(pc0, pc1, xv, y[v * Bv:(v + 1) * Bv], tune)
            return y
        self._chk(x)
        N, cin, H, W = x.shape
        y = self.empty(N, 8, H, W)
        self._stem_launch(pc0, pc1, x, y, tune)
        return y

    def _stem_launch(self, pc0: PackedConv, pc1: PackedConv, x, y, tune=None):
        N, cin, H, W = x.shape
        if not (cin == 3 and pc0.cin == 3 and pc0.cout == 8 and pc1.cin == 8 and pc1.cout == 8 and pc0.k == (3, 3) and
                pc1.k == (3, 3) and pc0.stride == 1 and pc1.stride == 1 and pc0.pad == (1, 1) and pc1.pad == (1, 1) and
                pc0.cout_pad == 8 and pc1.cout_pad == 8):
            raise _lib.DmvsError("featurenet_stem: expects the 3->8->8 3x3 stem of FeatureNet")
        tune = self.tune["stem"] if tune is None else tune
        if self.conv_arith != ARITH_SPLIT:      # conv0.1 in split-bf16 arithmetic (the library's default) only under conv_arith = "split"
            tune |= _lib.TUNE_STEM_EXACT
        self._call("dmvs_featurenet_stem_f32", _ptr(x), _ptr(pc0.weight), _ptr(pc0.scale), _ptr(pc0.shift), _ptr(pc1.weight),
                   _ptr(pc1.scale), _ptr(pc1.shift), _ptr(y), N, H, W, tune, self.stream())

    def conv2d_wgrad(self, pc: PackedConv, x0, grad_out, x1=None, *, mul0=None, in_mode=IN_PLAIN, want_bias=False, tune=None,
                     into_gw=None, into_gb=None):
        """Weight gradient of conv2d(pc, x0, x1, mul0=..., in_mode=...) in torch layout [cout, cin, kh, kw]
        (and the bias gradient [cout] when want_bias) -> gw | (gw, gb).  into_gw (/ into_gb): ADD the gradient to these running
        gradients (contiguous fp32 tensors of the right shapes, e.g. views of the trainer's flat bucket) instead of returning new tensors."""
        self._chk(x0, x1, mul0, grad_out, into_gw, into_gb)
        B = x0.shape[0]
        if in_mode == IN_PLAIN:
            c0, Hin, Win = x0.shape[1], x0.shape[2], x0.shape[3]
        elif in_mode in (IN_UPSAMPLE2, IN_ZEROINSERT2):
            c0, Hin, Win = x0.shape[1], x0.shape[2] * 2, x0.shape[3] * 2
        else:
            c0, Hin, Win = x0.shape[1] * 4, x0.shape[2] // 2, x0.shape[3] // 2
        c1 = 0 if x1 is None else x1.shape[1]
        kh, kw = pc.k
        Hout, Wout = grad_out.shape[2], grad_out.shape[3]
        acc = into_gw is not None
        if acc and (tuple(into_gw.shape) != (pc.cout, pc.cin, kh, kw) or (want_bias and (into_gb is None or tuple(into_gb.shape) != (pc.cout,)))):
            raise _lib.DmvsError("conv2d_wgrad: into_gw / into_gb do not have the gradient's shape")
        gw = into_gw if acc else self.empty(pc.cout, pc.cin, kh, kw)
        gb = (into_gb if acc else self.empty(pc.cout)) if want_bias else None
        d = _lib.Conv2dDesc(
            in0=_ptr(x0), in1=_ptr(x1), mul0=_ptr(mul0), weight=None, scale=None, shift=None, residual=None, gru_z=None,
            gru_h=None, out=None, gn_stats=None, gn_groups=0, B=B, c0=c0, c1=c1, Hin=Hin, Win=Win, Hout=Hout, Wout=Wout,
            cout=pc.cout, cout_pad=pc.cout_pad, kh=kh, kw=kw, stride=pc.stride, pad_h=pc.pad[0], pad_w=pc.pad[1],
            in_mode=in_mode, act=ACT_NONE, res_mode=IN_PLAIN, res_after_act=0, out_layout=LAYOUT_NCHW,
            out_cstride=pc.cout, out_coffset=0, post_scale=1.0, gate_cstride=0, arith=ARITH_F32,
            tune=((self.tune["conv2d"] & _lib.TUNE_PIECES4) if tune is None else tune) | (_lib.TUNE_WGRAD_ACCUMULATE if acc else 0))
        nbytes = C.c_int64(0)
        self._call("dmvs_conv2d_wgrad_workspace_f32", C.byref(d), C.byref(nbytes))
        ws = self.empty(nbytes.value // 4)
        self._call("dmvs_conv2d_wgrad_f32", C.byref(d), _ptr(grad_out), _ptr(gw), _ptr(gb), _ptr(ws), nbytes.value, self.stream())
        if self.timers is not None and "dmvs_conv2d_wgrad_f32" in self.timers:      # bench cfg4: MFMA roofline over every weight-gradient launch
            self.timers.setdefault("_wgrad_flops", []).append(2.0 * B * Hout * Wout * pc.cout * pc.cin * kh * kw)
        return (gw, gb) if want_bias else gw

    # ------------------------------------------------------------------ conv3d
    def conv3d(self, pc: PackedConv, x, *, act=ACT_NONE, residual=None, out=None, tune=None):
        self._chk(x, residual, out)
        B, cin, Din, Hin, Win = x.shape
        assert cin == pc.cin
        if pc.transposed:
            Dout, Hout, Wout = 2 * Din, 2 * Hin, 2 * Win
        else:
            s = pc.stride
            Dout, Hout, Wout = (Din - 1) // s + 1, (Hin - 1) // s + 1, (Win - 1) // s + 1
        if out is None:
            out = self.empty(B, pc.cout, Dout, Hout, Wout)
        d = _lib.Conv3dDesc(in_=_ptr(x), weight=_ptr(pc.weight), scale=_ptr(pc.scale), shift=_ptr(pc.shift),
                            residual=_ptr(residual), out=_ptr(out), B=B, cin=cin, cout=pc.cout, cout_pad=pc.cout_pad,
                            Din=Din, Hin=Hin, Win=Win, Dout=Dout, Hout=Hout, Wout=Wout, stride=pc.stride,
                            transposed=int(pc.transposed), act=act, tune=(self.tune["conv3d"] if tune is None else tune))
        self._call("dmvs_conv3d_f32", C.byref(d), self.stream())
        return out

    def conv3d_wgrad(self, x, grad_out, cout, stride, want_bias=False):
        """gw of a (non-transposed) 3x3x3 conv: x [B,cin,D,H,W], grad_out [B,cout,Do,Ho,Wo] -> [cout,cin,3,3,3] (, gb [cout])"""
        self._chk(x, grad_out)
        B, cin, Din, Hin, Win = x.shape
        Dout, Hout, Wout = grad_out.shape[2:]
        gw = self.empty(cout, cin, 3, 3, 3)
        gb = self.empty(cout) if want_bias else None
        d = _lib.Conv3dDesc(in_=_ptr(x), weight=None, scale=None, shift=None, residual=None, out=None, B=B, cin=cin,
                            cout=cout, cout_pad=_pad_cout(cout), Din=Din, Hin=Hin, Win=Win, Dout=Dout, Hout=Hout, Wout=Wout,
                            stride=stride, transposed=0, act=ACT_NONE)
        nbytes = C.c_int64(0)
        self._call("dmvs_conv3d_wgrad_workspace_f32", C.byref(d), C.byref(nbytes))
        ws = self.empty(nbytes.value // 4)
        self._call("dmvs_conv3d_wgrad_f32", C.byref(d), _ptr(grad_out), _ptr(gw), _ptr(gb), _ptr(ws), nbytes.value, self.stream())
        return (gw, gb) if want_bias else gw

    # ------------------------------------------------------------------ geometry / cost volumes
    def compose_proj(self, proj):
        """proj [B,V,2,4,4] -> [B,S,12] (rot row-major, trans)."""
        self._chk(proj)
        B, V = proj.shape[0], proj.shape[1]
        out = self.empty(B, V - 1, 12)
        self._call("dmvs_compose_proj_f32", _ptr(proj), _ptr(out), B, V, self.stream())
        return out

    def warp_corr_init_quad(self, ref, src, rt, disp_min, disp_max, D, G=4, tune=None, plain=False):
        """quad-per-pixel plane sweep: ref [B,H,W,C], src [S,B,Hs,Ws,C] in the NHWC-g4 channel order (fp32) or plain NHWC
        (bf16 / fp16 feature storage; fp32 with plain=True: the training graph's features) -> [B,S,G,D,H,W] fp32"""
        self._chk(rt, disp_min, disp_max)
        fdt = self._chk_feat(ref, src)
        if plain and fdt == DTYPE_F32:
            fdt = _lib.DTYPE_F32_PLAIN
        B, H, W, Cc = ref.shape
        S, _, Hs, Ws, _ = src.shape
        out = self.empty(B, S, G, D, H, W)
        self._call("dmvs_warp_corr_init_quad_f32", _ptr(ref), _ptr(src), _ptr(rt), _ptr(disp_min), _ptr(disp_max), _ptr(out),
                   B, S, Cc, G, D, H, W, Hs, Ws, fdt, self.tune["sweep"] if tune is None else tune, self.stream())
        if self.timers is not None and "dmvs_warp_corr_init_quad_f32" in self.timers:
            self.timers.setdefault("_warp_init_bytes", []).append(B * H * W * (ref.element_size() * (Cc + S * Cc) + 4 * S * G * D))
        return out

    def getcost_quad(self, ref, src, rt, inv_depth, confidence, view_w, disp_min, disp_max, n, interval, min_radius,
                     max_radius, vw_shift, out_cost=None, cost_cstride=None, cost_coffset=0, out_samples=None,
                     samp_cstride=None, samp_coffset=0, G=4, plain=False):
        """quad-per-pixel GetCost, one launch for any geometry; ref / src in the NHWC-g4 channel order (fp32) or plain NHWC
        (bf16 / fp16 feature storage; fp32 with plain=True: the training graph's features)"""
        self._chk(rt, inv_depth, confidence, view_w, disp_min, disp_max, out_cost, out_samples)
        fdt = self._chk_feat(ref, src)
        if plain and fdt == DTYPE_F32:
            fdt = _lib.DTYPE_F32_PLAIN
        B, H, W, Cc = ref.shape
        S = src.shape[0]
        if out_cost is None:
            cost_cstride = G * n
            out_cost = self.empty(B, G * n, H, W)
        if out_samples is None:
            samp_cstride = n
            out_samples = self.empty(B, n, H, W)
        d = _lib.GetCostDesc(ref=_ptr(ref), src=_ptr(src), rt=_ptr(rt), inv_depth=_ptr(inv_depth),
                             confidence=_ptr(confidence), view_w=_ptr(view_w), disp_min=_ptr(disp_min),
                             disp_max=_ptr(disp_max), out_cost=_ptr(out_cost), out_samples=_ptr(out_samples),
                             worklist=None, B=B, S=S, C=Cc, G=G, n=n, H=H, W=W, vw_shift=vw_shift, cost_cstride=cost_cstride,
                             cost_coffset=cost_coffset, samp_cstride=samp_cstride, samp_coffset=samp_coffset,
                             interval=interval, min_radius=min_radius, max_radius=max_radius, feat_dtype=fdt)
        if self.getcost_hook is not None:
            self.getcost_hook(d, {"ref": ref, "src": src, "rt": rt, "inv_depth": inv_depth, "confidence": confidence, "view_w": view_w,
                                  "disp_min": disp_min, "disp_max": disp_max, "out_cost": out_cost, "out_samples": out_samples})
        self._call("dmvs_getcost_quad_f32", C.byref(d), self.stream())
        if self.timers is not None and "dmvs_getcost_quad_f32" in self.timers:      # bench: algorithmic bytes of this launch
            es = ref.element_size()
            self.timers.setdefault("_getcost_bytes", []).append(B * H * W * (es * (Cc + S * Cc) + 4 * (n + S + G * n)))
        return out_cost, out_samples

    def warp_volume(self, src, rt, depth):
        """differentiable_warping: src [B,C,Hs,Ws], rt [B,12], depth [B,D,H,W] -> [B,C,D,H,W]."""
        self._chk(src, rt, depth)
        B, Cc, Hs, Ws = src.shape
        D, H, W = depth.shape[1:]
        out = self.empty(B, Cc, D, H, W)
        self._call("dmvs_warp_volume_f32", _ptr(src), _ptr(rt), _ptr(depth), _ptr(out), B, Cc, D, H, W, Hs, Ws,
                   self.stream())
        return out

    # ------------------------------------------------------------------ backward (training step)
    def warp_corr_init_bwd(self, ref, src, rt, disp_min, disp_max, gcor, gsrc=None, gather=True):
        """-> gref [B,H,W,C], gsrc [S,B,Hs,Ws,C] (accumulated into `gsrc` if given).  gather=False selects the LDS-window
        kernel (C = 48); measured slower than the per-pixel kernel at training sizes (10.9 vs 8.4 ms at cfg4: only ~10^3
        long-running tile x view workgroups), so it is not the default."""
        self._chk(ref, src, rt, disp_min, disp_max, gcor, gsrc)
        B, H, W, Cc = ref.shape
        S, _, Hs, Ws, _ = src.shape
        G, D = gcor.shape[2], gcor.shape[3]
        gref = self.empty_like(ref)
        if gsrc is None:
            gsrc = torch.zeros_like(src)
        mode = (_lib.BWD_GATHER_INTERLEAVED if self.tune.get("bwd_il") else 1) if gather else 0
        self._call("dmvs_warp_corr_init_bwd_f32", _ptr(ref), _ptr(src), _ptr(rt), _ptr(disp_min), _ptr(disp_max), _ptr(gcor),
                   _ptr(gref), _ptr(gsrc), B, S, Cc, G, D, H, W, Hs, Ws, mode, self.stream())
        return gref, gsrc

    def getcost_bwd(self, ref, src, rt, inv_depth, confidence, view_w, disp_min, disp_max, n, interval, min_radius,
                    max_radius, vw_shift, gcost, gsrc=None, G=4, gather=None):
        gather = self.tune.get("bwd_gather", False) if gather is None else gather
        self._chk(ref, src, rt, inv_depth, confidence, view_w, disp_min, disp_max, gcost, gsrc)
        B, H, W, Cc = ref.shape
        S = src.shape[0]
        gref = self.empty_like(ref)
        if gsrc is None:
            gsrc = torch.zeros_like(src)
        ntiles = B * ((H + 15) // 16) * ((W + 15) // 16)
        wl = None if (gather or Cc == 48 or S > MAX_WINDOW_VIEWS) else torch.empty(4 + 66 * ntiles, dtype=torch.int32, device=self.device)
        d = _lib.GetCostDesc(ref=_ptr(ref), src=_ptr(src), rt=_ptr(rt), inv_depth=_ptr(inv_depth),
                             confidence=_ptr(confidence), view_w=_ptr(view_w), disp_min=_ptr(disp_min),
                             disp_max=_ptr(disp_max), out_cost=None, out_samples=None, worklist=_ptr(wl), B=B, S=S, C=Cc, G=G, n=n,
                             H=H, W=W, vw_shift=vw_shift, cost_cstride=G * n, cost_coffset=0, samp_cstride=n, samp_coffset=0,
                             interval=interval, min_radius=min_radius, max_radius=max_radius,
                             tune=(_lib.TUNE_BWD_INTERLEAVED if self.tune.get("bwd_il") else 0))
        self._call("dmvs_getcost_bwd_f32", C.byref(d), _ptr(gcost), _ptr(gref), _ptr(gsrc), self.stream())
        if self.timers is not None and "dmvs_getcost_bwd_f32" in self.timers:
            # bench cfg4, algorithmic bytes: read ref + src + grad_cost, write grad_ref, read-modify-write grad_src (the scatter target)
            self.timers.setdefault("_getcost_bwd_bytes", []).append(4.0 * B * H * W * (Cc + S * Cc + G * n + Cc + 2 * S * Cc))
        return gref, gsrc

    def view_aggregate_bwd(self, cor, w, out, gout):
        self._chk(cor, w, out, gout)
        B, S, G, D, H, W = cor.shape
        gcor, gw = self.empty_like(cor), self.empty_like(w)
        self._call("dmvs_view_aggregate_bwd_f32", _ptr(cor), _ptr(w), _ptr(out), _ptr(gout), _ptr(gcor), _ptr(gw), B, S,
                   G * D, H * W, self.stream())
        return gcor, gw

    def view_aggregate(self, cor, w):
        """cor [B,S,G,D,H,W], w [B,S,H,W] -> [B,G,D,H,W]."""
        self._chk(cor, w)
        B, S, G, D, H, W = cor.shape
        out = self.empty(B, G, D, H, W)
        self._call("dmvs_view_aggregate_f32", _ptr(cor), _ptr(w), _ptr(out), B, S, G * D, H * W, self.stream())
        return out

    def sigmoid_max_d(self, x):
        """x [N,D,H,W] -> [N,H,W]."""
        self._chk(x)
        N, D, H, W = x.shape
        out = self.empty(N, H, W)
        self._call("dmvs_sigmoid_max_d_f32", _ptr(x), _ptr(out), N, D, H * W, self.stream())
        return out

    def depth_regress(self, logits, disp_min, disp_max):
        """logits [B,D,H,W] -> norm_depth [B,1,H,W], depth [B,H,W], conf [B,1,H,W]."""
        self._chk(logits, disp_min, disp_max)
        B, D, H, W = logits.shape
        nd, depth, conf = self.empty(B, 1, H, W), self.empty(B, H, W), self.empty(B, 1, H, W)
        self._call("dmvs_depth_regress_f32", _ptr(logits), _ptr(disp_min), _ptr(disp_max), _ptr(nd), _ptr(depth),
                      _ptr(conf), B, D, H * W, self.stream())
        return nd, depth, conf

    def convex_upsample(self, inv, mask, disp_min, disp_max, ratio, want_inv=True):
        """inv [B,1,H,W] or [B,H,W]; mask [B,9*r*r,H,W] -> (inv_up [B,rH,rW] | None, depth_up [B,rH,rW])."""
        self._chk(inv, mask, disp_min, disp_max)
        B, H, W = inv.shape[0], inv.shape[-2], inv.shape[-1]
        assert mask.shape[1] == 9 * ratio * ratio
        out_inv = self.empty(B, H * ratio, W * ratio) if want_inv else None
        out_depth = self.empty(B, H * ratio, W * ratio)
        self._call("dmvs_convex_upsample_f32", _ptr(inv), _ptr(mask), _ptr(disp_min), _ptr(disp_max), _ptr(out_inv),
                      _ptr(out_depth), B, H, W, ratio, self.stream())
        return out_inv, out_depth

    def mask_upsample4(self, pc: PackedConv, x, inv, disp_min, disp_max, post_scale=0.25, want_inv=False):
        """post_scale * conv1x1(pc, x) -> softmax over the 9 taps -> convex x4 upsampling of inv -> metric depth, in one launch (the
        144-channel mask is never materialised): x [B,64,H,W], inv [B,1,H,W] or [B,H,W] -> (inv_up | None, depth_up [B,4H,4W]).
        Bit-identical to convex_upsample(inv, conv2d(pc, x, post_scale=post_scale), ..., 4)."""
        self._chk(x, inv, disp_min, disp_max)
        B, cin, H, W = x.shape
        if pc.k != (1, 1) or pc.cin != cin or pc.cout != 144 or pc.scale is not None:
            raise _lib.DmvsError("mask_upsample4: expects the 64 -> 144 1x1 layer of the DiffMVS mask head")
        out_inv = self.empty(B, H * 4, W * 4) if want_inv else None
        out_depth = self.empty(B, H * 4, W * 4)
        self._call("dmvs_mask_upsample4_f32", _ptr(x), _ptr(pc.weight), _ptr(pc.shift), post_scale, _ptr(inv), _ptr(disp_min), _ptr(disp_max),
                   _ptr(out_inv), _ptr(out_depth), B, cin, pc.cout_pad, H, W, self.stream())
        return out_inv, out_depth

    def groupnorm_silu(self, x, gamma, beta, groups, scale_shift=None, residual=None, out=None, eps=1e-5):
        self._chk(x, gamma, beta, scale_shift, residual, out)
        B, Cc, H, W = x.shape
        if out is None:
            out = self.empty(B, Cc, H, W)
        stats = torch.empty(B * groups * 2, dtype=torch.float64, device=self.device)
        self._call("dmvs_groupnorm_silu_f32", _ptr(x), _ptr(gamma), _ptr(beta), _ptr(scale_shift), _ptr(residual),
                      _ptr(out), _ptr(stats), B, Cc, H * W, groups, eps, self.stream())
        return out

    def groupnorm_silu_train(self, x, gamma, beta, groups, scale_shift=None, eps=1e-5):
        """forward that also returns the statistics buffer the backward needs"""
        self._chk(x, gamma, beta, scale_shift)
        B, Cc, H, W = x.shape
        out = self.empty(B, Cc, H, W)
        stats = torch.empty(B * groups * 2, dtype=torch.float64, device=self.device)
        self._call("dmvs_groupnorm_silu_f32", _ptr(x), _ptr(gamma), _ptr(beta), _ptr(scale_shift), None, _ptr(out), _ptr(stats), B, Cc,
                   H * W, groups, eps, self.stream())
        return out, stats

    def groupnorm_silu_bwd(self, x, dy, gamma, beta, groups, stats, scale_shift=None, eps=1e-5):
        self._chk(x, dy, gamma, beta, scale_shift)
        B, Cc, H, W = x.shape
        dx = self.empty_like(x)
        dgamma, dbeta = self.empty(Cc), self.empty(Cc)
        dss = self.empty(B, 2 * Cc) if scale_shift is not None else None
        n = C.c_int64(0)
        self._call("dmvs_groupnorm_silu_bwd_workspace_f32", B, Cc, H * W, C.byref(n))
        ws = self.empty(max(n.value // 4, 1))
        self._call("dmvs_groupnorm_silu_bwd_f32", _ptr(x), _ptr(dy), _ptr(gamma), _ptr(beta), _ptr(scale_shift), _ptr(stats), _ptr(dx),
                   _ptr(dgamma), _ptr(dbeta), _ptr(dss), _ptr(ws), n.value, B, Cc, H * W, groups, eps, self.stream())
        return dx, dgamma, dbeta, dss

    def groupnorm_apply(self, x, gamma, beta, groups, stats, scale_shift=None, residual=None, out=None, eps=1e-5):
        """normalise + scale/shift + SiLU (+ residual) with statistics accumulated by conv2d(gn_stats=...)."""
        self._chk(x, gamma, beta, scale_shift, residual, out)
        B, Cc, H, W = x.shape
        if out is None:
            out = self.empty(B, Cc, H, W)
        self._call("dmvs_groupnorm_apply_f32", _ptr(x), _ptr(gamma), _ptr(beta), _ptr(scale_shift), _ptr(residual),
                   _ptr(out), _ptr(stats), B, Cc, H * W, groups, eps, self.stream())
        return out

    def delta_update(self, inv, delta_in, update, delta_in_scale=1.0, new2=None, new2_cstride=0, new2_coffset=0):
        """-> (delta_out, new_inv), also mirrors new_inv into channel new2_coffset of new2."""
        self._chk(inv, delta_in, update, new2)
        B = inv.shape[0]
        HW = inv.numel() // B
        delta_out = self.empty_like(inv)
        new_inv = self.empty_like(inv)
        self._call("dmvs_delta_update_f32", _ptr(inv), _ptr(delta_in), _ptr(update), delta_in_scale,
                      _ptr(delta_out), _ptr(new_inv), _ptr(new2), new2_cstride, new2_coffset, B, HW, self.stream())
        return delta_out, new_inv

    def depth_convert(self, x, disp_min, disp_max, mode):
        self._chk(x, disp_min, disp_max)
        B = x.shape[0]
        out = self.empty_like(x)
        self._call("dmvs_depth_convert_f32", _ptr(x), _ptr(disp_min), _ptr(disp_max), _ptr(out), mode, B,
                      x.numel() // B, self.stream())
        return out

    def act_slice(self, x, act, c_from, c_count, out=None, out_cstride=None, out_coffset=0):
        self._chk(x, out)
        B, Ct, H, W = x.shape
        if out is None:
            out_cstride = c_count
            out = self.empty(B, c_count, H, W)
        self._call("dmvs_act_slice_f32", _ptr(x), _ptr(out), act, B, c_count, H * W, Ct, c_from, out_cstride,
                      out_coffset, self.stream())
        return out

    def upsample_nearest(self, x, factor):
        """x [..., H, W] -> [..., fH, fW]."""
        self._chk(x)
        H, W = x.shape[-2], x.shape[-1]
        N = x.numel() // (H * W)
        out = self.empty(*x.shape[:-2], H * factor, W * factor)
        self._call("dmvs_upsample_nearest_f32", _ptr(x), _ptr(out), N, H, W, factor, self.stream())
        return out

    def nchw_to_nhwc(self, x):
        self._chk(x)
        B, Cc, H, W = x.shape
        out = self.empty(B, H, W, Cc)
        self._call("dmvs_nchw_to_nhwc_f32", _ptr(x), _ptr(out), B, Cc, H * W, self.stream())
        return out

    # ------------------------------------------------------------------ training-mode BatchNorm
    def _bn_ws(self, B, Cc, S, views):
        n = C.c_int64(0)
        self._call("dmvs_batchnorm_workspace_f32", B, Cc, S, views, C.byref(n))
        return self.empty(max(n.value // 4, 1)), n.value

    def batchnorm_train_fwd(self, x, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5, act=ACT_NONE, views=1,
                            view_major=True):
        """x [B,C,*spatial] -> y, save_mean [views,C], save_rstd [views,C]; running stats updated in place (once per view)"""
        self._chk(x, gamma, beta, running_mean, running_var)
        B, Cc = x.shape[0], x.shape[1]
        S = x.numel() // (B * Cc)
        y = self.empty_like(x)
        mean, rstd = self.empty(views, Cc), self.empty(views, Cc)
        ws, nb = self._bn_ws(B, Cc, S, views)
        bump_weights_generation()         # running_mean / running_var are rewritten in place
        self._call("dmvs_batchnorm_train_fwd_f32", _ptr(x), _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var), _ptr(y),
                   _ptr(mean), _ptr(rstd), _ptr(ws), nb, B, Cc, S, views, int(view_major), momentum, eps, act, self.stream())
        return y, mean, rstd

    def batchnorm_train_bwd(self, x, dy, gamma, beta, mean, rstd, act=ACT_NONE, views=1, view_major=True):
        self._chk(x, dy, gamma, beta, mean, rstd)
        B, Cc = x.shape[0], x.shape[1]
        S = x.numel() // (B * Cc)
        dx = self.empty_like(x)
        dgamma, dbeta = self.empty(Cc), self.empty(Cc)
        ws, nb = self._bn_ws(B, Cc, S, views)
        self._call("dmvs_batchnorm_train_bwd_f32", _ptr(x), _ptr(dy), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(rstd), _ptr(dx),
                   _ptr(dgamma), _ptr(dbeta), _ptr(ws), nb, B, Cc, S, views, int(view_major), act, self.stream())
        return dx, dgamma, dbeta

    # ------------------------------------------------------------------ training-step tail
    def sumsq(self, g, out=None):
        """sum of squares of a flat fp32 tensor -> device double scalar"""
        self._chk(g)
        if out is None:
            out = self.empty(1, dtype=torch.float64)
        self._call("dmvs_sumsq_f32", _ptr(g), g.numel(), _ptr(out), self.stream())
        return out

    def adamw_step(self, p, g, m, v, lr, beta1, beta2, eps, wd, step, grad_scale=1.0, sumsq=None, max_norm=0.0):
        self._chk(p, g, m, v)
        bump_weights_generation()         # parameters are rewritten in place
        self._call("dmvs_adamw_step_f32", _ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), lr, beta1, beta2, eps, wd, step,
                   grad_scale, _ptr(sumsq), max_norm, self.stream())
