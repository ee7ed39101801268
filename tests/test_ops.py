"""Kernel-level parity: every C-ABI entry point against the oracle / the matching ATen op.
Runs twice: on the host emulation of the kernel sources (CPU container) and, with -m gpu,
through libdmvs_hip.so on the MI355X."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from diffmvs_amd import ops as K
from diffmvs_amd import synth
from oracle import diffmvs_oracle as O


def rnd(*shape, seed=0, lo=-1.0, hi=1.0):
    rs = np.random.RandomState(seed + sum(shape))
    return torch.from_numpy(rs.uniform(lo, hi, shape).astype(np.float32))


def dev(ops, *ts):
    r = [None if t is None else t.to(ops.device).contiguous() for t in ts]
    return r if len(r) > 1 else r[0]


def _warp_feats(ops, feats, plain):
    """ref [B,H,W,C], src [S,B,H,W,C] for the quad warp kernels from a list of V NCHW feature maps: plain NHWC fp32 (the training graph's
    order, feat_dtype DMVS_DTYPE_F32_PLAIN) or the group-interleaved NHWC-g4 order the inference engine emits"""
    ref = feats[0].permute(0, 2, 3, 1)
    src = torch.stack([f.permute(0, 2, 3, 1) for f in feats[1:]])
    if not plain:
        perm = K.g4_channels(ref.shape[-1])
        ref, src = ref[..., perm], src[..., perm]
    return dev(ops, ref.contiguous()), dev(ops, src.contiguous())


def close(a, b, tol=1e-5):
    a = a.detach().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.isfinite(a).all()
    err = float((a - b).abs().max())
    scale = max(1.0, float(b.abs().max()))
    assert err <= tol * scale, f"max abs err {err:.3e} (scale {scale:.3e})"


ACTS = {K.ACT_NONE: lambda x: x, K.ACT_RELU: F.relu, K.ACT_SIGMOID: torch.sigmoid, K.ACT_TANH: torch.tanh,
        K.ACT_SILU: F.silu}


@pytest.mark.parametrize("cin,cout,k,stride,pad,act", [
    (3, 8, (3, 3), 1, (1, 1), K.ACT_RELU),
    (8, 16, (5, 5), 2, (2, 2), K.ACT_RELU),
    (16, 48, (1, 1), 1, (0, 0), K.ACT_NONE),
    (12, 31, (3, 3), 1, (1, 1), K.ACT_RELU),
    (10, 16, (7, 7), 1, (3, 3), K.ACT_NONE),
    (9, 20, (1, 5), 1, (0, 2), K.ACT_SIGMOID),
    (9, 20, (5, 1), 1, (2, 0), K.ACT_TANH),
    (6, 36, (3, 3), 2, (1, 1), K.ACT_SILU),
    (64, 144, (1, 1), 1, (0, 0), K.ACT_NONE),
])
def test_conv2d_basic(ops, cin, cout, k, stride, pad, act):
    B, H, W = 2, 13, 18
    x = rnd(B, cin, H, W, seed=1)
    w = rnd(cout, cin, *k, seed=2) * 0.3
    bias = rnd(cout, seed=3)
    bn = {"weight": rnd(cout, seed=4, lo=0.5, hi=1.5), "bias": rnd(cout, seed=5),
          "running_mean": rnd(cout, seed=6), "running_var": rnd(cout, seed=7, lo=0.5, hi=1.5)}
    for use_bn in (False, True):
        ref = F.conv2d(x, w, None if use_bn else bias, stride, pad)
        if use_bn:
            ref = F.batch_norm(ref, bn["running_mean"], bn["running_var"], bn["weight"], bn["bias"], False, 0.0, 1e-5)
        ref = ACTS[act](ref)
        pc = K.pack_conv2d(*dev(ops, w, None if use_bn else bias), bn=None if not use_bn else
                           {k_: v.to(ops.device) for k_, v in bn.items()}, stride=stride, pad=pad)
        out = ops.conv2d(pc, dev(ops, x), act=act)
        close(out, ref, 2e-5)


@pytest.mark.parametrize("cin,cout,k,stride,pad", [(5, 8, (3, 3), 1, (1, 1)), (16, 48, (1, 1), 1, (0, 0)), (12, 31, (3, 3), 1, (1, 1)),
                                                   (10, 16, (7, 7), 1, (3, 3)), (9, 20, (1, 5), 1, (0, 2)), (9, 20, (5, 1), 1, (2, 0)),
                                                   (6, 36, (3, 3), 2, (1, 1)), (7, 16, (5, 5), 2, (2, 2)), (33, 64, (3, 3), 1, (1, 1))])
@pytest.mark.parametrize("W,misalign", [(20, False), (36, False), (20, True)])
def test_conv2d_row_alignment(ops, cin, cout, k, stride, pad, W, misalign):
    """rows that are 16-byte multiples, on a 16-byte aligned tensor and on the same tensor at a 4-byte offset (no staging or
    epilogue path may assume more than element alignment); tiles ragged in both axes, partial last channel chunk, image
    narrower than a tile, every kernel shape of the model"""
    B, H = 2, 21
    x = rnd(B, cin, H, W, seed=1)
    w = rnd(cout, cin, *k, seed=2) * 0.3
    bias = rnd(cout, seed=3)
    ref = F.relu(F.conv2d(x, w, bias, stride, pad))
    xd = dev(ops, x)
    if misalign:
        store = torch.zeros(x.numel() + 1, device=ops.device)
        store[1:] = xd.reshape(-1)
        xd = store[1:].view(x.shape)
        assert xd.data_ptr() % 16 == 4
    out = ops.conv2d(K.pack_conv2d(*dev(ops, w, bias), stride=stride, pad=pad), xd, act=K.ACT_RELU)
    close(out, ref, 2e-5)


def _at_4_byte_offset(ops, t):
    td = dev(ops, t)
    store = torch.zeros(t.numel() + 1, device=ops.device)
    store[1:] = td.reshape(-1)
    out = store[1:].view(t.shape)
    assert out.data_ptr() % 16 == 4
    return out


@pytest.mark.parametrize("cin,cout,k,stride,pad,nhwc", [(16, 16, (3, 3), 1, (1, 1), False), (20, 32, (3, 3), 1, (1, 1), False), (32, 64, (3, 3), 1, (1, 1), False),
                                                        (8, 16, (5, 5), 2, (2, 2), False), (16, 32, (5, 5), 2, (2, 2), False), (12, 16, (7, 7), 1, (3, 3), False),
                                                        (16, 24, (1, 5), 1, (0, 2), False), (16, 24, (5, 1), 1, (2, 0), False), (8, 32, (3, 3), 2, (1, 1), False),
                                                        (32, 144, (1, 1), 1, (0, 0), False), (32, 64, (1, 1), 1, (0, 0), True), (16, 32, (3, 3), 1, (1, 1), True)])
@pytest.mark.parametrize("H,W", [(37, 52), (70, 40)])
def test_conv2d_16_byte_staging_pieces(ops, cin, cout, k, stride, pad, nhwc, H, W):
    """the input halo staged in 16-byte pieces (rows of 16-byte multiples on a 16-byte aligned tensor) against torch and BIT FOR BIT
    against the 4-byte form (the same tensor at a 4-byte offset): every kernel shape, several tiles in both axes with ragged last
    tiles, pieces outside the image on all four borders, partial last channel chunk, planar and channel-last outputs"""
    B = 2
    x = rnd(B, cin, H, W, seed=1)
    w = rnd(cout, cin, *k, seed=2) * 0.3
    bias = rnd(cout, seed=3)
    ref = F.relu(F.conv2d(x, w, bias, stride, pad))
    pc = K.pack_conv2d(*dev(ops, w, bias), stride=stride, pad=pad)
    lay = K.LAYOUT_NHWC if nhwc else K.LAYOUT_NCHW
    a = ops.conv2d(pc, dev(ops, x), act=K.ACT_RELU, out_layout=lay)
    b = ops.conv2d(pc, _at_4_byte_offset(ops, x), act=K.ACT_RELU, out_layout=lay)
    close(a, ref.permute(0, 2, 3, 1).contiguous() if nhwc else ref, 2e-5)
    assert torch.equal(a.cpu(), b.cpu())


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_conv2d_random_shapes_16_byte_vs_4_byte_pieces(seed):
    """seeded random layer shapes (every kernel-size family, 3..64 input and 1..144 output channels, planes from 1 x 4 to 40 x 68,
    planar / channel-last output, optional residual): torch reference, and the 16-byte staging form BIT FOR BIT against the 4-byte
    form.  Host-emulated only for now: random shapes reach instantiations no GPU run has exercised yet (add the hip arm after one)."""
    import random
    from conftest import emu_ops
    o = emu_ops()
    rng = random.Random(seed)
    fams = [((3, 3), 1, (1, 1)), ((3, 3), 2, (1, 1)), ((5, 5), 2, (2, 2)), ((7, 7), 1, (3, 3)), ((1, 5), 1, (0, 2)), ((5, 1), 1, (2, 0)),
            ((1, 1), 1, (0, 0))]
    for it in range(12):
        k, s, pad = rng.choice(fams)
        cin = rng.choice([3, 4, 5, 8, 12, 16, 17, 24, 32, 33, 48, 64])
        cout = rng.choice([1, 8, 12, 16, 20, 31, 32, 36, 48, 64, 144])
        B, H, W = rng.choice([1, 2, 3]), rng.choice([1, 2, 5, 8, 9, 16, 17, 31, 33, 40]), 4 * rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 12, 13, 16, 17])
        if k == (7, 7) and (cin > 40 or cout > 32):
            cin, cout = 12, 16
        x = rnd(B, cin, H, W, seed=100 * seed + it)
        w, b = rnd(cout, cin, *k, seed=7) * 0.3, rnd(cout, seed=8)
        lay = rng.choice([K.LAYOUT_NCHW, K.LAYOUT_NCHW, K.LAYOUT_NHWC])
        ref = F.conv2d(x, w, b, s, pad)
        res = rnd(*ref.shape, seed=9) if (lay == K.LAYOUT_NCHW and rng.random() < 0.4) else None
        ref = F.relu(ref + res if res is not None else ref)
        pc = K.pack_conv2d(w, b, stride=s, pad=pad)
        a = o.conv2d(pc, x, act=K.ACT_RELU, out_layout=lay, residual=res)
        c = o.conv2d(pc, _at_4_byte_offset(o, x), act=K.ACT_RELU, out_layout=lay, residual=res)
        close(a, ref.permute(0, 2, 3, 1).contiguous() if lay == K.LAYOUT_NHWC else ref, 2e-5)
        assert torch.equal(a, c), (k, s, cin, cout, B, H, W, lay)


def test_conv2d_16_byte_staging_pieces_fused_inputs(ops):
    """... with the second concat input and the r*h gating of the GRU candidate conv (scales the staged tile in place)"""
    B, H, W = 2, 21, 24
    h, x = rnd(B, 12, H, W, seed=1), rnd(B, 10, H, W, seed=2)
    r, z = torch.sigmoid(rnd(B, 12, H, W, seed=3)), torch.sigmoid(rnd(B, 12, H, W, seed=4))
    for k, pad in (((1, 5), (0, 2)), ((5, 1), (2, 0)), ((3, 3), (1, 1))):
        w, bias = rnd(12, 22, *k, seed=5) * 0.3, rnd(12, seed=6)
        q = torch.tanh(F.conv2d(torch.cat([r * h, x], 1), w, bias, 1, pad))
        pc = K.pack_conv2d(*dev(ops, w, bias), pad=pad)
        a = ops.conv2d(pc, *dev(ops, h, x), mul0=dev(ops, r), act=K.ACT_TANH, gru_z=dev(ops, z), gru_h=dev(ops, h))
        b = ops.conv2d(pc, _at_4_byte_offset(ops, h), dev(ops, x), mul0=dev(ops, r), act=K.ACT_TANH, gru_z=dev(ops, z), gru_h=dev(ops, h))
        close(a, (1 - z) * h + z * q, 2e-5)
        assert torch.equal(a.cpu(), b.cpu())


@pytest.mark.parametrize("hd,cx,H,W", [(12, 10, 21, 24), (32, 48, 9, 20), (6, 10, 10, 13)])
def test_conv2d_gru_pass_with_the_gate_product_in_the_producer(ops, hd, cx, H, W):
    """one SepConvGRU pass (module.py:164-177) the way the engine runs it since round 4: the merged z|r convolution writes [z | r * h]
    (out_mul / out_mul_c0: channels >= hd multiplied by h after the sigmoid), the candidate convolution reads r * h as a plain
    channel slice of that tensor (in0_cstride) -- against torch, and BIT FOR BIT against the form that gates the staged tile
    (mul0): the same fp32 product, computed once in the producer"""
    B = 2
    h, x = rnd(B, hd, H, W, seed=1), rnd(B, cx, H, W, seed=2)
    for k, pad in (((1, 5), (0, 2)), ((5, 1), (2, 0))):
        wz, wr, wq = (rnd(hd, hd + cx, *k, seed=s) * 0.3 for s in (3, 4, 5))
        bz, br, bq = (rnd(hd, seed=s) for s in (6, 7, 8))
        hx = torch.cat([h, x], 1)
        z, r = torch.sigmoid(F.conv2d(hx, wz, bz, 1, pad)), torch.sigmoid(F.conv2d(hx, wr, br, 1, pad))
        q = torch.tanh(F.conv2d(torch.cat([r * h, x], 1), wq, bq, 1, pad))
        ref = (1 - z) * h + z * q
        pzr = K.pack_conv2d(*dev(ops, torch.cat([wz, wr]), torch.cat([bz, br])), pad=pad)
        pq = K.pack_conv2d(*dev(ops, wq, bq), pad=pad)
        hdv, xdv = dev(ops, h, x)
        zr = ops.conv2d(pzr, hdv, xdv, act=K.ACT_SIGMOID, out_mul=hdv, out_mul_c0=hd)
        close(zr[:, :hd], z, 2e-5)
        close(zr[:, hd:], r * h, 2e-5)
        out = ops.conv2d(pq, zr[:, hd:], xdv, in0_cstride=2 * hd, act=K.ACT_TANH, gru_z=zr[:, :hd], gru_h=hdv, gate_cstride=2 * hd)
        close(out, ref, 2e-5)
        zr0 = ops.conv2d(pzr, hdv, xdv, act=K.ACT_SIGMOID)
        old = ops.conv2d(pq, hdv, xdv, mul0=zr0[:, hd:], act=K.ACT_TANH, gru_z=zr0[:, :hd], gru_h=hdv, gate_cstride=2 * hd)
        assert torch.equal(zr[:, :hd].cpu(), zr0[:, :hd].cpu()) and torch.equal(out.cpu(), old.cpu())
    with pytest.raises(K._lib.DmvsError):      # the slice must come from a [B, in0_cstride, H, W] tensor
        ops.conv2d(pq, hdv, xdv, in0_cstride=2 * hd, act=K.ACT_TANH)


@pytest.mark.parametrize("c0,cout,H,W,res", [(32, 64, 10, 16, "up"), (64, 144, 7, 8, None), (48, 32, 5, 44, "same"), (6, 36, 33, 36, None), (64, 96, 16, 64, "up")])
def test_conv2d_1x1_16_byte_form_wide(c0, cout, H, W, res):
    """the 2..4 n-tile instantiations and the channel groups of the 16-byte 1x1 form (2 n-tiles per workgroup with a residual, up to 4
    without) against torch and bit for bit against the tiled kernel -- host-emulated"""
    from conftest import emu_ops
    ops = emu_ops()
    B = 2
    x = rnd(B, c0, H, W, seed=1)
    w, bias = rnd(cout, c0, 1, 1, seed=3) * 0.3, rnd(cout, seed=4)
    ref = F.conv2d(x, w, bias)
    r = None
    if res == "same":
        r = rnd(B, cout, H, W, seed=5)
        ref = ref + r
    elif res == "up":
        r = rnd(B, cout, H // 2, W // 2, seed=5)
        ref = ref + F.interpolate(r, scale_factor=2, mode="nearest")
    out = ops.conv2d(K.pack_conv2d(w, bias), x, residual=r, res_mode=K.IN_UPSAMPLE2 if res == "up" else K.IN_PLAIN, act=K.ACT_RELU,
                     tune=K._lib.TUNE_1X1_TILED)
    close(out, F.relu(ref), 2e-5)
    px4 = ops.conv2d(K.pack_conv2d(w, bias), x, residual=r, res_mode=K.IN_UPSAMPLE2 if res == "up" else K.IN_PLAIN, act=K.ACT_RELU)
    assert torch.equal(out, px4)


@pytest.mark.parametrize("W,misalign", [(13, False), (18, False), (16, True)])
def test_conv2d_fusions_ragged_rows(ops, W, misalign):
    """the fused epilogues (GRU blend with r*h gating, residual before / after the activation, nearest-x2 residual, GroupNorm
    statistics) on rows that are not 16-byte multiples or on operands at a 4-byte offset: the 4-pixel groups of the
    transposed accumulators take their element-wise form, partially outside the image in the last group of a row"""
    B, H = 2, 10

    def place(t):      # the tensor at a 4-byte offset from a 16-byte boundary
        td = dev(ops, t)
        if not misalign:
            return td
        store = torch.zeros(t.numel() + 1, device=ops.device)
        store[1:] = td.reshape(-1)
        return store[1:].view(t.shape)

    h, x = rnd(B, 6, H, W, seed=1), rnd(B, 10, H, W, seed=2)
    r, z = torch.sigmoid(rnd(B, 6, H, W, seed=3)), torch.sigmoid(rnd(B, 6, H, W, seed=4))
    w, bias = rnd(6, 16, 3, 3, seed=5) * 0.3, rnd(6, seed=6)
    q = torch.tanh(F.conv2d(torch.cat([r * h, x], 1), w, bias, 1, 1))
    out = ops.conv2d(K.pack_conv2d(*dev(ops, w, bias), pad=1), place(h), place(x), mul0=place(r), act=K.ACT_TANH, gru_z=place(z), gru_h=place(h))
    close(out, (1 - z) * h + z * q, 2e-5)
    # residual before and after the activation
    x8, res = rnd(B, 8, H, W, seed=7), rnd(B, 12, H, W, seed=8)
    w3, b3 = rnd(12, 8, 3, 3, seed=9) * 0.3, rnd(12, seed=10)
    pc = K.pack_conv2d(*dev(ops, w3, b3), pad=1)
    close(ops.conv2d(pc, place(x8), residual=place(res), act=K.ACT_RELU), F.relu(F.conv2d(x8, w3, b3, 1, 1) + res), 2e-5)
    close(ops.conv2d(pc, place(x8), residual=place(res), act=K.ACT_RELU, res_after_act=True), F.relu(F.conv2d(x8, w3, b3, 1, 1)) + res, 2e-5)
    # GroupNorm statistics of the pre-activation output
    stats = torch.zeros(B, 4, 2, dtype=torch.float64, device=ops.device)
    y = ops.conv2d(pc, place(x8), gn_stats=stats, gn_groups=4)
    ref = F.conv2d(x8, w3, b3, 1, 1).reshape(B, 4, -1)
    got = stats.view(torch.int64).double().cpu() / 65536.0
    assert float((got[..., 0] - ref.sum(-1).double()).abs().max()) < 1e-2
    assert float((got[..., 1] - (ref.double() ** 2).sum(-1)).abs().max()) < 1e-1
    close(y, F.conv2d(x8, w3, b3, 1, 1), 2e-5)
    if W % 2 == 0:      # nearest-x2 residual
        rs = rnd(B, 12, H // 2, W // 2, seed=11)
        close(ops.conv2d(pc, place(x8), residual=place(rs), res_mode=K.IN_UPSAMPLE2), F.conv2d(x8, w3, b3, 1, 1) + F.interpolate(rs, scale_factor=2, mode="nearest"), 2e-5)


@pytest.mark.parametrize("c0,c1,cout,H,W,res,act", [
    (16, 0, 1, 9, 12, None, "sigmoid"), (32, 0, 16, 21, 20, None, "relu"), (32, 0, 64, 10, 16, "up", "none"), (64, 0, 144, 7, 8, None, "none"),
    (48, 0, 13, 5, 44, "same", "relu"), (6, 0, 9, 33, 36, None, "tanh"), (20, 10, 16, 12, 28, "same", "none"), (64, 0, 16, 16, 64, "up", "relu"),
    (3, 0, 8, 1, 4, None, "none"), (65, 0, 16, 6, 8, None, "none"), (20, 10, 31, 12, 28, "same", "none"), (30, 0, 12, 40, 52, "up", "none")])
def test_conv2d_1x1_direct(ops, c0, c1, cout, H, W, res, act):
    """1x1 layers with rows of 16-byte multiples on aligned tensors take the 16-byte direct form (no LDS input tile), the others the
    tiled kernel with transposed accumulators: channel counts that are not multiples of 4, a concatenated
    second input, same-size and nearest-x2 residuals before the activation, a ragged last pixel tile, output into a
    channel slice, more than 64 input channels (tiled kernel again)"""
    B = 2
    x0 = rnd(B, c0, H, W, seed=1)
    x1 = rnd(B, c1, H, W, seed=2) if c1 else None
    w, bias = rnd(cout, c0 + c1, 1, 1, seed=3) * 0.3, rnd(cout, seed=4)
    ref = F.conv2d(x0 if x1 is None else torch.cat([x0, x1], 1), w, bias)
    r = None
    if res == "same":
        r = rnd(B, cout, H, W, seed=5)
        ref = ref + r
    elif res == "up":
        r = rnd(B, cout, H // 2, W // 2, seed=5)
        ref = ref + F.interpolate(r, scale_factor=2, mode="nearest")
    ref = {"none": lambda t: t, "relu": F.relu, "sigmoid": torch.sigmoid, "tanh": torch.tanh}[act](ref) * 0.5
    big = torch.full((B, cout + 5, H, W), 7.0)
    outs = []
    for tune in (0, K._lib.TUNE_1X1_TILED):      # the 16-byte form (round 4, where it applies) and the earlier forms: same bits
        bigd = dev(ops, big)
        ops.conv2d(K.pack_conv2d(*dev(ops, w, bias)), dev(ops, x0), None if x1 is None else dev(ops, x1),
                   residual=None if r is None else dev(ops, r), res_mode=K.IN_UPSAMPLE2 if res == "up" else K.IN_PLAIN,
                   act={"none": K.ACT_NONE, "relu": K.ACT_RELU, "sigmoid": K.ACT_SIGMOID, "tanh": K.ACT_TANH}[act], post_scale=0.5,
                   out=bigd, out_cstride=cout + 5, out_coffset=3, tune=tune)
        close(bigd[:, 3:3 + cout], ref, 2e-5)
        assert float(bigd[:, :3].min()) == 7.0 and float(bigd[:, 3 + cout:].min()) == 7.0
        outs.append(bigd.cpu())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("cin,cout,H,W,dt", [(64, 48, 9, 20, torch.float32), (32, 31, 7, 8, torch.float32), (64, 48, 9, 20, torch.bfloat16),
                                              (48, 16, 5, 12, torch.float16), (64, 144, 6, 8, torch.float32)])
def test_conv2d_1x1_16_byte_form_channel_last(ops, cin, cout, H, W, dt):
    """the 16-byte 1x1 form writing channel-last outputs (FeatureNet's out1: fp32 NHWC / NHWC-g4 and the 16-bit feature storage): 16-byte
    stores of 4 channels per pixel, ragged channel counts, several channel groups per layer; BIT FOR BIT the tiled kernel"""
    B = 2
    x, w, bias = rnd(B, cin, H, W, seed=1), rnd(cout, cin, 1, 1, seed=2) * 0.3, rnd(cout, seed=3)
    ref = F.conv2d(x, w, bias).permute(0, 2, 3, 1)
    pc = K.pack_conv2d(*dev(ops, w, bias))
    a = ops.conv2d(pc, dev(ops, x), out_layout=K.LAYOUT_NHWC, out_dtype=dt)
    b = ops.conv2d(pc, dev(ops, x), out_layout=K.LAYOUT_NHWC, out_dtype=dt, tune=K._lib.TUNE_1X1_TILED)
    close(a.float(), ref, 2e-5 if dt == torch.float32 else 1e-2)
    assert torch.equal(a.cpu(), b.cpu())


@pytest.mark.parametrize("cin,cout,stride,H,W", [(3, 8, 1, 250, 262), (8, 8, 1, 256, 256), (8, 16, 2, 260, 500), (8, 24, 1, 256, 258)])
def test_conv2d_large_image_stem(ops, cin, cout, stride, H, W):
    """image sizes that select the 16x16-pixel tile (MT=4) instantiations of the FeatureNet / ContextNet stem layers"""
    B = 2
    x = rnd(B, cin, H, W, seed=1)
    w, bias = rnd(cout, cin, 3, 3, seed=2) * 0.3, rnd(cout, seed=3)
    ref = F.relu(F.conv2d(x, w, bias, stride, 1))
    out = ops.conv2d(K.pack_conv2d(*dev(ops, w, bias), stride=stride, pad=1), dev(ops, x), act=K.ACT_RELU)
    close(out, ref, 2e-5)


def test_conv2d_fusions(ops):
    B, H, W = 2, 10, 12
    # concat + r*h gating + GRU blend  (reference models/module.py:164-177)
    h, x = rnd(B, 6, H, W, seed=1), rnd(B, 10, H, W, seed=2)
    r, z = torch.sigmoid(rnd(B, 6, H, W, seed=3)), torch.sigmoid(rnd(B, 6, H, W, seed=4))
    w, bias = rnd(6, 16, 1, 5, seed=5) * 0.3, rnd(6, seed=6)
    q = torch.tanh(F.conv2d(torch.cat([r * h, x], 1), w, bias, 1, (0, 2)))
    ref = (1 - z) * h + z * q
    pc = K.pack_conv2d(*dev(ops, w, bias), pad=(0, 2))
    out = ops.conv2d(pc, *dev(ops, h, x), mul0=dev(ops, r), act=K.ACT_TANH, gru_z=dev(ops, z), gru_h=dev(ops, h))
    close(out, ref, 2e-5)
    # nearest-x2 upsampled input (update.py:38-42) and pre-activation residual read through upsampling
    xs = rnd(B, 8, H // 2, W // 2, seed=7)
    w3, b3 = rnd(12, 8, 3, 3, seed=8) * 0.3, rnd(12, seed=9)
    ref = F.conv2d(F.interpolate(xs, scale_factor=2, mode="nearest"), w3, b3, 1, 1)
    out = ops.conv2d(K.pack_conv2d(*dev(ops, w3, b3), pad=1), dev(ops, xs), in_mode=K.IN_UPSAMPLE2)
    close(out, ref, 2e-5)
    res = rnd(B, 12, H // 2, W // 2, seed=10)
    x8 = rnd(B, 8, H, W, seed=11)
    w1 = rnd(12, 8, 1, 1, seed=12)
    ref = F.interpolate(res, scale_factor=2, mode="nearest") + F.conv2d(x8, w1, b3)
    out = ops.conv2d(K.pack_conv2d(*dev(ops, w1, b3)), dev(ops, x8), residual=dev(ops, res), res_mode=K.IN_UPSAMPLE2)
    close(out, ref, 2e-5)
    # pixel-unshuffle + 1x1 (update.py:44-48)
    xu = rnd(B, 5, H, W, seed=13)
    wu, bu = rnd(7, 20, 1, 1, seed=14), rnd(7, seed=15)
    ref = F.conv2d(O._pixel_unshuffle(xu), wu, bu)
    out = ops.conv2d(K.pack_conv2d(*dev(ops, wu, bu)), dev(ops, xu), in_mode=K.IN_UNSHUFFLE2)
    close(out, ref, 2e-5)
    # relu(x + y) residual, post-scale, channel-offset output, NHWC output
    xr = rnd(B, 8, H, W, seed=16)
    ref = F.relu(F.conv2d(x8, w3[:8], None, 1, 1) + xr) * 0.25
    big = torch.full((B, 20, H, W), 7.0)
    bigd = dev(ops, big)
    ops.conv2d(K.pack_conv2d(dev(ops, w3[:8].contiguous()), pad=1), dev(ops, x8), residual=dev(ops, xr),
               act=K.ACT_RELU, post_scale=0.25, out=bigd, out_cstride=20, out_coffset=5)
    close(bigd[:, 5:13], ref, 2e-5)
    assert float(bigd[:, :5].min()) == 7.0 and float(bigd[:, 13:].min()) == 7.0
    out = ops.conv2d(K.pack_conv2d(dev(ops, w3[:8].contiguous()), pad=1), dev(ops, x8), out_layout=K.LAYOUT_NHWC)
    close(out, F.conv2d(x8, w3[:8], None, 1, 1).permute(0, 2, 3, 1).contiguous(), 2e-5)


@pytest.mark.parametrize("cin,cout,stride,transposed", [(4, 8, 1, False), (8, 16, 2, False), (16, 12, 1, False),
                                                         (16, 8, 2, True), (8, 1, 1, False), (12, 20, 2, True),
                                                         (20, 8, 2, False), (24, 16, 2, True)])
def test_conv3d(ops, cin, cout, stride, transposed):
    B, D, H, W = 2, 6, 7, 10
    if transposed:
        D, H, W = 3, 4, 5
    x = rnd(B, cin, D, H, W, seed=1)
    bn = {"weight": rnd(cout, seed=4, lo=0.5, hi=1.5), "bias": rnd(cout, seed=5),
          "running_mean": rnd(cout, seed=6), "running_var": rnd(cout, seed=7, lo=0.5, hi=1.5)}
    if transposed:
        w = rnd(cin, cout, 3, 3, 3, seed=2) * 0.2
        ref = F.conv_transpose3d(x, w, None, 2, 1, 1)
    else:
        w = rnd(cout, cin, 3, 3, 3, seed=2) * 0.2
        ref = F.conv3d(x, w, None, stride, 1)
    ref = F.relu(F.batch_norm(ref, bn["running_mean"], bn["running_var"], bn["weight"], bn["bias"], False, 0.0, 1e-5))
    res = rnd(*ref.shape, seed=9)
    ref = ref + res
    pc = K.pack_conv3d(dev(ops, w), bn={k_: v.to(ops.device) for k_, v in bn.items()}, stride=stride,
                       transposed=transposed)
    out = ops.conv3d(pc, dev(ops, x), act=K.ACT_RELU, residual=dev(ops, res))
    close(out, ref, 2e-5)


def _cams(B, V, H, W, seed):
    rs = np.random.RandomState(seed)
    pm = np.zeros((B, V, 2, 4, 4), np.float32)
    for b in range(B):
        for v in range(V):
            a = 0.06 * v + 0.01 * b
            R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
            E = np.eye(4)
            E[:3, :3] = R
            E[:3, 3] = [-25.0 * v, 3.0 * v, 2.0 * b]
            pm[b, v, 0] = E
            pm[b, v, 1, :3, :3] = [[1.1 * W, 0, W / 2 + rs.uniform(-1, 1)], [0, 1.1 * W, H / 2], [0, 0, 1]]
    return torch.from_numpy(pm)


def test_compose_proj(ops):
    pm = _cams(3, 4, 16, 20, 0)
    out = ops.compose_proj(dev(ops, pm)).cpu()
    ref_proj = O.compose_proj(pm[:, 0])
    for s in range(3):
        m = torch.matmul(O.compose_proj(pm[:, s + 1]).double(), torch.inverse(ref_proj.double()))
        close(out[:, s, :9].reshape(-1, 3, 3), m[:, :3, :3].float(), 1e-6)
        assert float((out[:, s, 9:] - m[:, :3, 3].float()).abs().max()) < 1e-3 * float(m[:, :3, 3].abs().max())


@pytest.mark.parametrize("C", [48, 32, 16])
def test_warp_corr_init(ops, C):
    B, S, D, H, W = 2, 3, 7, 11, 14
    pm = _cams(B, S + 1, H, W, 1)
    feats = [rnd(B, C, H, W, seed=10 + v) for v in range(S + 1)]
    dv = torch.tensor([[1 / 935.0, 1 / 425.0], [1 / 800.0, 1 / 500.0]])
    disp_min, disp_max = 1 / (1 / dv[:, 0]), 1 / (1 / dv[:, 1])
    hyp = (torch.arange(D).view(1, -1, 1, 1) / (D - 1.0)).repeat(B, 1, H, W)
    hyp = O.disp_to_depth(hyp, (1 / dv[:, 1]).view(-1, 1, 1, 1), (1 / dv[:, 0]).view(-1, 1, 1, 1))[1]
    ref_proj = O.compose_proj(pm[:, 0])
    want = torch.stack([O.group_corr(O.warp(feats[v], O.compose_proj(pm[:, v]), ref_proj, hyp), feats[0], 4)
                        for v in range(1, S + 1)], 1)
    rt = ops.compose_proj(dev(ops, pm))
    outs = []
    for plain in (True, False):         # plain NHWC fp32 (training graph) and NHWC-g4 (inference engine)
        ref_f, src_f = _warp_feats(ops, feats, plain)
        for tune in (0, K._lib.TUNE_SWEEP_GLOBAL):      # LDS band / every texel from global memory
            outs.append(ops.warp_corr_init_quad(ref_f, src_f, rt, dev(ops, disp_min), dev(ops, disp_max), D, plain=plain, tune=tune))
            close(outs[-1], want, 1e-4)
    for o_ in outs[1:]:
        assert torch.equal(o_.cpu(), outs[0].cpu())      # same arithmetic, other address pattern


@pytest.mark.parametrize("H,W,D,scene", [(40, 56, 16, True), (36, 30, 48, True), (24, 40, 9, False)])
def test_warp_corr_init_plain_features_tiles(ops, H, W, D, scene):
    """stage-1 plane sweep on the TRAINING graph's plain NHWC fp32 features (C = 48): several band tiles incl. partial ones, plane
    groups; `scene` uses the synthetic cameras (bands fit), else strongly rotated cameras (groups fall back to global memory).
    Against the oracle and bit for bit against the global-memory form."""
    B, S, C = 2, 3, 48
    feats = [rnd(B, C, H, W, seed=80 + v) for v in range(S + 1)]
    if scene:
        _, proj, dvs = synth.synth_inputs(H * 8, W * 8, S, B=B, seed=7)
        pm = proj["stage1"]
        dv = torch.stack([dvs[:, 0], dvs[:, -1]], 1)
    else:
        pm = _cams(B, S + 1, H, W, 9)
        dv = torch.tensor([[1 / 935.0, 1 / 425.0], [1 / 800.0, 1 / 500.0]])
    disp_min, disp_max = dv[:, 0].contiguous(), dv[:, 1].contiguous()
    hyp = (torch.arange(D).view(1, -1, 1, 1) / (D - 1.0)).repeat(B, 1, H, W)
    hyp = O.disp_to_depth(hyp, (1 / dv[:, 1]).view(-1, 1, 1, 1), (1 / dv[:, 0]).view(-1, 1, 1, 1))[1]
    ref_proj = O.compose_proj(pm[:, 0])
    want = torch.stack([O.group_corr(O.warp(feats[v], O.compose_proj(pm[:, v]), ref_proj, hyp), feats[0], 4)
                        for v in range(1, S + 1)], 1)
    rt = ops.compose_proj(dev(ops, pm))
    ref_nhwc, src_nhwc = _warp_feats(ops, feats, True)
    out = ops.warp_corr_init_quad(ref_nhwc, src_nhwc, rt, dev(ops, disp_min), dev(ops, disp_max), D, plain=True)
    out_g = ops.warp_corr_init_quad(ref_nhwc, src_nhwc, rt, dev(ops, disp_min), dev(ops, disp_max), D, plain=True, tune=K._lib.TUNE_SWEEP_GLOBAL)
    close(out, want, 1e-4)
    assert torch.equal(out.cpu(), out_g.cpu())


def test_warp_golden_edge_cases(ops, golden):
    """differentiable_warping edge cases recorded from the reference (big rotations, points behind
    the camera, z == 0, source grid != hypothesis grid), pushed through the fused kernel by using
    an all-ones reference feature: cor[g] = mean over group of warped."""
    g = golden("warp_edge.npz")
    for ci in range(int(g.np("n_cases"))):
        src, depth, want = g.t(f"c{ci}.src"), g.t(f"c{ci}.depth"), g.t(f"c{ci}.out")
        B, Cc, Hs, Ws = src.shape
        D, H, W = depth.shape[1:]
        # the fused kernel generates uniform inverse-depth hypotheses itself; per-pixel depth maps go
        # through getcost-style sampling, so here we test each depth plane d separately via D=2
        # kernels is not possible -> use the generic C=16 path with channel padding and constant depth
        # planes: only cases whose depth is constant per plane are comparable; others are covered by
        # test_getcost below.  Case 4 (z==0) has constant planes.
        if ci != 4:
            continue
        pad = 16 - Cc
        srcp = torch.cat([src, torch.zeros(B, pad, Hs, Ws)], 1)
        P = torch.matmul(g.t(f"c{ci}.src_proj"), torch.inverse(g.t(f"c{ci}.ref_proj")))
        rt = torch.cat([P[:, :3, :3].reshape(B, 9), P[:, :3, 3]], 1).view(B, 1, 12)
        d0, d1 = float(depth[0, 0, 0, 0]), float(depth[0, 1, 0, 0])
        out = ops.warp_corr_init_quad(dev(ops, torch.ones(B, H, W, 16)), dev(ops, srcp.permute(0, 2, 3, 1).unsqueeze(0).contiguous()),
                                      dev(ops, rt), dev(ops, torch.tensor([1 / d1])), dev(ops, torch.tensor([1 / d0])), 2, plain=True)
        # hypothesis 0 = disp_min -> depth d1 ; hypothesis 1 = disp_max -> depth d0
        got = out.cpu()[:, 0, 0] * 4.0   # group 0 = channels 0..3 = the real channels, mean -> sum
        want_g = want.sum(1)             # [B,D,H,W]
        close(got[:, 1], want_g[:, 0], 1e-4)
        close(got[:, 0], want_g[:, 1], 1e-4)


@pytest.mark.parametrize("C,n,with_conf", [(32, 6, False), (32, 6, True), (16, 4, True), (48, 4, False)])
def test_getcost(ops, C, n, with_conf):
    B, S, H, W = 2, 3, 12, 16
    pm = _cams(B, S + 1, H, W, 2)
    feats = [rnd(B, C, H, W, seed=20 + v) for v in range(S + 1)]
    inv = rnd(B, 1, H, W, seed=30, lo=-0.1, hi=1.1)      # some hypotheses get clamped
    conf = rnd(B, H, W, seed=31, lo=0.0, hi=1.0) if with_conf else None
    vw = rnd(B, S, H // 2, W // 2, seed=32, lo=0.0, hi=1.0)
    dv0, dv1 = torch.tensor([1 / 935.0, 1 / 700.0]), torch.tensor([1 / 425.0, 1 / 450.0])
    dmax, dmin = (1 / dv0).view(-1, 1, 1, 1), (1 / dv1).view(-1, 1, 1, 1)
    interval = 2.0 / 384
    want_cost, want_s = O.get_cost(feats, pm, inv, interval, dmax, dmin, n,
                                   F.interpolate(vw, scale_factor=2, mode="nearest"), conf, 4, 0.25, 4.0)
    rt = ops.compose_proj(dev(ops, pm))
    costs = []
    for plain in (True, False):         # plain NHWC fp32 (training graph) and NHWC-g4 (inference engine)
        ref_f, src_f = _warp_feats(ops, feats, plain)
        cost, samp = ops.getcost_quad(ref_f, src_f, rt, dev(ops, inv), dev(ops, conf), dev(ops, vw), dev(ops, 1 / (1 / dv0)),
                                      dev(ops, 1 / (1 / dv1)), n, interval, 0.25, 4.0, vw_shift=1, plain=plain)
        close(samp, want_s, 1e-6)
        close(cost, want_cost, 1e-4)
        costs.append(cost.cpu())
    assert torch.equal(costs[0], costs[1])


@pytest.mark.parametrize("C,n,interval,H,W", [(32, 6, 2.0 / 384, 40, 56), (16, 4, 1.0 / 384, 36, 50), (32, 6, 0.15, 24, 40),
                                              (16, 4, 0.3, 20, 36)])
def test_getcost_plain_features_tiles(ops, C, n, interval, H, W):
    """GetCost on the training graph's plain NHWC fp32 features: several workgroups incl. partial ones; the large intervals spread a
    pixel's hypotheses beyond the 8x8 texel grid (one chunk per hypothesis).  Against the oracle and the NHWC-g4 form."""
    B, S = 2, 3
    pm = _cams(B, S + 1, H, W, 2)
    feats = [rnd(B, C, H, W, seed=40 + v) for v in range(S + 1)]
    inv = rnd(B, 1, H, W, seed=50, lo=-0.05, hi=1.05)
    conf = rnd(B, H, W, seed=51, lo=0.0, hi=1.0)
    vw = rnd(B, S, H // 2, W // 2, seed=52, lo=0.0, hi=1.0)
    dv0, dv1 = torch.tensor([1 / 935.0, 1 / 700.0]), torch.tensor([1 / 425.0, 1 / 450.0])
    dmax, dmin = (1 / dv0).view(-1, 1, 1, 1), (1 / dv1).view(-1, 1, 1, 1)
    want_cost, want_s = O.get_cost(feats, pm, inv, interval, dmax, dmin, n,
                                   F.interpolate(vw, scale_factor=2, mode="nearest"), conf, 4, 0.25, 4.0)
    rt = ops.compose_proj(dev(ops, pm))
    tail = (rt, dev(ops, inv), dev(ops, conf), dev(ops, vw), dev(ops, 1 / (1 / dv0)), dev(ops, 1 / (1 / dv1)), n, interval, 0.25, 4.0)
    cost, samp = ops.getcost_quad(*_warp_feats(ops, feats, True), *tail, vw_shift=1, plain=True)
    cost_g, samp_g = ops.getcost_quad(*_warp_feats(ops, feats, False), *tail, vw_shift=1)
    close(samp, want_s, 1e-6)
    close(cost, want_cost, 1e-4)
    assert torch.equal(cost.cpu(), cost_g.cpu())


@pytest.mark.parametrize("S", [16, 17])
def test_getcost_many_source_views(ops, S):
    """16 source views = the most the BACKWARD's window kernel keeps footprint boxes for; 17 must take the per-pixel backward kernel
    instead of overrunning them -- results identical either way; the quad forward has no such limit."""
    B, C, n, H, W = 1, 32, 6, 24, 40
    pm = _cams(B, S + 1, H, W, 5)
    feats = [rnd(B, C, H, W, seed=60 + v) for v in range(S + 1)]
    inv = rnd(B, 1, H, W, seed=70, lo=0.3, hi=0.7)
    vw = rnd(B, S, H // 2, W // 2, seed=72, lo=0.0, hi=1.0)
    dv0, dv1 = torch.tensor([1 / 935.0]), torch.tensor([1 / 425.0])
    dmax, dmin = (1 / dv0).view(-1, 1, 1, 1), (1 / dv1).view(-1, 1, 1, 1)
    want_cost, want_s = O.get_cost(feats, pm, inv, 0.004, dmax, dmin, n, F.interpolate(vw, scale_factor=2, mode="nearest"),
                                   None, 4, 0.25, 4.0)
    rt = ops.compose_proj(dev(ops, pm))
    args = (dev(ops, feats[0].permute(0, 2, 3, 1)), dev(ops, torch.stack([f.permute(0, 2, 3, 1) for f in feats[1:]])), rt,
            dev(ops, inv), None, dev(ops, vw), dev(ops, 1 / (1 / dv0)), dev(ops, 1 / (1 / dv1)), n, 0.004, 0.25, 4.0)
    cost, samp = ops.getcost_quad(*args, vw_shift=1, plain=True)
    close(samp, want_s, 1e-6)
    close(cost, want_cost, 1e-4)
    gcost = dev(ops, rnd(B, 4 * n, H, W, seed=75))
    gref, gsrc = ops.getcost_bwd(*args, vw_shift=1, gcost=gcost)
    gref_g, gsrc_g = ops.getcost_bwd(*args, vw_shift=1, gcost=gcost, gather=True)
    close(gref, gref_g.cpu(), 1e-5)
    close(gsrc, gsrc_g.cpu(), 1e-5)


def test_getcost_extreme_geometry(ops, golden):
    """per-pixel depth maps + the reference's own warping edge cases pushed through the fused GetCost kernel on plain NHWC fp32
    features (the training graph's; test_getcost_quad_extreme_geometry does the same in the NHWC-g4 order): case 0 mild, 1 large rotation (big out-of-bounds
    regions), 2 camera looking backwards (negative z), 3 source grid != hypothesis grid.  (Case 4, exact z == 0 on
    constant planes, goes through the stage-1 kernel in test_warp_golden_edge_cases.)  The kernels' contract is one
    H x W grid for reference and source, so case 3 embeds both in a common canvas: zero texels beyond the source ARE
    grid_sample's zero padding, and the extra reference pixels are not compared."""
    g = golden("warp_edge.npz")
    for ci in (0, 1, 2, 3):
        src, depth, want = g.t(f"c{ci}.src"), g.t(f"c{ci}.depth"), g.t(f"c{ci}.out")
        B, Cc, Hs, Ws = src.shape
        D, H, W = depth.shape[1:]
        Hc, Wc = max(H, Hs), max(W, Ws)
        srcp = torch.zeros(B, 16, Hc, Wc)
        srcp[:, :Cc, :Hs, :Ws] = src
        P = torch.matmul(g.t(f"c{ci}.src_proj"), torch.inverse(g.t(f"c{ci}.ref_proj")))
        rt = torch.cat([P[:, :3, :3].reshape(B, 9), P[:, :3, 3]], 1).view(B, 1, 12)
        lo, hi = torch.full((B,), 1 / 2000.0), torch.full((B,), 1 / 100.0)
        for d in range(D):
            dpl = torch.full((B, 1, Hc, Wc), 500.0)
            dpl[:, :, :H, :W] = depth[:, d:d + 1]
            # zero radius: all 4 hypotheses sit on the recorded depth (up to the inverse-depth round trip)
            inv = ((1 / dpl) - lo.view(-1, 1, 1, 1)) / (hi - lo).view(-1, 1, 1, 1)
            w = want[:, :, d].sum(1)
            cost, samp = ops.getcost_quad(dev(ops, torch.ones(B, Hc, Wc, 16)), dev(ops, srcp.permute(0, 2, 3, 1).unsqueeze(0).contiguous()),
                                          dev(ops, rt), dev(ops, inv.contiguous()), None, dev(ops, torch.ones(B, 1, Hc, Wc)),
                                          dev(ops, lo), dev(ops, hi), 4, 0.0, 1.0, 1.0, vw_shift=0, plain=True)
            cost = cost.cpu()
            got = (cost[:, 0] + cost[:, 4] + cost[:, 8] + cost[:, 12])[:, :H, :W] * 4.0     # hypothesis 0 of the 4 groups: mean -> sum
            # inverse-depth round trip perturbs depth by ~1e-4 relative; compare loosely but everywhere
            assert float((got - w).abs().mean()) <= 2e-3 * max(1.0, float(w.abs().mean())), (ci, d)


def test_misc_kernels(ops):
    B, S, G, D, H, W = 2, 3, 4, 6, 5, 7
    cor, w = rnd(B, S, G, D, H, W, seed=1), rnd(B, S, H, W, seed=2, lo=0, hi=1)
    want = (cor * w.view(B, S, 1, 1, H, W)).sum(1) / (1e-8 + w.sum(1)).view(B, 1, 1, H, W)
    close(ops.view_aggregate(*dev(ops, cor, w)), want, 1e-5)
    # planes of 16-byte multiples take the 16-byte form (a lane = 4 pixels x a strip of (group, depth) planes): same operations in the
    # same order as the element-wise kernel, which a view of the same data at a 4-byte offset still takes
    for S2, H2, W2 in ((5, 6, 10), (2, 4, 4), (11, 3, 8)):
        cor2, w2 = rnd(B, S2, G, 19, H2, W2, seed=11), rnd(B, S2, H2, W2, seed=12, lo=0, hi=1)
        want2 = (cor2 * w2.view(B, S2, 1, 1, H2, W2)).sum(1) / (1e-8 + w2.sum(1)).view(B, 1, 1, H2, W2)
        a = ops.view_aggregate(*dev(ops, cor2, w2))
        close(a, want2, 1e-5)
        off = dev(ops, torch.cat([torch.zeros(1), cor2.flatten()]))[1:].view(cor2.shape)      # 4-byte offset: element-wise kernel
        assert torch.equal(a.cpu(), ops.view_aggregate(off, dev(ops, w2)).cpu())
    x = rnd(B * S, D, H, W, seed=3) * 3
    close(ops.sigmoid_max_d(dev(ops, x)), torch.sigmoid(x).max(1)[0], 1e-6)
    # depth regression (module.py:553-571)
    logits = rnd(B, D, H, W, seed=4) * 4
    lo, hi = torch.tensor([1 / 935.0, 1 / 800.0]), torch.tensor([1 / 425.0, 1 / 500.0])
    prob = F.softmax(logits, 1)
    index = (torch.arange(D).view(1, D, 1, 1) * prob).sum(1, keepdim=True)
    nd = index / (D - 1.0)
    depth = O.disp_to_depth(nd, (1 / hi).view(-1, 1, 1, 1), (1 / lo).view(-1, 1, 1, 1))[1].squeeze(1)
    padded = F.pad(prob, (0, 0, 0, 0, 1, 2))
    sum4 = padded[:, 0:D] + padded[:, 1:D + 1] + padded[:, 2:D + 2] + padded[:, 3:D + 3]
    conf = torch.gather(sum4, 1, index.long().clamp(0, D - 1))
    g_nd, g_depth, g_conf = ops.depth_regress(*dev(ops, logits, 1 / (1 / lo), 1 / (1 / hi)))
    close(g_nd, nd, 1e-5)
    close(g_depth, depth, 1e-5)
    assert float(((g_conf.cpu() - conf).abs() > 1e-4).float().mean()) < 0.05
    # convex upsampling (module.py:237-248) for both ratios
    for r in (2, 4):
        inv, mask = rnd(B, 1, H, W, seed=5, lo=0, hi=1), rnd(B, 9 * r * r, H, W, seed=6) * 2
        up = O.upsample_depth(inv, mask, r)
        g_inv, g_depth = ops.convex_upsample(*dev(ops, inv, mask, 1 / (1 / lo), 1 / (1 / hi)), r)
        close(g_inv, up, 1e-5)
        close(g_depth, O.disp_to_depth(up.unsqueeze(1), (1 / hi).view(-1, 1, 1, 1), (1 / lo).view(-1, 1, 1, 1))[1].squeeze(1), 1e-5)
    # depth <-> disp
    dep = rnd(B, 1, H, W, seed=7, lo=430, hi=900)
    close(ops.depth_convert(*dev(ops, dep, 1 / (1 / lo), 1 / (1 / hi)), 0),
          O.depth_to_disp(dep, (1 / hi).view(-1, 1, 1, 1), (1 / lo).view(-1, 1, 1, 1)), 1e-5)
    # refinement bookkeeping
    inv, dl, upd = rnd(B, 1, H, W, seed=8, lo=0, hi=1), rnd(B, 1, H, W, seed=9), rnd(B, 1, H, W, seed=10)
    new = (inv + 0.5 * dl + upd).clamp(0, 1)
    big = dev(ops, torch.zeros(B, 3, H, W))
    g_delta, g_new = ops.delta_update(*dev(ops, inv, dl, upd), 0.5, new2=big, new2_cstride=3, new2_coffset=2)
    close(g_new, new, 1e-6)
    close(g_delta, new - inv, 1e-6)
    close(big[:, 2:3], new, 1e-6)
    g_delta, g_new = ops.delta_update(*dev(ops, inv, dl), None, 0.5)
    close(g_new, (inv + 0.5 * dl).clamp(0, 1), 1e-6)
    # slices / resampling / layout
    x = rnd(B, 9, H, W, seed=11)
    close(ops.act_slice(dev(ops, x), K.ACT_TANH, 2, 4), torch.tanh(x[:, 2:6]), 1e-6)
    close(ops.upsample_nearest(dev(ops, x), 4), F.interpolate(x, scale_factor=4, mode="nearest"), 0)
    for f, xs in ((2, x), (8, x), (3, x), (2, x[..., :7].contiguous())):      # 16-byte rows (4 outputs per lane) and the ragged fallback (odd row lengths)
        close(ops.upsample_nearest(dev(ops, xs), f), F.interpolate(xs, scale_factor=f, mode="nearest"), 0)
    close(ops.nchw_to_nhwc(dev(ops, x)), x.permute(0, 2, 3, 1).contiguous(), 0)


@pytest.mark.parametrize("C,HW", [(16, (9, 13)), (32, (40, 70))])
def test_groupnorm_silu(ops, C, HW):
    B = 2
    x = rnd(B, C, *HW, seed=1) * 2 + 0.3
    gamma, beta = rnd(C, seed=2, lo=0.5, hi=1.5), rnd(C, seed=3)
    ss, res = rnd(B, 2 * C, seed=4), rnd(B, C, *HW, seed=5)
    y = F.group_norm(x, 4, gamma, beta, 1e-5)
    want = F.silu(y * (ss[:, :C, None, None] + 1) + ss[:, C:, None, None]) + res
    close(ops.groupnorm_silu(*dev(ops, x, gamma, beta), 4, scale_shift=dev(ops, ss), residual=dev(ops, res)), want, 2e-5)
    close(ops.groupnorm_silu(*dev(ops, x, gamma, beta), 4), F.silu(y), 2e-5)


@pytest.mark.parametrize("cin,cout,HW", [(16, 16, (21, 37)), (48, 32, (16, 16)), (8, 8, (9, 50))])
def test_conv_fused_groupnorm_stats(ops, cin, cout, HW):
    """WS-conv -> GroupNorm(4) -> scale/shift -> SiLU with the statistics taken in the conv epilogue
    (reference models/update.py:124-133)."""
    B = 2
    x = rnd(B, cin, *HW, seed=1)
    w, bias = rnd(cout, cin, 3, 3, seed=2) * 0.3, rnd(cout, seed=3)
    gamma, beta, ss = rnd(cout, seed=4, lo=0.5, hi=1.5), rnd(cout, seed=5), rnd(B, 2 * cout, seed=6)
    y = F.conv2d(x, w, bias, 1, 1)
    want = F.silu(F.group_norm(y, 4, gamma, beta, 1e-5) * (ss[:, :cout, None, None] + 1) + ss[:, cout:, None, None])
    stats = torch.zeros(B * 8, dtype=torch.float64, device=ops.device)
    pc = K.pack_conv2d(*dev(ops, w, bias), pad=1)
    h = ops.conv2d(pc, dev(ops, x), gn_stats=stats)
    close(h, y, 2e-5)
    n = (cout // 4) * HW[0] * HW[1]
    st = (stats.cpu().view(torch.int64).double() / 65536.0).view(B, 4, 2)      # opaque slots: 2^-16 fixed point (dmvs_common.h)
    yg = y.view(B, 4, -1).double()
    assert torch.allclose(st[..., 0] / n, yg.mean(-1), atol=1e-5)
    assert torch.allclose(st[..., 1] / n, (yg * yg).mean(-1), rtol=1e-5, atol=1e-6)
    out = ops.groupnorm_apply(h, *dev(ops, gamma, beta), 4, stats, scale_shift=dev(ops, ss), out=h)
    close(out, want, 3e-5)


@pytest.mark.parametrize("cin,cout,k,stride,HW,extras", [
    (16, 16, (3, 3), 1, (21, 37), "plain"), (28, 31, (3, 3), 1, (16, 33), "res_relu"), (8, 16, (5, 5), 2, (37, 50), "bn"),
    (32, 64, (5, 5), 2, (20, 24), "bn"), (64, 16, (7, 7), 1, (18, 20), "plain"), (64, 64, (1, 5), 1, (9, 40), "gru"),
    (64, 32, (5, 1), 1, (12, 17), "plain"), (32, 32, (3, 3), 2, (22, 30), "plain"), (24, 48, (3, 3), 1, (8, 8), "concat_gn"),
    (37, 20, (3, 3), 1, (40, 70), "plain")])
def test_conv2d_bf16_matrix_arithmetic(ops, cin, cout, k, stride, HW, extras):
    """arith = ARITH_BF16: inputs and weights rounded to bf16 (nearest even) on their way into the matrix cores, fp32
    accumulation, fp32 tensors and epilogue -- against torch's fp32 convolution of the ROUNDED operands (exact products, only
    the summation order differs), for every kernel shape of the networks, channel counts that are not multiples of the
    8-channel chunks, and the fused staging / epilogue paths (concat, GRU gating + blend, residual, BN, GroupNorm statistics).
    The fp32 result of the same call differs by the bf16 rounding (~4e-3 relative), i.e. the mode is really on.  Layers the
    contract exempts (stride 2, fewer than 24 input channels: include/dmvs.h) return the fp32 result in either mode."""
    B = 2
    kh, kw = k
    pad = (kh // 2, kw // 2)
    x = rnd(B, cin, *HW, seed=1)
    w = rnd(cout, cin, kh, kw, seed=2) * (2.0 / (cin * kh * kw) ** 0.5)
    rb = lambda t: t.to(torch.bfloat16).float()      # noqa: E731
    kw_call, ref_in, x0, x1 = {}, x, x, None
    bn = None
    if extras == "bn":
        bn = {"weight": rnd(cout, seed=4, lo=0.5, hi=1.5), "bias": rnd(cout, seed=5), "running_mean": rnd(cout, seed=6),
              "running_var": rnd(cout, seed=7, lo=0.5, hi=1.5)}
    if extras == "concat_gn":
        x0, x1 = x[:, :10].contiguous(), x[:, 10:].contiguous()
    if extras == "gru":      # candidate conv of SepConvGRU: input cat(r * h, x), output blended with z and h
        r = rnd(B, 32, *HW, seed=11, lo=0.0, hi=1.0)
        h = x[:, :32].contiguous()
        x0, x1 = h, x[:, 32:].contiguous()
        ref_in = torch.cat([r * h, x[:, 32:]], 1)
        z, hh = rnd(B, cout, *HW, seed=12, lo=0.0, hi=1.0), rnd(B, cout, *HW, seed=13)
    honoured = stride == 1 and cin >= 24
    ref = F.conv2d(rb(ref_in), rb(w), None, stride, pad) if honoured else F.conv2d(ref_in, w, None, stride, pad)
    full = F.conv2d(ref_in, w, None, stride, pad)
    if bn is not None:
        f = lambda t: F.relu(F.batch_norm(t, bn["running_mean"], bn["running_var"], bn["weight"], bn["bias"], False, 0.0, 1e-5))      # noqa: E731
        ref, full = f(ref), f(full)
        kw_call["act"] = K.ACT_RELU
    if extras == "res_relu":
        res = rnd(*ref.shape, seed=9)
        ref, full = F.relu(ref + res), F.relu(full + res)
        kw_call.update(residual=dev(ops, res), act=K.ACT_RELU)
    if extras == "gru":
        ref, full = (1 - z) * hh + z * torch.tanh(ref), (1 - z) * hh + z * torch.tanh(full)
        kw_call.update(mul0=dev(ops, r), gru_z=dev(ops, z), gru_h=dev(ops, hh), act=K.ACT_TANH)
    stats = None
    if extras == "concat_gn":
        stats = torch.zeros(B * 8, dtype=torch.float64, device=ops.device)
        kw_call.update(gn_stats=stats)
    pc = K.pack_conv2d(dev(ops, w), bn=None if bn is None else {k_: v.to(ops.device) for k_, v in bn.items()}, stride=stride, pad=pad)
    out = ops.conv2d(pc, dev(ops, x0), None if x1 is None else dev(ops, x1), arith=K.ARITH_BF16, **kw_call)
    close(out, ref, 2e-5)
    if honoured:
        assert float((out.cpu() - full).abs().max()) > 1e-4 * float(full.abs().max())       # not the fp32 kernel
    out32 = ops.conv2d(pc, dev(ops, x0), None if x1 is None else dev(ops, x1), **{k_: v for k_, v in kw_call.items() if k_ != "gn_stats"})
    close(out32, full, 2e-5)
    if stats is not None:
        st = (stats.cpu().view(torch.int64).double() / 65536.0).view(B, 4, 2)
        yg = ref.view(B, 4, -1).double()
        assert torch.allclose(st[..., 0] / yg.shape[-1], yg.mean(-1), atol=1e-4)


@pytest.mark.parametrize("cin,cout,k,stride,HW,extras", [
    (16, 16, (3, 3), 1, (24, 36), "plain"), (28, 31, (3, 3), 1, (16, 32), "res_relu"), (8, 16, (5, 5), 2, (36, 52), "bn"),
    (32, 64, (5, 5), 2, (20, 24), "bn"), (32, 16, (7, 7), 1, (18, 20), "plain"), (64, 64, (1, 5), 1, (9, 40), "gru"),
    (64, 32, (5, 1), 1, (12, 16), "plain"), (32, 32, (3, 3), 2, (22, 32), "plain"), (24, 48, (3, 3), 1, (8, 8), "concat_gn"),
    (37, 20, (3, 3), 1, (40, 72), "plain"), (3, 8, (3, 3), 1, (33, 40), "bn"), (16, 16, (3, 3), 1, (70, 36), "mt4"),
    (64, 32, (3, 3), 1, (24, 40), "nhwc"), (20, 16, (3, 3), 1, (37, 36), "nhwc")])
def test_conv2d_split_bf16_arithmetic(ops, cin, cout, k, stride, HW, extras):
    """arith = ARITH_SPLIT: every fp32 operand split exactly into three bf16 values, six partial products per product on the bf16 matrix
    cores, fp32 accumulation -- fp32 ACCURACY, not fp32 bits.  Against torch's convolution in fp64: the split kernel's error is of the size
    of the exact-fp32 kernel's own rounding error (both a few 1e-7 of the output scale; bf16 arithmetic would be 4e-3), for every kernel
    shape, channel counts that are not multiples of the 8-channel chunks, both tile heights, and the fused staging / epilogue paths (concat,
    GRU gating + blend, residual, BN, GroupNorm statistics)."""
    B = 2
    kh, kw = k
    pad = (kh // 2, kw // 2)
    x = rnd(B, cin, *HW, seed=1)
    w = rnd(cout, cin, kh, kw, seed=2) * (2.0 / (cin * kh * kw) ** 0.5)
    kw_call, ref_in, x0, x1 = {}, x, x, None
    bn = None
    if extras == "bn":
        bn = {"weight": rnd(cout, seed=4, lo=0.5, hi=1.5), "bias": rnd(cout, seed=5), "running_mean": rnd(cout, seed=6),
              "running_var": rnd(cout, seed=7, lo=0.5, hi=1.5)}
    if extras == "concat_gn":
        x0, x1 = x[:, :10].contiguous(), x[:, 10:].contiguous()
    if extras == "gru":      # candidate conv of SepConvGRU: input cat(r * h, x), output blended with z and h
        r = rnd(B, 32, *HW, seed=11, lo=0.0, hi=1.0)
        h = x[:, :32].contiguous()
        x0, x1 = h, x[:, 32:].contiguous()
        ref_in = torch.cat([r * h, x[:, 32:]], 1)
        z, hh = rnd(B, cout, *HW, seed=12, lo=0.0, hi=1.0), rnd(B, cout, *HW, seed=13)
    ref = F.conv2d(ref_in.double(), w.double(), None, stride, pad)
    if bn is not None:
        ref = F.relu(F.batch_norm(ref, bn["running_mean"].double(), bn["running_var"].double(), bn["weight"].double(), bn["bias"].double(), False, 0.0, 1e-5))
        kw_call["act"] = K.ACT_RELU
    if extras == "res_relu":
        res = rnd(*ref.shape, seed=9)
        ref = F.relu(ref + res.double())
        kw_call.update(residual=dev(ops, res), act=K.ACT_RELU)
    if extras == "gru":
        ref = (1 - z.double()) * hh.double() + z.double() * torch.tanh(ref)
        kw_call.update(mul0=dev(ops, r), gru_z=dev(ops, z), gru_h=dev(ops, hh), act=K.ACT_TANH)
    stats = None
    if extras == "concat_gn":
        stats = torch.zeros(B * 8, dtype=torch.float64, device=ops.device)
        kw_call.update(gn_stats=stats)
    kw_call["tune"] = K._lib.TUNE_SPLIT_ALL | (K._lib.tune_tile_mt(4) if extras == "mt4" else 0)      # (the dispatcher's own rule takes the form only where it measured faster)
    if extras == "nhwc":      # FeatureNet's channel-last feature outputs
        kw_call["out_layout"] = K.LAYOUT_NHWC
        ref = ref.permute(0, 2, 3, 1)
    pc = K.pack_conv2d(dev(ops, w), bn=None if bn is None else {k_: v.to(ops.device) for k_, v in bn.items()}, stride=stride, pad=pad)
    out = ops.conv2d(pc, dev(ops, x0), None if x1 is None else dev(ops, x1), arith=K.ARITH_SPLIT, **kw_call).cpu().double()
    kw_call.pop("tune")
    out32 = ops.conv2d(pc, dev(ops, x0), None if x1 is None else dev(ops, x1), **{k_: v for k_, v in kw_call.items() if k_ != "gn_stats"}).cpu().double()
    scale = float(ref.abs().max())
    e_split, e_f32 = float((out - ref).abs().max()) / scale, float((out32 - ref).abs().max()) / scale
    print("split vs fp64: %.2e   exact fp32 vs fp64: %.2e" % (e_split, e_f32))
    # the same class of error as the fma chain's own rounding.  (The host emulation sums the 32 products of a bf16 matrix instruction one by one
    # in fp32 -- the instruction's internal order is not architecturally specified -- which overstates the error of its six accumulations per 32
    # channels x taps: up to 2.4x the fma chain's there; the GPU run of this test is the measurement.)
    assert e_f32 < 2e-6 and e_split < 4e-6, (e_split, e_f32)
    assert e_split < 4.0 * e_f32 + 2e-7, (e_split, e_f32)
    if stats is not None:
        st = (stats.cpu().view(torch.int64).double() / 65536.0).view(B, 4, 2)
        yg = ref.view(B, 4, -1)
        assert torch.allclose(st[..., 0] / yg.shape[-1], yg.mean(-1), atol=1e-4)


@pytest.mark.parametrize("cin,cout,k,stride,HW,with_res", [(16, 16, 3, 1, (40, 52), False), (32, 32, 3, 1, (37, 36), True), (8, 16, 5, 2, (50, 72), False),
                                                          (16, 32, 3, 2, (44, 40), False), (12, 20, 3, 1, (17, 20), True), (16, 16, 3, 1, (128, 176), False)])
def test_conv2d_tile_walking_workgroups(ops, cin, cout, k, stride, HW, with_res):
    """"lean" layers (one plain input, BN + ReLU, optional residual, rows of 16-byte multiples) run on resident, tile-walking
    workgroups: several tiles per workgroup (the emulation keeps 2 workgroups resident, the last case has more tiles than an
    MI355X holds), interior tiles after border tiles and back (stale padding), partial last chunks, both tile widths."""
    B = 3
    x = rnd(B, cin, *HW, seed=1)
    w = rnd(cout, cin, k, k, seed=2) * 0.2
    bn = {"weight": rnd(cout, seed=4, lo=0.5, hi=1.5), "bias": rnd(cout, seed=5), "running_mean": rnd(cout, seed=6),
          "running_var": rnd(cout, seed=7, lo=0.5, hi=1.5)}
    ref = F.batch_norm(F.conv2d(x, w, None, stride, k // 2), bn["running_mean"], bn["running_var"], bn["weight"], bn["bias"], False, 0.0, 1e-5)
    res = rnd(*ref.shape, seed=9) if with_res else None
    if with_res:
        ref = ref + res
    ref = F.relu(ref)
    pc = K.pack_conv2d(dev(ops, w), bn={k_: v.to(ops.device) for k_, v in bn.items()}, stride=stride, pad=k // 2)
    out = ops.conv2d(pc, dev(ops, x), act=K.ACT_RELU, residual=dev(ops, res))
    close(out, ref, 2e-5)


@pytest.mark.parametrize("cin,c1,cout,HW,extra", [(16, 0, 16, (64, 48), "res"), (16, 0, 16, (40, 52), None), (12, 4, 16, (33, 36), "gn"), (32, 0, 13, (70, 20), None),
                                                   (16, 0, 32, (64, 48), "res"), (24, 8, 32, (37, 36), "gn"), (32, 0, 31, (40, 20), None), (16, 0, 16, (130, 36), None)])
def test_conv2d_tall_tiles(ops, cin, c1, cout, HW, extra):
    """DMVS_TUNE_TALL: the plain 3x3 layers on 16 x 32-pixel tiles (MT = 8, 4-channel chunks; the default at large batches): ragged last tile rows, border tiles, a second concat input, the residual and the GroupNorm
    statistics -- against torch and BIT FOR BIT the 16 x 16 / 16 x 4 tiles"""
    B = 2
    x0 = rnd(B, cin, *HW, seed=1)
    x1 = rnd(B, c1, *HW, seed=2) if c1 else None
    w, bias = rnd(cout, cin + c1, 3, 3, seed=3) * 0.2, rnd(cout, seed=4)
    ref = F.conv2d(x0 if x1 is None else torch.cat([x0, x1], 1), w, bias, 1, 1)
    res = rnd(*ref.shape, seed=5) if extra == "res" else None
    pc = K.pack_conv2d(*dev(ops, w, bias), pad=1)
    outs, sts = [], []
    for tune in (K._lib.TUNE_NO_TALL, K._lib.TUNE_TALL):
        stats = torch.zeros(B * 8, dtype=torch.float64, device=ops.device) if extra == "gn" else None
        out = ops.conv2d(pc, dev(ops, x0), None if x1 is None else dev(ops, x1), residual=None if res is None else dev(ops, res),
                         act=K.ACT_NONE if extra == "gn" else K.ACT_RELU, gn_stats=stats, tune=tune)
        close(out, ref if extra == "gn" else F.relu(ref + res if res is not None else ref), 2e-5)
        outs.append(out.cpu())
        sts.append(None if stats is None else stats.cpu().view(torch.int64))
    assert torch.equal(outs[0], outs[1])
    if extra == "gn":      # fixed-point sums of float partials: the partials differ with the tile shape, the totals agree to rounding
        a, b = sts[0].double() / 65536.0, sts[1].double() / 65536.0
        assert float((a - b).abs().max()) <= 1e-3 * max(1.0, float(a.abs().max()))


@pytest.mark.parametrize("cin,cout,k,stride,HW,B", [(16, 16, 3, 1, (64, 80), 5), (16, 32, 3, 1, (40, 52), 3), (8, 16, 5, 2, (50, 72), 4), (32, 16, 7, 1, (37, 36), 9),
                                                     (16, 16, 3, 1, (33, 36), 17)])
def test_conv2d_xcd_grouped_tiles(ops, cin, cout, k, stride, HW, B):
    """DMVS_TUNE_XCD_GROUP(n): which tiles meet in one XCD's L2 is a bijection of the tile indices -- every group size, with tile counts
    that are and are not multiples of 8 x the group, gives the round-robin order's bits"""
    x = rnd(B, cin, *HW, seed=1)
    w, bias = rnd(cout, cin, k, k, seed=2) * 0.2, rnd(cout, seed=3)
    ref = F.relu(F.conv2d(x, w, bias, stride, k // 2))
    pc = K.pack_conv2d(*dev(ops, w, bias), stride=stride, pad=k // 2)
    outs = [ops.conv2d(pc, dev(ops, x), act=K.ACT_RELU, tune=K._lib.tune_xcd_group(n)).cpu() for n in range(1, 8)]
    close(outs[0], ref, 2e-5)
    for o in outs[1:]:
        assert torch.equal(outs[0], o)


@pytest.mark.parametrize("cin,cout,k,stride,HW", [(8, 16, 5, 2, (50, 72)), (16, 32, 5, 2, (44, 40)), (32, 16, 7, 1, (37, 36)), (8, 16, 3, 2, (40, 52)),
                                                   (32, 64, 5, 2, (36, 40)), (16, 16, 3, 1, (33, 36))])
def test_conv2d_forced_tile_heights(ops, cin, cout, k, stride, HW):
    """DMVS_TUNE_TILE_MT(1 | 2 | 4): every tile height a family has gives the same bits (the stride-2 / 5x5 / 7x7 families have 16 x 4 and
    16 x 8 tiles only: their 16 x 16 form was timed in round 5 and removed -- a forced 4 falls back to the dispatcher's choice)"""
    B = 2
    x = rnd(B, cin, *HW, seed=1)
    w, bias = rnd(cout, cin, k, k, seed=2) * 0.2, rnd(cout, seed=3)
    ref = F.relu(F.conv2d(x, w, bias, stride, k // 2))
    pc = K.pack_conv2d(*dev(ops, w, bias), stride=stride, pad=k // 2)
    outs = [ops.conv2d(pc, dev(ops, x), act=K.ACT_RELU, tune=K._lib.tune_tile_mt(mt) | K._lib.TUNE_NO_TALL).cpu() for mt in (1, 2, 4)]
    close(outs[0], ref, 2e-5)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("bad", [float("nan"), float("inf"), 3.0e7])
def test_groupnorm_stats_propagate_non_finite_and_out_of_range(ops, bad):
    """the fixed-point statistics slots (dmvs_common.h): a NaN / Inf activation -- or one beyond the documented magnitude
    contract -- makes ITS group's outputs NaN (as fp statistics would), the other batch item stays exact"""
    B, C, H, W = 2, 16, 16, 16
    x = rnd(B, C, H, W, seed=1)
    x[1, 5, 3, 4] = bad                                   # batch item 1: reaches channels of every group through the 3x3 conv
    w = rnd(C, C, 3, 3, seed=2) * 0.3
    gamma, beta = rnd(C, seed=4, lo=0.5, hi=1.5), rnd(C, seed=5)
    stats = torch.zeros(B * 8, dtype=torch.float64, device=ops.device)
    h = ops.conv2d(K.pack_conv2d(dev(ops, w), pad=1), dev(ops, x), gn_stats=stats)
    out = ops.groupnorm_apply(h, *dev(ops, gamma, beta), 4, stats).cpu()
    want0 = F.silu(F.group_norm(F.conv2d(x[:1], w, None, 1, 1), 4, gamma, beta, 1e-5))
    close(out[:1], want0, 3e-5)
    assert torch.isnan(out[1]).all()


# ------------------------------------------------------------------ backward kernels (training step)
def _oracle_init_cor(feats, pm, dv, D):
    B, _, H, W = feats[0].shape
    hyp = (torch.arange(D).view(1, -1, 1, 1) / (D - 1.0)).repeat(B, 1, H, W)
    hyp = O.disp_to_depth(hyp, (1 / dv[:, 1]).view(-1, 1, 1, 1), (1 / dv[:, 0]).view(-1, 1, 1, 1))[1]
    ref_proj = O.compose_proj(pm[:, 0])
    return torch.stack([O.group_corr(O.warp(feats[v], O.compose_proj(pm[:, v]), ref_proj, hyp), feats[0], 4)
                        for v in range(1, len(feats))], 1)


@pytest.mark.parametrize("C", [48, 32, 16])
def test_warp_corr_init_backward(ops, C, monkeypatch):
    """grad w.r.t. reference and source features against autograd through the oracle
    (grid_sample backward, reference module.py:212-218; the grid itself is built under no_grad :187); the per-pixel kernel in both
    channel -> lane mappings (consecutive channels per lane / interleaved: include/dmvs.h DMVS_BWD_GATHER_INTERLEAVED)"""
    B, S, D, H, W = 2, 2, 5, 9, 12
    pm = _cams(B, S + 1, H, W, 4)
    feats = [rnd(B, C, H, W, seed=40 + v).requires_grad_(True) for v in range(S + 1)]
    dv = torch.tensor([[1 / 935.0, 1 / 425.0], [1 / 800.0, 1 / 500.0]])
    cor = _oracle_init_cor(feats, pm, dv, D)
    gcor = rnd(*cor.shape, seed=50)
    cor.backward(gcor)
    rt = ops.compose_proj(dev(ops, pm))
    ref_nhwc = dev(ops, feats[0].detach().permute(0, 2, 3, 1))
    src_nhwc = dev(ops, torch.stack([f.detach().permute(0, 2, 3, 1) for f in feats[1:]]))
    for gather, il in ((False, False), (True, False), (True, True)):
        monkeypatch.setitem(ops.tune, "bwd_il", il)
        gref, gsrc = ops.warp_corr_init_bwd(ref_nhwc, src_nhwc, rt, dev(ops, 1 / (1 / dv[:, 0])), dev(ops, 1 / (1 / dv[:, 1])),
                                            dev(ops, gcor), gather=gather)
        close(gref.permute(0, 3, 1, 2), feats[0].grad, 1e-4)
        for v in range(S):
            close(gsrc[v].permute(0, 3, 1, 2), feats[v + 1].grad, 1e-4)


@pytest.mark.parametrize("H,W,D,scene", [(36, 44, 12, True), (20, 36, 7, False)])
def test_warp_corr_init_backward_window_tiles(ops, H, W, D, scene):
    """plane-sweep backward through LDS windows (C = 48): several tiles, depth chunks; synthetic cameras (windows fit) and
    strongly rotated cameras (chunks scatter in global memory)"""
    B, S, C = 2, 2, 48
    feats = [rnd(B, C, H, W, seed=90 + v).requires_grad_(True) for v in range(S + 1)]
    if scene:
        _, proj, dvs = synth.synth_inputs(H * 8, W * 8, S, B=B, seed=7)
        pm = proj["stage1"]
        dv = torch.stack([dvs[:, 0], dvs[:, -1]], 1)
    else:
        pm = _cams(B, S + 1, H, W, 9)
        dv = torch.tensor([[1 / 935.0, 1 / 425.0], [1 / 800.0, 1 / 500.0]])
    cor = _oracle_init_cor(feats, pm, dv, D)
    gcor = rnd(*cor.shape, seed=95)
    cor.backward(gcor)
    rt = ops.compose_proj(dev(ops, pm))
    ref_nhwc = dev(ops, feats[0].detach().permute(0, 2, 3, 1))
    src_nhwc = dev(ops, torch.stack([f.detach().permute(0, 2, 3, 1) for f in feats[1:]]))
    gref, gsrc = ops.warp_corr_init_bwd(ref_nhwc, src_nhwc, rt, dev(ops, dv[:, 0].contiguous()), dev(ops, dv[:, 1].contiguous()),
                                        dev(ops, gcor))
    close(gref.permute(0, 3, 1, 2), feats[0].grad, 1e-4)
    for v in range(S):
        close(gsrc[v].permute(0, 3, 1, 2), feats[v + 1].grad, 1e-4)


@pytest.mark.parametrize("C,n,with_conf,H,W,interval", [(32, 6, True, 10, 12, 2.0 / 384), (16, 4, False, 10, 12, 2.0 / 384),
                                                        (32, 6, True, 36, 44, 2.0 / 384), (16, 4, True, 40, 24, 1.0 / 384),
                                                        (32, 4, False, 20, 36, 0.2)])
def test_getcost_backward(ops, C, n, with_conf, H, W, interval, monkeypatch):
    """grad_ref / grad_src of GetCost against autograd through the oracle: LDS-window kernel (several tiles, partial
    tiles; the large interval sends the tiles to the per-pixel kernel through the worklist) and the per-pixel kernel, each with both
    channel -> lane mappings of the scatter (DMVS_TUNE_BWD_INTERLEAVED)"""
    B, S = 2, 3
    pm = _cams(B, S + 1, H, W, 5)
    feats = [rnd(B, C, H, W, seed=60 + v).requires_grad_(True) for v in range(S + 1)]
    inv = rnd(B, 1, H, W, seed=70, lo=-0.1, hi=1.1)
    conf = rnd(B, H, W, seed=71, lo=0.0, hi=1.0) if with_conf else None
    vw = rnd(B, S, H // 2, W // 2, seed=72, lo=0.0, hi=1.0)
    dv0, dv1 = torch.tensor([1 / 935.0, 1 / 700.0]), torch.tensor([1 / 425.0, 1 / 450.0])
    dmax, dmin = (1 / dv0).view(-1, 1, 1, 1), (1 / dv1).view(-1, 1, 1, 1)
    cost, _ = O.get_cost(feats, pm, inv, interval, dmax, dmin, n, F.interpolate(vw, scale_factor=2, mode="nearest"),
                         conf, 4, 0.25, 4.0)
    gcost = rnd(*cost.shape, seed=73)
    cost.backward(gcost)
    rt = ops.compose_proj(dev(ops, pm))
    for gather, il in ((False, False), (True, False), (False, True), (True, True)):
        monkeypatch.setitem(ops.tune, "bwd_il", il)
        gref, gsrc = ops.getcost_bwd(dev(ops, feats[0].detach().permute(0, 2, 3, 1)),
                                     dev(ops, torch.stack([f.detach().permute(0, 2, 3, 1) for f in feats[1:]])), rt,
                                     dev(ops, inv), dev(ops, conf), dev(ops, vw), dev(ops, 1 / (1 / dv0)), dev(ops, 1 / (1 / dv1)),
                                     n, interval, 0.25, 4.0, 1, dev(ops, gcost), gather=gather)
        close(gref.permute(0, 3, 1, 2), feats[0].grad, 1e-4)
        for v in range(S):
            close(gsrc[v].permute(0, 3, 1, 2), feats[v + 1].grad, 1e-4)


def test_view_aggregate_backward(ops):
    B, S, G, D, H, W = 2, 3, 4, 5, 4, 6
    cor = rnd(B, S, G, D, H, W, seed=1).requires_grad_(True)
    w = rnd(B, S, H, W, seed=2, lo=0.05, hi=1).requires_grad_(True)
    out = (cor * w.view(B, S, 1, 1, H, W)).sum(1) / (1e-8 + w.sum(1)).view(B, 1, 1, H, W)
    gout = rnd(*out.shape, seed=3)
    out.backward(gout)
    gcor, gw = ops.view_aggregate_bwd(*dev(ops, cor.detach(), w.detach(), out.detach(), gout))
    close(gcor, cor.grad, 1e-5)
    close(gw, w.grad, 1e-5)


@pytest.mark.parametrize("cin,cout,k,stride,pad,in_mode", [
    (8, 16, 3, 1, 1, "plain"), (6, 10, 5, 2, 2, "plain"), (16, 8, 3, 2, 1, "plain"), (12, 20, 1, 1, 0, "plain"),
    (5, 16, 7, 1, 3, "plain"), (8, 8, 3, 1, 1, "upsample"), (4, 12, 1, 1, 0, "unshuffle"), (9, 7, (1, 5), 1, (0, 2), "plain"),
])
def test_conv2d_autograd(ops, cin, cout, k, stride, pad, in_mode):
    """forward / input gradient / weight gradient / bias gradient of the training conv against F.conv2d's autograd"""
    from diffmvs_amd import autograd as A
    B, H, W = 2, 12, 20
    ks = (k, k) if isinstance(k, int) else k
    x = rnd(B, cin, H, W, seed=1).requires_grad_(True)
    w = (rnd(cout, cin * (4 if in_mode == "unshuffle" else 1), *ks, seed=2) * 0.3).requires_grad_(True)
    b = rnd(cout, seed=3).requires_grad_(True)
    if in_mode == "upsample":
        xin = F.interpolate(x, scale_factor=2, mode="nearest")
    elif in_mode == "unshuffle":
        xin = O._pixel_unshuffle(x)
    else:
        xin = x
    ref = F.conv2d(xin, w, b, stride, pad)
    g = rnd(*ref.shape, seed=4)
    ref.backward(g)
    xd, wd, bd = [t.detach().to(ops.device).requires_grad_(True) for t in (x, w, b)]
    mode = {"plain": K.IN_PLAIN, "upsample": K.IN_UPSAMPLE2, "unshuffle": K.IN_UNSHUFFLE2}[in_mode]
    out = A.conv2d(ops, xd, wd, bd, stride=stride, pad=pad, in_mode=mode)
    close(out, ref.detach(), 3e-5)
    out.backward(dev(ops, g))
    close(xd.grad, x.grad, 1e-4)
    close(wd.grad, w.grad, 1e-4)
    close(bd.grad, b.grad, 1e-4)


def test_conv2d_wgrad_refuses_channel_slices(ops):
    """the weight-gradient entry points read dense inputs only: a descriptor carrying the forward's channel-slice / producer-product
    fields (in0_cstride, gate_cstride, out_mul) is refused with DMVS_EINVAL instead of being read from the wrong memory"""
    import ctypes as C
    from diffmvs_amd import _lib
    x, g = dev(ops, rnd(1, 8, 6, 10, seed=1), rnd(1, 8, 6, 10, seed=2))
    base = dict(in0=x.data_ptr(), B=1, c0=8, c1=0, Hin=6, Win=10, Hout=6, Wout=10, cout=8, cout_pad=8, kh=3, kw=3, stride=1, pad_h=1, pad_w=1,
                in_mode=K.IN_PLAIN, out_cstride=8, post_scale=1.0)
    nbytes = C.c_int64(0)
    ops.lib.call("dmvs_conv2d_wgrad_workspace_f32", C.byref(_lib.Conv2dDesc(**base)), C.byref(nbytes))      # the dense descriptor is fine
    for extra in ({"in0_cstride": 16}, {"gate_cstride": 16}, {"out_mul": g.data_ptr()}):
        with pytest.raises(_lib.DmvsError, match="-22"):
            ops.lib.call("dmvs_conv2d_wgrad_workspace_f32", C.byref(_lib.Conv2dDesc(**base, **extra)), C.byref(nbytes))
        with pytest.raises(_lib.DmvsError, match="-22"):
            ops.lib.call("dmvs_conv2d_wgrad_f32", C.byref(_lib.Conv2dDesc(**base, **extra)), C.c_void_p(g.data_ptr()), C.c_void_p(g.data_ptr()), None,
                         C.c_void_p(g.data_ptr()), 1 << 30, ops.stream())


@pytest.mark.parametrize("cin,cout,stride,transposed", [(4, 8, 1, False), (8, 16, 2, False), (16, 8, 2, True), (8, 1, 1, False),
                                                         (20, 12, 1, False)])
def test_conv3d_autograd(ops, cin, cout, stride, transposed):
    from diffmvs_amd import autograd as A
    B, D, H, W = 2, 6, 8, 10
    if transposed:
        D, H, W = 3, 4, 5
    x = rnd(B, cin, D, H, W, seed=1).requires_grad_(True)
    b = rnd(cout, seed=3).requires_grad_(True)
    if transposed:
        w = (rnd(cin, cout, 3, 3, 3, seed=2) * 0.2).requires_grad_(True)
        ref = F.conv_transpose3d(x, w, b, 2, 1, 1)
    else:
        w = (rnd(cout, cin, 3, 3, 3, seed=2) * 0.2).requires_grad_(True)
        ref = F.conv3d(x, w, b, stride, 1)
    g = rnd(*ref.shape, seed=4)
    ref.backward(g)
    xd, wd, bd = [t.detach().to(ops.device).requires_grad_(True) for t in (x, w, b)]
    out = A.conv3d(ops, xd, wd, bd, stride=stride, transposed=transposed)
    close(out, ref.detach(), 3e-5)
    out.backward(dev(ops, g))
    close(xd.grad, x.grad, 1e-4)
    close(wd.grad, w.grad, 1e-4)
    close(bd.grad, b.grad, 1e-4)


@pytest.mark.parametrize("shape,relu", [((3, 8, 12, 20), True), ((2, 5, 7, 9), False), ((2, 16, 4, 6, 10), True),
                                        ((1, 3, 200, 170), True)])
def test_batchnorm_train_autograd(ops, shape, relu):
    """training-mode BatchNorm(+ReLU): output, running stats, dx / dgamma / dbeta against ATen's batch_norm autograd"""
    from diffmvs_amd import autograd as A
    C_ = shape[1]
    x = (rnd(*shape, seed=1) * 2 + 0.5).requires_grad_(True)
    gamma = (rnd(C_, seed=2) * 0.5 + 1).requires_grad_(True)
    beta = (rnd(C_, seed=3) * 0.3).requires_grad_(True)
    rm, rv = rnd(C_, seed=4) * 0.1, rnd(C_, seed=5).abs() + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    ref = F.batch_norm(x, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5)
    if relu:
        ref = F.relu(ref)
    g = rnd(*shape, seed=6)
    ref.backward(g)
    xd, gd, bd = [t.detach().to(ops.device).requires_grad_(True) for t in (x, gamma, beta)]
    rmd, rvd = dev(ops, rm.clone(), rv.clone())
    out = A.batchnorm_act(ops, xd, gd, bd, rmd, rvd, 0.1, 1e-5, relu)
    close(out, ref.detach(), 2e-5)
    close(rmd, rm_ref, 1e-5)
    close(rvd, rv_ref, 1e-5)
    out.backward(dev(ops, g))
    close(xd.grad, x.grad, 1e-4)
    close(gd.grad, gamma.grad, 1e-4)
    close(bd.grad, beta.grad, 1e-4)


@pytest.mark.parametrize("view_major", [True, False])
def test_batchnorm_train_views(ops, view_major):
    """V batched BatchNorm calls (per-view statistics, V sequential running-stat updates) == V separate ATen calls"""
    from diffmvs_amd import autograd as A
    V, Bv, C_, H, W = 3, 2, 6, 9, 12
    xs = [(rnd(Bv, C_, H, W, seed=10 + v) * (1 + v) + v).requires_grad_(True) for v in range(V)]
    gamma = (rnd(C_, seed=2) * 0.5 + 1).requires_grad_(True)
    beta = (rnd(C_, seed=3) * 0.3).requires_grad_(True)
    rm_ref, rv_ref = rnd(C_, seed=4) * 0.1, rnd(C_, seed=5).abs() + 0.5
    rm, rv = rm_ref.clone(), rv_ref.clone()
    gs = [rnd(Bv, C_, H, W, seed=20 + v) for v in range(V)]
    refs = []
    for v in range(V):
        y = F.relu(F.batch_norm(xs[v], rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5))
        y.backward(gs[v])
        refs.append(y.detach())
    if view_major:
        xcat, gcat = torch.cat([x.detach() for x in xs], 0), torch.cat(gs, 0)
    else:
        xcat, gcat = torch.stack([x.detach() for x in xs], 1).reshape(V * Bv, C_, H, W), torch.stack(gs, 1).reshape(V * Bv, C_, H, W)
    xd = xcat.to(ops.device).requires_grad_(True)
    gd, bd = [t.detach().to(ops.device).requires_grad_(True) for t in (gamma, beta)]
    rmd, rvd = dev(ops, rm, rv)
    out = A.batchnorm_act(ops, xd, gd, bd, rmd, rvd, 0.1, 1e-5, True, V, view_major)
    out.backward(dev(ops, gcat))
    o_ = out.detach().cpu()
    gx = xd.grad.cpu()
    for v in range(V):
        sel = (lambda t: t[v * Bv:(v + 1) * Bv]) if view_major else (lambda t: t.view(Bv, V, C_, H, W)[:, v])
        close(sel(o_), refs[v], 2e-5)
        close(sel(gx), xs[v].grad, 1e-4)
    close(rmd, rm_ref, 1e-5)
    close(rvd, rv_ref, 1e-5)
    close(gd.grad, gamma.grad, 1e-4)
    close(bd.grad, beta.grad, 1e-4)


@pytest.mark.parametrize("N,H,W", [(2, 32, 48), (3, 37, 53), (1, 16, 16)])
def test_featurenet_stem_fused(ops, N, H, W):
    """conv0.0 + BN + ReLU + conv0.1 + BN + ReLU in one kernel == the two conv2d launches == ATen (partial tiles, image
    borders: the intermediate must be ZERO outside the image, not conv0.0 of padding)"""
    x = rnd(N, 3, H, W, seed=1)
    w0, w1 = rnd(8, 3, 3, 3, seed=2) * 0.4, rnd(8, 8, 3, 3, seed=3) * 0.3

    def bn(seed):
        return {"weight": rnd(8, seed=seed, lo=0.5, hi=1.5), "bias": rnd(8, seed=seed + 1) * 0.3,
                "running_mean": rnd(8, seed=seed + 2) * 0.2, "running_var": rnd(8, seed=seed + 3, lo=0.5, hi=1.5)}
    b0, b1 = bn(10), bn(20)
    ref = F.relu(F.batch_norm(F.conv2d(x, w0, None, 1, 1), b0["running_mean"], b0["running_var"], b0["weight"], b0["bias"], False, 0.0, 1e-5))
    ref = F.relu(F.batch_norm(F.conv2d(ref, w1, None, 1, 1), b1["running_mean"], b1["running_var"], b1["weight"], b1["bias"], False, 0.0, 1e-5))
    d = lambda b: {k_: v.to(ops.device) for k_, v in b.items()}
    pc0 = K.pack_conv2d(dev(ops, w0), bn=d(b0), pad=1)
    pc1 = K.pack_conv2d(dev(ops, w1), bn=d(b1), pad=1)
    out = ops.featurenet_stem(pc0, pc1, dev(ops, x))
    close(out, ref, 2e-5)
    two = ops.conv2d(pc1, ops.conv2d(pc0, dev(ops, x), act=K.ACT_RELU), act=K.ACT_RELU)
    close(out, two.cpu(), 1e-6)          # conv0.0 sums its 27 products in a different grouping: last-bit differences only


@pytest.mark.parametrize("N,H,W", [(2, 37, 52), (1, 64, 96), (2, 9, 12)])
def test_featurenet_stem_16_byte_pieces(ops, N, H, W):
    """the stem's input halo staged in 16-byte pieces (the default wherever rows are 16-byte multiples: 1094 -> 938 us per 96 images on
    the MI355X), BIT FOR BIT the 4-byte form (tune = DMVS_TUNE_PIECES4); several tiles per workgroup, all four borders"""
    x = dev(ops, rnd(N, 3, H, W, seed=1))
    w0, w1 = rnd(8, 3, 3, 3, seed=2) * 0.4, rnd(8, 8, 3, 3, seed=3) * 0.3
    pc0, pc1 = K.pack_conv2d(*dev(ops, w0, rnd(8, seed=4)), pad=1), K.pack_conv2d(*dev(ops, w1, rnd(8, seed=5)), pad=1)
    a = ops.featurenet_stem(pc0, pc1, x, tune=K._lib.TUNE_PIECES4)
    b = ops.featurenet_stem(pc0, pc1, x)
    x = x.cpu()
    ref = F.relu(F.conv2d(F.relu(F.conv2d(x, w0, rnd(8, seed=4), 1, 1)), w1, rnd(8, seed=5), 1, 1))
    close(b, ref, 2e-5)
    assert torch.equal(a.cpu(), b.cpu())


@pytest.mark.parametrize("N,H,W", [(2, 37, 52), (1, 64, 96), (2, 9, 12)])
def test_featurenet_stem_split_bf16_arithmetic(ops, N, H, W):
    """conv0.1 of the fused stem in split-bf16 arithmetic (the default under conv_arith = "split"; conv0.0's epilogue writes the intermediate as
    bf16 triples): against torch in fp64 its error is of the size of the exact-fp32 stem's own, both staging forms give the same bits"""
    x = rnd(N, 3, H, W, seed=1)
    w0, w1 = rnd(8, 3, 3, 3, seed=2) * 0.4, rnd(8, 8, 3, 3, seed=3) * 0.3
    b0, b1 = rnd(8, seed=4), rnd(8, seed=5)
    pc0, pc1 = K.pack_conv2d(*dev(ops, w0, b0), pad=1), K.pack_conv2d(*dev(ops, w1, b1), pad=1)
    so = ops.with_conv_arith(K.ARITH_SPLIT)
    a = so.featurenet_stem(pc0, pc1, dev(ops, x)).cpu()
    b = so.featurenet_stem(pc0, pc1, dev(ops, x), tune=K._lib.TUNE_PIECES4).cpu()
    e = ops.featurenet_stem(pc0, pc1, dev(ops, x)).cpu()
    ref = F.relu(F.conv2d(F.relu(F.conv2d(x.double(), w0.double(), b0.double(), 1, 1)), w1.double(), b1.double(), 1, 1))
    scale = float(ref.abs().max())
    e_split, e_f32 = float((a.double() - ref).abs().max()) / scale, float((e.double() - ref).abs().max()) / scale
    print("stem split vs fp64: %.2e   exact fp32 vs fp64: %.2e" % (e_split, e_f32))
    assert e_f32 < 1e-6 and e_split < 2e-6 and e_split < 4.0 * e_f32 + 2e-7, (e_split, e_f32)
    assert torch.equal(a, b)
    assert not torch.equal(a, e)      # (the mode is really on)


@pytest.mark.parametrize("N,H,W", [(2, 37, 52), (3, 64, 96)])
def test_featurenet_stem_xcd_grouped_tiles(ops, N, H, W):
    """DMVS_TUNE_XCD_GROUP on the stem's tile walk: every group size gives the round-robin order's bits"""
    x = dev(ops, rnd(N, 3, H, W, seed=1))
    w0, w1 = rnd(8, 3, 3, 3, seed=2) * 0.4, rnd(8, 8, 3, 3, seed=3) * 0.3
    pc0, pc1 = K.pack_conv2d(*dev(ops, w0, rnd(8, seed=4)), pad=1), K.pack_conv2d(*dev(ops, w1, rnd(8, seed=5)), pad=1)
    outs = [ops.featurenet_stem(pc0, pc1, x, tune=K._lib.tune_xcd_group(n)).cpu() for n in (1, 0, 2, 4)]
    for o in outs[1:]:
        assert torch.equal(outs[0], o)


@pytest.mark.parametrize("cin,cout,stride,transposed,B,D,H,W", [(4, 8, 1, False, 2, 23, 62, 100), (3, 16, 1, False, 2, 23, 62, 100), (8, 8, 1, False, 3, 9, 14, 36),
                                                                 (8, 16, 2, False, 2, 10, 20, 36), (16, 8, 2, True, 2, 5, 10, 18)])
def test_conv3d_xcd_grouped_tiles(ops, cin, cout, stride, transposed, B, D, H, W):
    """DMVS_TUNE3D_XCD_GROUP: which tiles meet in one XCD's L2 is a bijection of the tile indices -- streamed, paired, generic, stride-2 and
    transposed kernels give the round-robin order's bits with every group size"""
    x = dev(ops, rnd(B, cin, D, H, W, seed=1))
    w = rnd(cin, cout, 3, 3, 3, seed=2) * 0.2 if transposed else rnd(cout, cin, 3, 3, 3, seed=2) * 0.2
    pc = K.pack_conv3d(*dev(ops, w, rnd(cout, seed=3)), stride=stride, transposed=transposed)
    outs = [ops.conv3d(pc, x, act=K.ACT_RELU, tune=K._lib.tune3d_xcd_group(n)).cpu() for n in (1, 0, 2, 4)]
    for o in outs[1:]:
        assert torch.equal(outs[0], o)


@pytest.mark.parametrize("cin,cout,with_res", [(4, 8, False), (3, 8, True)])
def test_conv3d_streamed_tiles(ops, cin, cout, with_res):
    """cin <= 4 on a volume with more tiles than resident workgroups: the persistent, tile-pipelined instantiation
    (PixelViewWeight conv0 / CostReg conv0 shapes), ragged sizes so that border tiles and partial tiles are hit"""
    B, D, H, W = 2, 23, 62, 100
    x = rnd(B, cin, D, H, W, seed=1)
    w = rnd(cout, cin, 3, 3, 3, seed=2) * 0.2
    bn = {"weight": rnd(cout, seed=4, lo=0.5, hi=1.5), "bias": rnd(cout, seed=5),
          "running_mean": rnd(cout, seed=6), "running_var": rnd(cout, seed=7, lo=0.5, hi=1.5)}
    ref = F.relu(F.batch_norm(F.conv3d(x, w, None, 1, 1), bn["running_mean"], bn["running_var"], bn["weight"], bn["bias"], False, 0.0, 1e-5))
    res = rnd(*ref.shape, seed=9) if with_res else None
    if with_res:
        ref = ref + res
    pc = K.pack_conv3d(dev(ops, w), bn={k_: v.to(ops.device) for k_, v in bn.items()})
    out = ops.conv3d(pc, dev(ops, x), act=K.ACT_RELU, residual=dev(ops, res))
    close(out, ref, 2e-5)


@pytest.mark.parametrize("cin,cout,B,D,H,W", [(4, 8, 2, 23, 62, 100),      # pair kernel (4 -> 8, two depth slices per MFMA), >= 512 tiles
                                              (3, 16, 2, 23, 62, 100),     # streamed kernel (cin <= 4, one n-tile)
                                              (8, 8, 1, 9, 14, 36), (16, 16, 2, 6, 10, 20), (8, 32, 1, 5, 6, 44)])      # generic kernel, one / two n-tiles
def test_conv3d_16_byte_halo_pieces(ops, cin, cout, B, D, H, W):
    """the halo tile of the stride-1 MFMA kernels staged in 16-byte pieces (the default wherever rows are 16-byte multiples: 2-6 %
    per layer on the MI355X) -- against torch and BIT FOR BIT against the 4-byte form (tune = DMVS_TUNE3D_PIECES4); ragged volumes,
    pieces outside the volume on every face"""
    x = rnd(B, cin, D, H, W, seed=1)
    w = rnd(cout, cin, 3, 3, 3, seed=2) * 0.2
    bias = rnd(cout, seed=3)
    res = rnd(B, cout, D, H, W, seed=4)
    ref = F.relu(F.conv3d(x, w, bias, 1, 1)) + res
    pc = K.pack_conv3d(*dev(ops, w, bias))
    a = ops.conv3d(pc, dev(ops, x), act=K.ACT_RELU, residual=dev(ops, res), tune=K._lib.TUNE3D_PIECES4)
    b = ops.conv3d(pc, dev(ops, x), act=K.ACT_RELU, residual=dev(ops, res))
    close(b, ref, 2e-5)
    assert torch.equal(a.cpu(), b.cpu())


@pytest.mark.parametrize("cin,cout,B,D,H,W", [(4, 8, 2, 23, 62, 100), (3, 5, 2, 17, 64, 96)])
def test_conv3d_paired_kernel_weights_in_registers(ops, cin, cout, B, D, H, W):
    """the 4 -> 8 paired kernel (two output depth slices per MFMA, each lane's 36 paired weights in registers; the default since round 5)
    against torch and BIT FOR BIT the streamed one-slice kernel (DMVS_TUNE3D_NO_PAIR); ragged volumes, fewer than 4 input / 8 output
    channels"""
    x = rnd(B, cin, D, H, W, seed=1)
    w, bias = rnd(cout, cin, 3, 3, 3, seed=2) * 0.2, rnd(cout, seed=3)
    res = rnd(B, cout, D, H, W, seed=4)
    ref = F.relu(F.conv3d(x, w, bias, 1, 1)) + res
    pc = K.pack_conv3d(*dev(ops, w, bias))
    a = ops.conv3d(pc, dev(ops, x), act=K.ACT_RELU, residual=dev(ops, res), tune=K._lib.TUNE3D_NO_PAIR)
    b = ops.conv3d(pc, dev(ops, x), act=K.ACT_RELU, residual=dev(ops, res))
    close(b, ref, 2e-5)
    assert torch.equal(a.cpu(), b.cpu())


@pytest.mark.parametrize("cin,cout,B,D,H,W", [(8, 8, 2, 23, 62, 100), (6, 5, 2, 17, 64, 96), (5, 8, 3, 9, 30, 180)])
def test_conv3d_two_chunk_paired_kernel(ops, cin, cout, B, D, H, W):
    """5..8 -> <= 8 channel layers (CostRegNet conv1) on the paired kernel over two 4-channel chunks per tile (the default since round 5:
    1517 -> 983 us per 96 volumes) -- against torch and BIT FOR BIT the generic kernel (DMVS_TUNE3D_NO_PAIR); ragged volumes and channel
    counts"""
    x = rnd(B, cin, D, H, W, seed=1)
    w, bias = rnd(cout, cin, 3, 3, 3, seed=2) * 0.2, rnd(cout, seed=3)
    res = rnd(B, cout, D, H, W, seed=4)
    ref = F.relu(F.conv3d(x, w, bias, 1, 1)) + res
    pc = K.pack_conv3d(*dev(ops, w, bias))
    a = ops.conv3d(pc, dev(ops, x), act=K.ACT_RELU, residual=dev(ops, res), tune=K._lib.TUNE3D_NO_PAIR)
    b = ops.conv3d(pc, dev(ops, x), act=K.ACT_RELU, residual=dev(ops, res))
    close(b, ref, 2e-5)
    assert torch.equal(a.cpu(), b.cpu())


@pytest.mark.parametrize("cin,cout,D,H,W,with_res", [(8, 16, 11, 18, 70, False), (16, 32, 8, 9, 33, True), (6, 12, 5, 7, 20, False)])
def test_conv3d_stride2_matrix_core_form(ops, cin, cout, D, H, W, with_res):
    """CostRegNet_small's stride-2 layers (conv2 8 -> 16, conv4 16 -> 32) on the matrix cores: several output tiles per axis, odd and
    even input sizes (the last output voxel's window ends on / beyond the volume), 2 and 4 channel chunks, one and two n-tiles"""
    B = 2
    x = rnd(B, cin, D, H, W, seed=1)
    w = rnd(cout, cin, 3, 3, 3, seed=2) * 0.2
    bn = {"weight": rnd(cout, seed=4, lo=0.5, hi=1.5), "bias": rnd(cout, seed=5),
          "running_mean": rnd(cout, seed=6), "running_var": rnd(cout, seed=7, lo=0.5, hi=1.5)}
    ref = F.relu(F.batch_norm(F.conv3d(x, w, None, 2, 1), bn["running_mean"], bn["running_var"], bn["weight"], bn["bias"], False, 0.0, 1e-5))
    res = rnd(*ref.shape, seed=9) if with_res else None
    if with_res:
        ref = ref + res
    pc = K.pack_conv3d(dev(ops, w), bn={k_: v.to(ops.device) for k_, v in bn.items()}, stride=2)
    out = ops.conv3d(pc, dev(ops, x), act=K.ACT_RELU, residual=dev(ops, res))
    close(out, ref, 2e-5)


@pytest.mark.parametrize("cin,cout,D,H,W", [(4, 8, 1, 1, 1), (8, 8, 2, 3, 17), (8, 1, 1, 2, 33), (16, 16, 5, 4, 16), (3, 1, 4, 8, 32)])
def test_conv3d_volumes_smaller_than_a_tile(ops, cin, cout, D, H, W):
    """degenerate volumes: every staged halo element is border or padding in at least one axis (packed border test),
    tiles exactly one voxel / exactly one tile wide"""
    x = rnd(2, cin, D, H, W, seed=1)
    w = rnd(cout, cin, 3, 3, 3, seed=2) * 0.3
    bias = rnd(cout, seed=3)
    ref = F.relu(F.conv3d(x, w, bias, 1, 1))
    out = ops.conv3d(K.pack_conv3d(dev(ops, w), dev(ops, bias)), dev(ops, x), act=K.ACT_RELU)
    close(out, ref, 2e-5)


@pytest.mark.parametrize("D,H,W,act,with_res", [(6, 11, 72, "sigmoid", False), (6, 11, 36, "none", False), (6, 11, 30, "none", True),
                                                 (20, 19, 22, "none", False), (33, 17, 16, "sigmoid", True),
                                                 (14, 20, 80, "none", True), (5, 7, 256, "none", False), (3, 4, 260, "none", False)])
def test_conv3d_single_output_channel(ops, D, H, W, act, with_res):
    """cout = 1 (PixelViewWeight conv1 with its sigmoid, CostRegNet's prob head), every lane on 2 x 2 x 4 outputs: full-row
    tiles (16-byte LDS-DMA pieces, host-chosen tile shape) where rows are 16-byte multiples of 16..256 floats, 16 x 16 x 16
    tiles otherwise; several tiles per axis with ragged last tiles (odd extents: a lane's second row / slice falls
    outside), idle lane rows (256 not a multiple of W/4), 8 double-buffered input channels"""
    B, cin = 2, 8
    x = rnd(B, cin, D, H, W, seed=1)
    w = rnd(1, cin, 3, 3, 3, seed=2) * 0.2
    bias = rnd(1, seed=3)
    ref = F.conv3d(x, w, bias, 1, 1)
    if act == "sigmoid":
        ref = torch.sigmoid(ref)
    res = rnd(B, 1, D, H, W, seed=4) if with_res else None
    if with_res:
        ref = ref + res
    pc = K.pack_conv3d(dev(ops, w), dev(ops, bias))
    out = ops.conv3d(pc, dev(ops, x), act=K.ACT_SIGMOID if act == "sigmoid" else K.ACT_NONE, residual=dev(ops, res) if with_res else None)
    close(out, ref, 2e-5)
    # the same volume at a 4-byte offset: the 16-byte forms do not apply, the values must not change
    store = torch.zeros(x.numel() + 1, device=ops.device)
    store[1:] = dev(ops, x).reshape(-1)
    out2 = ops.conv3d(pc, store[1:].view(x.shape), act=K.ACT_SIGMOID if act == "sigmoid" else K.ACT_NONE, residual=dev(ops, res) if with_res else None)
    assert torch.equal(out2, out)


@pytest.mark.parametrize("B,C_,H,W,with_ss", [(2, 8, 9, 13, True), (3, 16, 20, 24, False), (1, 32, 100, 90, True)])
def test_groupnorm_silu_autograd(ops, B, C_, H, W, with_ss):
    """Block.forward of the diffusion Unet in training: silu(GroupNorm(x)*(scale+1)+shift) forward and all five gradients"""
    from diffmvs_amd import autograd as A
    x = (rnd(B, C_, H, W, seed=1) * 2 + 0.3).requires_grad_(True)
    gamma = (rnd(C_, seed=2) * 0.5 + 1).requires_grad_(True)
    beta = (rnd(C_, seed=3) * 0.3).requires_grad_(True)
    ss = (rnd(B, 2 * C_, seed=4) * 0.5).requires_grad_(True) if with_ss else None
    ref = F.group_norm(x, 4, gamma, beta, 1e-5)
    if with_ss:
        sc, sh = ss[:, :, None, None].chunk(2, dim=1)
        ref = ref * (sc + 1) + sh
    ref = F.silu(ref)
    g = rnd(B, C_, H, W, seed=5)
    ref.backward(g)
    xd, gd, bd = [t.detach().to(ops.device).requires_grad_(True) for t in (x, gamma, beta)]
    sd = ss.detach().to(ops.device).requires_grad_(True) if with_ss else None
    out = A.groupnorm_silu(ops, xd, gd, bd, 4, sd, 1e-5)
    close(out, ref.detach(), 2e-5)
    out.backward(dev(ops, g))
    close(xd.grad, x.grad, 1e-4)
    close(gd.grad, gamma.grad, 1e-4)
    close(bd.grad, beta.grad, 1e-4)
    if with_ss:
        close(sd.grad, ss.grad, 1e-4)


# ------------------------------------------------------------------------------------------ quad-per-pixel warp kernels
def _g4(x_nhwc):
    from diffmvs_amd.ops import g4_channels
    return x_nhwc[..., g4_channels(x_nhwc.shape[-1])].contiguous()


def test_g4_channel_order():
    from diffmvs_amd.ops import g4_channels
    for C in (16, 32, 48):
        p = g4_channels(C)
        assert sorted(p.tolist()) == list(range(C))
        # lane q of a quad reads positions 16j + 4q .. 16j + 4q + 3: all channels of group q, in order
        for q in range(4):
            got = torch.cat([p[16 * j + 4 * q: 16 * j + 4 * q + 4] for j in range(C // 16)])
            assert got.tolist() == list(range(q * C // 4, (q + 1) * C // 4))


@pytest.mark.parametrize("C,n,interval,H,W,with_conf", [(32, 6, 2.0 / 384, 20, 28, True), (16, 4, 1.0 / 384, 18, 25, True),
                                                        (32, 6, 0.15, 12, 20, True), (16, 4, 0.3, 10, 18, False),
                                                        (48, 4, 2.0 / 384, 9, 13, False), (32, 4, 0.0, 8, 8, False)])
def test_getcost_quad(ops, C, n, interval, H, W, with_conf):
    """quad-per-pixel GetCost (any geometry in one launch): small intervals = the 8x8 texel grid path, the large ones spread
    a pixel's hypotheses over more than 8 texels (per-hypothesis chunks); hypotheses clamped at both ends; interval 0 =
    all hypotheses identical.  Against the oracle, and the plain-NHWC feature order bit for bit against the NHWC-g4 order."""
    B, S = 2, 3
    pm = _cams(B, S + 1, H, W, 2)
    feats = [rnd(B, C, H, W, seed=40 + v) for v in range(S + 1)]
    inv = rnd(B, 1, H, W, seed=50, lo=-0.05, hi=1.05)
    conf = rnd(B, H, W, seed=51, lo=0.0, hi=1.0) if with_conf else None
    vw = rnd(B, S, H // 2, W // 2, seed=52, lo=0.0, hi=1.0)
    dv0, dv1 = torch.tensor([1 / 935.0, 1 / 700.0]), torch.tensor([1 / 425.0, 1 / 450.0])
    dmax, dmin = (1 / dv0).view(-1, 1, 1, 1), (1 / dv1).view(-1, 1, 1, 1)
    # The half-resolution view weights cover the even part of the plane only (the model's planes are multiples of 32); for the odd
    # sizes the oracle runs on the FULL plane -- the warp's zero padding depends on the real source size -- with the weights of the
    # uncovered last row / column set to anything, and the costs are compared where the weights are defined.  The hypotheses do not
    # depend on the weights: compared everywhere.
    Hv, Wv = (H // 2) * 2, (W // 2) * 2
    vw_full = torch.zeros(B, S, H, W)
    vw_full[:, :, :Hv, :Wv] = F.interpolate(vw, scale_factor=2, mode="nearest")
    want_cost, want_s = O.get_cost(feats, pm, inv, interval, dmax, dmin, n, vw_full, conf, 4, 0.25, 4.0)
    rt = ops.compose_proj(dev(ops, pm))
    ref = feats[0].permute(0, 2, 3, 1)
    src = torch.stack([f.permute(0, 2, 3, 1) for f in feats[1:]])
    tail = (rt, dev(ops, inv), None if conf is None else dev(ops, conf), dev(ops, vw), dev(ops, 1 / (1 / dv0)), dev(ops, 1 / (1 / dv1)),
            n, interval, 0.25, 4.0)
    cost, samp = ops.getcost_quad(dev(ops, _g4(ref)), dev(ops, _g4(src)), *tail, vw_shift=1)
    cost_g, samp_g = ops.getcost_quad(dev(ops, ref.contiguous()), dev(ops, src.contiguous()), *tail, vw_shift=1, plain=True)
    close(samp, want_s, 1e-6)
    close(cost[:, :, :Hv, :Wv], want_cost[:, :, :Hv, :Wv], 1e-4)
    assert torch.equal(samp.cpu(), samp_g.cpu()) and torch.equal(cost.cpu()[:, :, :Hv, :Wv], cost_g.cpu()[:, :, :Hv, :Wv])


def test_getcost_quad_extreme_geometry(ops, golden):
    """the reference's own warping edge cases (mild, large rotation, camera looking backwards, source grid != hypothesis
    grid) through the quad-per-pixel GetCost kernel; same construction as test_getcost_extreme_geometry"""
    g = golden("warp_edge.npz")
    for ci in (0, 1, 2, 3):
        src, depth, want = g.t(f"c{ci}.src"), g.t(f"c{ci}.depth"), g.t(f"c{ci}.out")
        B, Cc, Hs, Ws = src.shape
        D, H, W = depth.shape[1:]
        Hc, Wc = max(H, Hs), max(W, Ws)
        srcp = torch.zeros(B, 16, Hc, Wc)
        srcp[:, :Cc, :Hs, :Ws] = src
        P = torch.matmul(g.t(f"c{ci}.src_proj"), torch.inverse(g.t(f"c{ci}.ref_proj")))
        rt = torch.cat([P[:, :3, :3].reshape(B, 9), P[:, :3, 3]], 1).view(B, 1, 12)
        lo, hi = torch.full((B,), 1 / 2000.0), torch.full((B,), 1 / 100.0)
        for d in range(D):
            dpl = torch.full((B, 1, Hc, Wc), 500.0)
            dpl[:, :, :H, :W] = depth[:, d:d + 1]
            inv = ((1 / dpl) - lo.view(-1, 1, 1, 1)) / (hi - lo).view(-1, 1, 1, 1)
            w = want[:, :, d].sum(1)
            cost, samp = ops.getcost_quad(dev(ops, torch.ones(B, Hc, Wc, 16)), dev(ops, _g4(srcp.permute(0, 2, 3, 1)).unsqueeze(0)),
                                          dev(ops, rt), dev(ops, inv.contiguous()), None, dev(ops, torch.ones(B, 1, Hc, Wc)),
                                          dev(ops, lo), dev(ops, hi), 4, 0.0, 1.0, 1.0, vw_shift=0)
            cost = cost.cpu()
            got = (cost[:, 0] + cost[:, 4] + cost[:, 8] + cost[:, 12])[:, :H, :W] * 4.0
            assert float((got - w).abs().mean()) <= 2e-3 * max(1.0, float(w.abs().mean())), (ci, d)


@pytest.mark.parametrize("C,H,W,D,scene", [(48, 20, 28, 16, True), (48, 18, 15, 48, True), (48, 12, 20, 9, False), (32, 11, 14, 7, False),
                                           (16, 11, 14, 10, False), (48, 10, 150, 48, "wide"), (48, 70, 21, 44, "tall")])
def test_warp_corr_init_quad(ops, C, H, W, D, scene):
    """quad-per-pixel plane sweep: planes in chunks of 8 (ragged last chunk), synthetic cameras and strongly rotated ones
    (chunks whose planes spread over more than 8 texels take the per-plane path).  "wide" / "tall": baselines long enough
    (~55 texels of disparity range along x / along y) that the LDS band of the whole sweep exceeds its 48 KB and the plane
    groups are halved, down to single chunks.  Against the oracle and, bit for bit, the global-memory form on plain NHWC features."""
    B, S = 2, 3
    feats = [rnd(B, C, H, W, seed=80 + v) for v in range(S + 1)]
    if scene is True:
        _, proj, dvs = synth.synth_inputs(H * 8, W * 8, S, B=B, seed=7)
        pm = proj["stage1"]
        dv = torch.stack([dvs[:, 0], dvs[:, -1]], 1)
    else:
        pm = _cams(B, S + 1, H, W, 9)
        if scene == "wide":
            pm[:, :, 0, 0, 3] *= 4.0                      # t_x: -100 .. -300 mm
        elif scene == "tall":
            pm[:, :, 0, 1, 3] = pm[:, :, 0, 0, 3] * 4.0   # the long baseline along y
            pm[:, :, 0, 0, 3] *= 0.1
        dv = torch.tensor([[1 / 935.0, 1 / 425.0], [1 / 800.0, 1 / 500.0]])
    disp_min, disp_max = dv[:, 0].contiguous(), dv[:, 1].contiguous()
    hyp = (torch.arange(D).view(1, -1, 1, 1) / (D - 1.0)).repeat(B, 1, H, W)
    hyp = O.disp_to_depth(hyp, (1 / dv[:, 1]).view(-1, 1, 1, 1), (1 / dv[:, 0]).view(-1, 1, 1, 1))[1]
    ref_proj = O.compose_proj(pm[:, 0])
    want = torch.stack([O.group_corr(O.warp(feats[v], O.compose_proj(pm[:, v]), ref_proj, hyp), feats[0], 4)
                        for v in range(1, S + 1)], 1)
    rt = ops.compose_proj(dev(ops, pm))
    ref_nhwc = feats[0].permute(0, 2, 3, 1)
    src_nhwc = torch.stack([f.permute(0, 2, 3, 1) for f in feats[1:]])
    out = ops.warp_corr_init_quad(dev(ops, _g4(ref_nhwc)), dev(ops, _g4(src_nhwc)), rt, dev(ops, disp_min), dev(ops, disp_max), D)
    out_g = ops.warp_corr_init_quad(dev(ops, ref_nhwc.contiguous()), dev(ops, src_nhwc.contiguous()), rt, dev(ops, disp_min), dev(ops, disp_max), D,
                                    plain=True, tune=K._lib.TUNE_SWEEP_GLOBAL)
    # long baselines: the fp32 projection chain of oracle and kernels differs by ~1e-5 texels per texel of disparity
    close(out, want, 1e-4 if isinstance(scene, bool) else 5e-4)
    assert torch.equal(out.cpu(), out_g.cpu())


# ------------------------------------------------------------------------------------------ 16-bit feature storage
_X16 = [torch.bfloat16, torch.float16]


@pytest.mark.parametrize("dt", _X16)
@pytest.mark.parametrize("cin,cout,k", [(64, 48, 1), (64, 32, 3), (24, 16, 3)])
def test_conv2d_16bit_channel_last_output(ops, dt, cin, cout, k):
    """FeatureNet's output convolutions storing bf16 / fp16 features: the fp32 result rounded to nearest even in the epilogue"""
    B, H, W = 2, 19, 37
    x, w = rnd(B, cin, H, W, seed=1), rnd(cout, cin, k, k, seed=2) * 0.2
    pc = K.pack_conv2d(dev(ops, w), pad=k // 2)
    full = ops.conv2d(pc, dev(ops, x), out_layout=K.LAYOUT_NHWC)
    got = ops.conv2d(pc, dev(ops, x), out_layout=K.LAYOUT_NHWC, out_dtype=dt)
    assert got.dtype == dt and got.shape == (B, H, W, cout)
    assert torch.equal(got.cpu(), full.cpu().to(dt))
    close(full.permute(0, 3, 1, 2), F.conv2d(x, w, None, 1, k // 2), 2e-5)


@pytest.mark.parametrize("dt", _X16)
@pytest.mark.parametrize("C,n", [(32, 6), (16, 4), (48, 4)])
def test_getcost_quad_16bit_features(ops, dt, C, n):
    """bf16 / fp16 feature storage: identical to the fp32 kernel (and the oracle) on the rounded feature values -- only the
    storage is reduced, the arithmetic is fp32"""
    B, S, H, W = 2, 3, 14, 22
    pm = _cams(B, S + 1, H, W, 2)
    feats = [rnd(B, C, H, W, seed=40 + v).to(dt).float() for v in range(S + 1)]
    inv = rnd(B, 1, H, W, seed=50, lo=-0.05, hi=1.05)
    conf = rnd(B, H, W, seed=51, lo=0.0, hi=1.0)
    vw = rnd(B, S, H // 2, W // 2, seed=52, lo=0.0, hi=1.0)
    dv0, dv1 = torch.tensor([1 / 935.0, 1 / 700.0]), torch.tensor([1 / 425.0, 1 / 450.0])
    dmax, dmin = (1 / dv0).view(-1, 1, 1, 1), (1 / dv1).view(-1, 1, 1, 1)
    want_cost, want_s = O.get_cost(feats, pm, inv, 2.0 / 384, dmax, dmin, n, F.interpolate(vw, scale_factor=2, mode="nearest"), conf, 4, 0.25, 4.0)
    rt = ops.compose_proj(dev(ops, pm))
    ref = feats[0].permute(0, 2, 3, 1).contiguous()
    src = torch.stack([f.permute(0, 2, 3, 1) for f in feats[1:]]).contiguous()
    tail = (rt, dev(ops, inv), dev(ops, conf), dev(ops, vw), dev(ops, 1 / (1 / dv0)), dev(ops, 1 / (1 / dv1)), n, 2.0 / 384, 0.25, 4.0)
    cost, samp = ops.getcost_quad(ref.to(dt).to(ops.device), src.to(dt).to(ops.device), *tail, vw_shift=1)
    cost32, _ = ops.getcost_quad(dev(ops, _g4(ref)), dev(ops, _g4(src)), *tail, vw_shift=1)
    close(samp, want_s, 1e-6)
    close(cost, want_cost, 1e-4)
    close(cost, cost32.cpu(), 2e-6)


@pytest.mark.parametrize("dt", _X16)
def test_warp_corr_init_quad_16bit_features(ops, dt):
    B, S, C, H, W, D = 2, 2, 48, 11, 17, 12
    feats = [rnd(B, C, H, W, seed=80 + v).to(dt).float() for v in range(S + 1)]
    _, proj, dvs = synth.synth_inputs(H * 8, W * 8, S, B=B, seed=7)
    pm = proj["stage1"]
    dv = torch.stack([dvs[:, 0], dvs[:, -1]], 1)
    hyp = (torch.arange(D).view(1, -1, 1, 1) / (D - 1.0)).repeat(B, 1, H, W)
    hyp = O.disp_to_depth(hyp, (1 / dv[:, 1]).view(-1, 1, 1, 1), (1 / dv[:, 0]).view(-1, 1, 1, 1))[1]
    ref_proj = O.compose_proj(pm[:, 0])
    want = torch.stack([O.group_corr(O.warp(feats[v], O.compose_proj(pm[:, v]), ref_proj, hyp), feats[0], 4) for v in range(1, S + 1)], 1)
    rt = ops.compose_proj(dev(ops, pm))
    ref = feats[0].permute(0, 2, 3, 1).contiguous().to(dt).to(ops.device)
    src = torch.stack([f.permute(0, 2, 3, 1) for f in feats[1:]]).contiguous().to(dt).to(ops.device)
    out = ops.warp_corr_init_quad(ref, src, rt, dev(ops, dv[:, 0].contiguous()), dev(ops, dv[:, 1].contiguous()), D)
    close(out, want, 1e-4)


@pytest.mark.parametrize("cin,cout,D,H,W,with_res", [(16, 8, 6, 9, 21, True), (8, 8, 5, 4, 16, False), (12, 6, 3, 5, 33, True), (4, 3, 2, 2, 2, False),
                                                     (32, 16, 5, 6, 20, True), (20, 12, 3, 4, 17, False)])
def test_deconv3d_matrix_core_form(ops, cin, cout, D, H, W, with_res):
    """transposed conv, stride 2, output_padding 1 with cout <= 8 on the matrix cores (CostRegNet conv7): several tiles per
    axis, ragged last tiles, channel counts that do not fill the 4-channel MFMA groups, both x-parities sharing the A rows"""
    B = 2
    x = rnd(B, cin, D, H, W, seed=1)
    w = rnd(cin, cout, 3, 3, 3, seed=2) * 0.2
    bn = {"weight": rnd(cout, seed=4, lo=0.5, hi=1.5), "bias": rnd(cout, seed=5),
          "running_mean": rnd(cout, seed=6), "running_var": rnd(cout, seed=7, lo=0.5, hi=1.5)}
    ref = F.relu(F.batch_norm(F.conv_transpose3d(x, w, None, 2, 1, 1), bn["running_mean"], bn["running_var"], bn["weight"], bn["bias"], False, 0.0, 1e-5))
    res = rnd(*ref.shape, seed=9) if with_res else None
    if with_res:
        ref = ref + res
    pc = K.pack_conv3d(dev(ops, w), bn={k_: v.to(ops.device) for k_, v in bn.items()}, stride=2, transposed=True)
    out = ops.conv3d(pc, dev(ops, x), act=K.ACT_RELU, residual=dev(ops, res))
    close(out, ref, 2e-5)


# ---------------------------------------------------------------------------------------------------------------------------------------
# Resident, tile-walking kernels at sizes where a workgroup really walks several tiles ON THE GPU (the op-level tests above are sized
# for the host emulation, whose "resident" grid is 2 workgroups: on the MI355X their grids fit the chip and nothing walks).  The round-4
# stem kernel read a halo whose LDS-DMA was still landing from the second tile of a workgroup on -- about one launch in ten at this size --
# and no op-level test could see it.  Each kernel: REPS launches bit-identical, and right against ATen.
_WALK_REPS = int(os.environ.get("DMVS_WALK_REPS", "24"))      # (a soak run raises it)


def _same_every_launch(fn, reps=_WALK_REPS):
    first = fn()
    torch.cuda.synchronize()
    for r in range(1, reps):
        again = fn()
        torch.cuda.synchronize()
        ne = int((again != first).sum())
        assert ne == 0, f"launch {r} differs from launch 0 in {ne} of {first.numel()} elements: the kernel is not reproducible"
    return first


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,W,tune", [(1, 864, 1152, 0), (1, 864, 1152, K._lib.TUNE_PIECES4), (2, 1056, 1920, 0), (3, 500, 1004, 0), (1, 600, 1003, 0)])
def test_featurenet_stem_walking_tiles_reproducible(N, H, W, tune):
    """3-6 tiles per resident workgroup, both halo forms, rows that are / are not 16-byte multiples"""
    from conftest import hip_ops
    ops = hip_ops()
    x = rnd(N, 3, H, W, seed=1)
    w0, w1 = rnd(8, 3, 3, 3, seed=2) * 0.4, rnd(8, 8, 3, 3, seed=3) * 0.3
    b0, b1 = rnd(8, seed=4), rnd(8, seed=5)
    pc0, pc1 = K.pack_conv2d(*dev(ops, w0, b0), pad=1), K.pack_conv2d(*dev(ops, w1, b1), pad=1)
    xd = dev(ops, x)
    out = _same_every_launch(lambda: ops.featurenet_stem(pc0, pc1, xd, tune=tune))
    ref = F.relu(F.conv2d(F.relu(F.conv2d(x, w0, b0, 1, 1)), w1, b1, 1, 1))
    close(out, ref, 2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,k,stride,B,H,W", [(16, 16, 3, 1, 8, 432, 576), (8, 16, 5, 2, 8, 864, 1152), (32, 32, 3, 1, 12, 264, 480),
                                                     (16, 16, 3, 1, 2, 528, 960), (8, 8, 3, 1, 12, 528, 960)])
def test_conv2d_walking_tiles_reproducible(cin, cout, k, stride, B, H, W):
    """the resident tile-walking instantiations of the tiled 2-D convolution (FeatureNet / ContextNet layers at cfg3 / cfg5 sizes)"""
    from conftest import hip_ops
    ops = hip_ops()
    x, w, b = rnd(B, cin, H, W, seed=1), rnd(cout, cin, k, k, seed=2) * 0.2, rnd(cout, seed=3)
    pc = K.pack_conv2d(*dev(ops, w, b), stride=stride, pad=k // 2)
    xd = dev(ops, x)
    out = _same_every_launch(lambda: ops.conv2d(pc, xd, act=K.ACT_RELU))
    close(out, F.relu(F.conv2d(x, w, b, stride, k // 2)), 3e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,N,D,H,W", [(4, 8, 7, 48, 108, 144), (4, 8, 11, 96, 132, 240), (8, 8, 1, 96, 132, 240), (8, 1, 7, 48, 108, 144)])
def test_conv3d_streamed_tiles_reproducible(cin, cout, N, D, H, W):
    """the persistent, tile-pipelined 3-D kernels (PixelViewWeight / CostRegNet conv0-1 shapes at cfg3 / cfg5 sizes)"""
    from conftest import hip_ops
    ops = hip_ops()
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn(N, cin, D, H, W, generator=g, device="cuda")
    w = torch.randn(cout, cin, 3, 3, 3, generator=g, device="cuda") * 0.2
    pc = K.pack_conv3d(w, None)
    out = _same_every_launch(lambda: ops.conv3d(pc, x, act=K.ACT_RELU), reps=12)
    n = min(N, 2)                                            # ATen on the host for two volumes is enough for the arithmetic
    ref = F.relu(F.conv3d(x[:n].cpu(), w.cpu(), None, 1, 1))
    close(out[:n], ref, 3e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("D,H,W,S", [(96, 132, 240, 11), (48, 108, 144, 7)])
def test_plane_sweep_band_reproducible(D, H, W, S):
    """the LDS-band plane sweep (its band is copied by LDS-DMA) at cfg5 / cfg3 stage-1 sizes: launches agree bit for bit, and with the
    global-memory form of the same arithmetic"""
    from conftest import hip_ops
    from diffmvs_amd import synth
    ops = hip_ops()
    proj, dv = synth.synth_cameras(H * 8, W * 8, S, B=1, numdepth=384)
    g = torch.Generator().manual_seed(3)
    perm = K.g4_channels(48)
    ref = torch.randn(1, H, W, 48, generator=g)[..., perm].contiguous().cuda()
    src = torch.randn(S, 1, H, W, 48, generator=g)[..., perm].contiguous().cuda()
    rt = ops.compose_proj(proj["stage1"].cuda().float().contiguous())
    kmin, kmax = dv[:, 0].contiguous().cuda(), dv[:, -1].contiguous().cuda()
    band = _same_every_launch(lambda: ops.warp_corr_init_quad(ref, src, rt, kmin, kmax, D), reps=8)
    glob = _same_every_launch(lambda: ops.warp_corr_init_quad(ref, src, rt, kmin, kmax, D, tune=K._lib.TUNE_SWEEP_GLOBAL), reps=8)
    assert torch.equal(band, glob)


@pytest.mark.parametrize("B,H,W", [(2, 16, 20), (1, 9, 13), (3, 32, 40)])
def test_mask_upsample4(ops, B, H, W):
    """The mask head's 1x1 layer (64 -> 144, bias, x 0.25; reference update.py:335-339, :473) + upsample_depth (module.py:237-248) +
    disp_to_depth in ONE kernel: against ATen + the oracle, and bit for bit against the two launches it replaces (conv2d then
    convex_upsample).  Sizes: whole pixel blocks, a ragged plane (117 pixels: a partly filled last wave, odd row length), several blocks
    per resident workgroup on the emulation."""
    x = F.relu(rnd(B, 64, H, W, seed=1))
    w, bias = rnd(144, 64, 1, 1, seed=2) * 0.4, rnd(144, seed=3)
    inv = rnd(B, 1, H, W, seed=4, lo=0, hi=1)
    lo, hi = torch.full((B,), 1 / 935.0), torch.full((B,), 1 / 425.0)
    pc = K.pack_conv2d(w, bias)
    pc = K.PackedConv(*dev(ops, pc.weight, None, pc.shift), pc.cin, pc.cout, pc.cout_pad, pc.k, pc.stride, pc.pad)
    mask = 0.25 * F.conv2d(x, w, bias)
    up = O.upsample_depth(inv, mask, 4)
    want_depth = O.disp_to_depth(up.unsqueeze(1), (1 / hi).view(-1, 1, 1, 1), (1 / lo).view(-1, 1, 1, 1))[1].squeeze(1)
    xd, invd, lod, hid = dev(ops, x, inv, lo, hi)
    g_inv, g_depth = ops.mask_upsample4(pc, xd, invd, lod, hid, post_scale=0.25, want_inv=True)
    close(g_inv, up, 2e-5)
    close(g_depth, want_depth, 2e-5)
    # the two launches the engine used until round 6: same products in the same order, same softmax arithmetic
    m2 = ops.conv2d(pc, xd, post_scale=0.25)
    t_inv, t_depth = ops.convex_upsample(invd, m2, lod, hid, 4)
    assert torch.equal(g_inv, t_inv) and torch.equal(g_depth, t_depth)
    # depth only (what the engine asks for)
    _, d_only = ops.mask_upsample4(pc, xd, invd, lod, hid, post_scale=0.25)
    assert torch.equal(d_only, t_depth)
    # anything but the DiffMVS head is refused (the engine keeps the two launches for those)
    pc36 = K.pack_conv2d(rnd(36, 64, 1, 1, seed=5), rnd(36, seed=6))
    with pytest.raises(K._lib.DmvsError):
        ops.mask_upsample4(pc36, xd, invd, lod, hid)


@pytest.mark.parametrize("cin,cout,k,stride,pad,H,W,two", [(8, 16, 3, 1, 1, 12, 20, False), (13, 20, 3, 1, 1, 33, 40, True), (6, 10, 5, 2, 2, 24, 32, False),
                                                            (16, 8, 7, 1, 3, 17, 24, False), (12, 20, 1, 1, 0, 9, 16, False), (9, 7, (1, 5), 1, (0, 2), 10, 12, False)])
def test_conv2d_wgrad_16_byte_staging_pieces(ops, cin, cout, k, stride, pad, H, W, two):
    """the weight-gradient kernel with both tiles staged in 16-byte LDS-DMA pieces (round 6: 9 instead of 27 DMA instructions per lane and
    tile) against autograd, and bit for bit against the 4-byte form (DMVS_TUNE_PIECES4): channel counts that are not multiples of the
    8-channel workgroup slice, a concatenated second input, ragged tiles in y, stride 2, several tiles per workgroup"""
    B = 2
    ks = (k, k) if isinstance(k, int) else k
    pd = (pad, pad) if isinstance(pad, int) else pad
    c1 = 5 if two else 0
    x0 = rnd(B, cin - c1, H, W, seed=1)
    x1 = rnd(B, c1, H, W, seed=2) if two else None
    x = (x0 if x1 is None else torch.cat([x0, x1], 1)).requires_grad_(False)
    w = (rnd(cout, cin, *ks, seed=3) * 0.2).requires_grad_(True)
    bias = rnd(cout, seed=4).requires_grad_(True)
    y = F.conv2d(x, w, bias, stride, pd)
    gy = rnd(*y.shape, seed=5)
    y.backward(gy)
    pc = K.pack_conv2d(w.detach(), bias.detach(), stride=stride, pad=pd)
    args = (pc, dev(ops, x0), dev(ops, gy), None if x1 is None else dev(ops, x1))
    gw, gb = ops.conv2d_wgrad(*args, want_bias=True)
    close(gw, w.grad, 2e-4)
    close(gb, bias.grad, 2e-4)
    gw4, gb4 = ops.conv2d_wgrad(*args, want_bias=True, tune=K._lib.TUNE_PIECES4)
    assert torch.equal(gw, gw4) and torch.equal(gb, gb4)
    # DMVS_TUNE_WGRAD_ACCUMULATE: the fold kernel adds into the caller's running gradient (the trainer's flat-bucket views)
    run_w, run_b = dev(ops, torch.full_like(w.detach(), 0.5), torch.full_like(bias.detach(), -1.0))
    for _ in range(2):
        ops.conv2d_wgrad(*args, want_bias=True, into_gw=run_w, into_gb=run_b)
    close(run_w, 0.5 + 2 * w.grad, 4e-4)
    close(run_b, -1.0 + 2 * bias.grad, 4e-4)
    assert torch.equal(run_w.cpu(), (0.5 + gw.cpu()) + gw.cpu())
