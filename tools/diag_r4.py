"""Kernel A/B diagnostics (rounds 4 and 5), one mode per gpurun call, JSON lines on stdout:

    python tools/diag_r4.py getcost     # GetCost at the bench batch on noise / random-confidence / scene geometry: the product library and
                                        # every tools/calib/libdmvs_hip_<name>.so present (tools/build_variant.py): gcexp1 no scattered traffic,
                                        # gcexp2 no hat / scatter, gcexp4[t1|t4] = the ceiling probe (loads only), gcc256s6 = the old 64-pixel
                                        # row-segment mapping, gcsame = the product source built like the variants (compare variants with it)
    python tools/diag_r4.py getcost_pmc # the product GetCost alone (for rocprofv3 --pmc passes)
    python tools/diag_r4.py warp_init   # the plane sweep against its diagnostic builds
    python tools/diag_r4.py pair3d      # the paired 3-D kernels against DMVS_TUNE3D_NO_PAIR, per layer
    python tools/diag_r4.py convexp | convexp2 | mtsweep | optins | stem     # conv2d tile shapes / 16-byte pieces, per layer
DIAG_DEVICE=cpu dry-runs a mode on the host emulation with every shape shrunk.
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffmvs_amd import _lib, synth  # noqa: E402
from diffmvs_amd import ops as K  # noqa: E402
from diffmvs_amd.ops import Ops, g4_channels  # noqa: E402


# DIAG_DEVICE=cpu runs a mode on the host emulation of the kernel sources (tests/hipemu) with every shape shrunk: a dry run of the script
# itself in the build container (no GPU), so that a typo does not cost a GPU session
DEV = os.environ.get("DIAG_DEVICE", "cuda:0")
DRY = DEV == "cpu"


def _ops():
    if not DRY:
        return Ops.for_device(DEV)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from hipemu.build import build_emu
    return Ops(_lib.Lib(build_emu()), "cpu")


def _gen(seed=0):
    return torch.Generator(device=DEV).manual_seed(seed)


def _n(n, lo=1):       # batch-like extents shrink to `lo` in a dry run
    return lo if DRY else n


def _hw(v, div=8):     # spatial extents shrink by `div` in a dry run (kept multiples of 4)
    return max(8, (v // div) // 4 * 4) if DRY else v


def timeit(fn, iters=20, warm=12):      # (warm: the first measurement after seconds of host-side input generation otherwise sees a GPU still
    # ramping its clocks -- up to 8 % on a 0.6 ms kernel; round 5 mistook that for a property of the build measured first)
    if DRY:
        import time
        fn()
        t0 = time.perf_counter()
        fn()
        return (time.perf_counter() - t0) * 1e6
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) * 1e3 / iters


def getcost_inputs(o, B, geometry, conf):
    dev = o.device
    nb = min(B, 16)                     # 16 rendered scenes, further batch items are copies (like bench.py)
    gi = synth.getcost_scene_inputs(512, 640, 5, nb, stage=2, C=32, noise=0.01, conf=0.5)
    inv, cf = gi["inv"], gi["conf"]
    rep = (B + nb - 1) // nb

    def tile(t, dim=0):
        return torch.cat([t] * rep, dim)[:B] if dim == 0 else torch.cat([t] * rep, dim)[:, :B]

    g = torch.Generator().manual_seed(5)
    inv = tile(inv)
    cf = tile(cf)
    if geometry == "noise":
        inv = (inv + 0.5 * torch.randn(inv.shape, generator=g)).clamp(0, 1)
    if conf == "random":
        cf = torch.rand(cf.shape, generator=g)
    elif conf is None:
        cf = None
    perm = g4_channels(32)
    ref = tile(gi["ref"])[..., perm].contiguous().to(dev)
    src = tile(gi["src"], 1)[..., perm].contiguous().to(dev)
    vw = tile(gi["view_w"]).contiguous().to(dev)
    rt = o.compose_proj(tile(gi["proj"]).to(dev).float().contiguous())
    kmin, kmax = tile(gi["disp_min"]).to(dev), tile(gi["disp_max"]).to(dev)
    return (ref, src, rt, inv.to(dev).contiguous(), None if cf is None else cf.to(dev).contiguous(), vw, kmin, kmax, 6, gi["interval"],
            0.25, 4.0, gi["vw_shift"])


def getcost(B=None):
    B = B or int(os.environ.get("DIAG_B", "96"))
    base = Ops.for_device("cuda:0")
    libs = [("product", None)] + [(n, os.path.join(ROOT, "tools", "calib", "libdmvs_hip_%s.so" % n)) for n in ("nopipe", "gcexp1", "gcexp2", "gcexp3", "gcexp4", "gcexp4t4", "gcexp4t1", "gcc256s6", "gcsame")]      # gcsame = the product source built as a variant (no flags): variants are compared with variants
    alg = 4.0 * B * 128 * 160 * (32 + 5 * 32 + 6 + 5 + 24)
    for geometry, conf in (("noise", None), ("noise", "random"), ("scene", 0.5)):
        args = getcost_inputs(base, B, geometry, conf)
        for name, path in libs:
            if path is not None and not os.path.exists(path):
                continue
            o = base if path is None else Ops(_lib.Lib(path), "cuda:0")
            out = o.getcost_quad(*args)[0]
            if name == "product":
                ref_out = out.clone()
            us = timeit(lambda: o.getcost_quad(*args))
            print(json.dumps({"diag": "getcost", "B": B, "geometry": geometry, "conf": conf, "build": name, "us": round(us, 1),
                              "frac_of_8TBs": round(alg / (us * 1e-6) / 8e12, 4),
                              "bit_identical_to_product": bool(torch.equal(out, ref_out)) if (name in ("product", "nopipe") or name.startswith("gcblk") or name.startswith("gcc") or name == "gcsame") else None}), flush=True)


def warp_init(B=96):
    """the stage-1 plane sweep at the bench batch: product build against gcexp2 (hat weights + scatter removed) and gcexp3 (one
    64-byte unit of the three per texel read): how much of the launch is the per-texel vector work"""
    base = Ops.for_device("cuda:0")
    dev = base.device
    proj, dv = synth.synth_cameras(512, 640, 5, B=16, numdepth=384)
    rep = B // 16
    g = torch.Generator().manual_seed(1)
    perm = g4_channels(48)
    ref = torch.randn(B, 64, 80, 48, generator=g)[..., perm].contiguous().to(dev)
    src = torch.randn(5, B, 64, 80, 48, generator=g)[..., perm].contiguous().to(dev)
    rt = base.compose_proj(torch.cat([proj["stage1"]] * rep, 0).to(dev).float().contiguous())
    dvb = torch.cat([dv] * rep, 0)
    kmin, kmax = dvb[:, 0].contiguous().to(dev), dvb[:, -1].contiguous().to(dev)
    alg = 4.0 * B * 64 * 80 * (48 + 5 * 48 + 5 * 4 * 48)
    for name in ("product", "gcexp2", "gcexp3"):
        path = os.path.join(ROOT, "tools", "calib", "libdmvs_hip_%s.so" % name)
        if name != "product" and not os.path.exists(path):
            continue
        o = base if name == "product" else Ops(_lib.Lib(path), "cuda:0")
        us = timeit(lambda: o.warp_corr_init_quad(ref, src, rt, kmin, kmax, 48), iters=10)
        print(json.dumps({"diag": "warp_init", "B": B, "build": name, "us": round(us, 1), "frac_of_8TBs": round(alg / (us * 1e-6) / 8e12, 4)}), flush=True)


def getcost_pmc(B=96):
    """the product kernel alone on the two noise geometries (for the rocprofv3 --pmc passes)"""
    o = Ops.for_device("cuda:0")
    for geometry, conf in (("noise", None), ("noise", "random")):
        args = getcost_inputs(o, B, geometry, conf)
        for _ in range(int(os.environ.get("DIAG_ITERS", "3"))):
            o.getcost_quad(*args)
        torch.cuda.synchronize()


def optins():
    o = Ops.for_device("cuda:0")
    g = torch.Generator(device="cuda").manual_seed(0)
    # fused stem, N = 96 x 6 images at 512 x 640
    x = torch.randn(96, 3, 512, 640, generator=g, device="cuda")
    w0, w1 = torch.randn(8, 3, 3, 3, generator=g, device="cuda") * 0.4, torch.randn(8, 8, 3, 3, generator=g, device="cuda") * 0.3
    b0, b1 = torch.randn(8, generator=g, device="cuda"), torch.randn(8, generator=g, device="cuda")
    pc0, pc1 = K.pack_conv2d(w0, b0, pad=1), K.pack_conv2d(w1, b1, pad=1)
    ref = o.featurenet_stem(pc0, pc1, x)
    for knob, tune in (("0", _lib.TUNE_PIECES4), ("1", 0)):      # (round 4, first session: the knob was an environment variable of the library)
        same = bool(torch.equal(ref, o.featurenet_stem(pc0, pc1, x, tune=tune)))
        us = timeit(lambda: o.featurenet_stem(pc0, pc1, x, tune=tune), iters=10)
        print(json.dumps({"diag": "stem", "DMVS_STEM_V16": knob, "us_per_96_images": round(us, 1), "bit_identical": same}), flush=True)
    del x, ref
    # 3-D MFMA kernels: PixelViewWeight conv0 (480 volumes), CostRegNet conv0 / conv1 (96 volumes), stride 1
    for name, N, cin, cout in (("pvw conv0 4->8 x480", 480, 4, 8), ("costreg conv0 4->8 x96", 96, 4, 8), ("costreg conv1 8->8 x96", 96, 8, 8),
                               ("costreg conv3 16->16 x96", 96, 16, 16), ("costreg conv5 32->32 x96", 96, 32, 32)):
        D, H, W = (48, 64, 80) if cin <= 8 else ((24, 32, 40) if cin == 16 else (12, 16, 20))
        v = torch.randn(N, cin, D, H, W, generator=g, device="cuda")
        w3 = torch.randn(cout, cin, 3, 3, 3, generator=g, device="cuda") * 0.2
        pc3 = K.pack_conv3d(w3, None)
        r3 = o.conv3d(pc3, v, act=K.ACT_RELU)
        for knob, tune in (("0", _lib.TUNE3D_PIECES4), ("1", 0)):
            same = bool(torch.equal(r3, o.conv3d(pc3, v, act=K.ACT_RELU, tune=tune)))
            us = timeit(lambda: o.conv3d(pc3, v, act=K.ACT_RELU, tune=tune), iters=10)
            print(json.dumps({"diag": "conv3d", "layer": name, "DMVS_CONV3D_V16": knob, "us": round(us, 1), "bit_identical": same}), flush=True)
        del v, r3
    # padding-pass skip (variant build) on the 16-channel layers it targets and two controls
    path = os.path.join(ROOT, "tools", "calib", "libdmvs_hip_padskip.so")
    if os.path.exists(path):
        o2 = Ops(_lib.Lib(path), "cuda:0")
        for name, N, cin, cout, k, H, W in (("16->16 3x3 256x320 x96", 96, 16, 16, 3, 256, 320), ("16->16 3x3 128x160 x96", 96, 16, 16, 3, 128, 160),
                                            ("32->32 3x3 128x160 x96", 96, 32, 32, 3, 128, 160), ("32->16 3x3 128x160 x96", 96, 32, 16, 3, 128, 160),
                                            ("64->64 3x3 64x80 x576", 576, 64, 64, 3, 64, 80)):
            xx = torch.randn(N, cin, H, W, generator=g, device="cuda")
            ww = torch.randn(cout, cin, k, k, generator=g, device="cuda") * 0.1
            pc = K.pack_conv2d(ww, None, pad=k // 2)
            a = o.conv2d(pc, xx, act=K.ACT_RELU)
            b = o2.conv2d(pc, xx, act=K.ACT_RELU)
            ua = timeit(lambda: o.conv2d(pc, xx, act=K.ACT_RELU), iters=10)
            ub = timeit(lambda: o2.conv2d(pc, xx, act=K.ACT_RELU), iters=10)
            print(json.dumps({"diag": "padskip", "layer": name, "product_us": round(ua, 1), "padskip_us": round(ub, 1),
                              "bit_identical": bool(torch.equal(a, b))}), flush=True)
            del xx, a, b


def convexp():
    """three convolution experiments at the bench batch, per layer (all bit-identical to the product, checked here):
    tall = DMVS_TUNE_TALL (16 x 32-pixel tiles for the one-n-tile plain 3x3 layers); s2b64 / ky55 / s2b64ky55 = variant builds
    (8-byte LDS reads of the stride-2 B operand; all five tap rows of a one-n-tile 5x5 layer in one loop trip)"""
    o = _ops()
    g = _gen()
    var = {}
    for n in ("s2b64", "ky55", "s2b64ky55"):
        path = os.path.join(ROOT, "tools", "calib", "libdmvs_hip_%s.so" % n)
        if os.path.exists(path):
            var[n] = Ops(_lib.Lib(path), "cuda:0")
    layers = [  # name, N, cin, cout, k, stride, H, W (input)
        ("16->16 3x3 256x320 x576", 576, 16, 16, 3, 1, 256, 320), ("16->16 3x3 128x160 x96", 96, 16, 16, 3, 1, 128, 160),
        ("16->16 3x3 256x320 x96", 96, 16, 16, 3, 1, 256, 320), ("16->16 3x3 64x80 x96", 96, 16, 16, 3, 1, 64, 80),
        ("32->16 3x3 128x160 x96", 96, 32, 16, 3, 1, 128, 160), ("64->16 3x3 128x160 x96", 96, 64, 16, 3, 1, 128, 160),
        ("8->16 5x5 s2 512x640 x576", 576, 8, 16, 5, 2, 512, 640), ("16->32 5x5 s2 256x320 x576", 576, 16, 32, 5, 2, 256, 320),
        ("32->64 5x5 s2 128x160 x576", 576, 32, 64, 5, 2, 128, 160), ("8->16 3x3 s2 512x640 x96", 96, 8, 16, 3, 2, 512, 640),
        ("16->32 3x3 s2 256x320 x96", 96, 16, 32, 3, 2, 256, 320), ("32->48 3x3 s2 128x160 x96", 96, 32, 48, 3, 2, 128, 160)]
    for name, N, cin, cout, k, s, H, W in layers:
        N, H, W = _n(N, 2), (H if not DRY else max(64, H // 4)), _hw(W)
        xx = torch.randn(N, cin, H, W, generator=g, device=DEV)
        ww = torch.randn(cout, cin, k, k, generator=g, device=DEV) * 0.1
        pc = K.pack_conv2d(ww, None, stride=s, pad=k // 2)
        ref = o.conv2d(pc, xx, act=K.ACT_RELU)
        row = {"diag": "convexp", "layer": name, "product_us": round(timeit(lambda: o.conv2d(pc, xx, act=K.ACT_RELU), iters=10), 1)}
        flops = 2.0 * N * (H // s) * (W // s) * cin * cout * k * k
        row["product_frac"] = round(flops / (row["product_us"] * 1e-6) / 157.3e12, 3)
        row["product_us"] = round(timeit(lambda: o.conv2d(pc, xx, act=K.ACT_RELU, tune=_lib.TUNE_NO_TALL), iters=10), 1) if s == 1 else row["product_us"]
        if s == 1:
            y = o.conv2d(pc, xx, act=K.ACT_RELU, tune=_lib.TUNE_TALL)
            row["tall_bit_identical"] = bool(torch.equal(ref, y))
            row["tall_us"] = round(timeit(lambda: o.conv2d(pc, xx, act=K.ACT_RELU, tune=_lib.TUNE_TALL), iters=10), 1)
            del y
        else:
            for n, ov in var.items():
                y = ov.conv2d(pc, xx, act=K.ACT_RELU)
                row[n + "_bit_identical"] = bool(torch.equal(ref, y))
                row[n + "_us"] = round(timeit(lambda: ov.conv2d(pc, xx, act=K.ACT_RELU), iters=10), 1)
                del y
        print(json.dumps(row), flush=True)
        del xx, ref


def convexp2():
    """the tall-tile form with TWO n-tiles (DMVS_TUNE_TALL(2) on 17..32-output-channel plain 3x3 layers) against the product's choice
    (walking / 32-wide / 16 x 16 tiles), as a plain ReLU layer and as a Unet block convolution (GroupNorm statistics: cannot walk)"""
    o = Ops.for_device("cuda:0")
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, N, cin, cout, H, W in (("32->32 128x160 x96", 96, 32, 32, 128, 160), ("64->32 128x160 x576", 576, 64, 32, 128, 160),
                                     ("32->32 128x160 x576", 576, 32, 32, 128, 160), ("64->31 128x160 x96", 96, 64, 31, 128, 160),
                                     ("32->32 64x80 x96", 96, 32, 32, 64, 80), ("24->32 128x160 x96", 96, 24, 32, 128, 160),
                                     ("48->32 64x80 x96", 96, 48, 32, 64, 80), ("16->32 64x80 x96", 96, 16, 32, 64, 80)):
        xx = torch.randn(N, cin, H, W, generator=g, device="cuda")
        ww = torch.randn(cout, cin, 3, 3, generator=g, device="cuda") * 0.1
        pc = K.pack_conv2d(ww, None, pad=1)
        row = {"diag": "convexp2", "layer": name}
        for kind in ("relu", "gn") if cout % 4 == 0 else ("relu",):
            def call(tune):
                if kind == "relu":
                    return o.conv2d(pc, xx, act=K.ACT_RELU, tune=tune)
                return o.conv2d(pc, xx, gn_stats=torch.zeros(N * 8, dtype=torch.float64, device="cuda"), tune=tune)
            ref, y = call(0), call(_lib.TUNE_TALL)
            row[kind + "_bit_identical"] = bool(torch.equal(ref, y))
            del ref, y
            row[kind + "_product_us"] = round(timeit(lambda: call(0), iters=10), 1)
            row[kind + "_tall_us"] = round(timeit(lambda: call(_lib.TUNE_TALL), iters=10), 1)
        print(json.dumps(row), flush=True)
        del xx


def pair3d():
    """the paired 3-D kernels (the default since round 5) against the kernels that ran these layers before (DMVS_TUNE3D_NO_PAIR): PixelViewWeight
    conv0 / CostRegNet conv0 (4 -> 8) and CostRegNet conv1 (8 -> 8) of the B = 96 step, per layer, bit-identical"""
    o = _ops()
    g = _gen()
    for name, N, cin in (("pvw conv0 4->8 x480", 480, 4), ("costreg conv0 4->8 x96", 96, 4), ("costreg conv1 8->8 x96", 96, 8)):
        N = _n(N, 8)
        v = torch.randn(N, cin, 48, 64, 80, generator=g, device=DEV) if not DRY else torch.randn(N, cin, 16, 32, 80, generator=g, device=DEV)
        pc3 = K.pack_conv3d(torch.randn(8, cin, 3, 3, 3, generator=g, device=DEV) * 0.2, None)
        ref = o.conv3d(pc3, v, act=K.ACT_RELU)
        same = bool(torch.equal(ref, o.conv3d(pc3, v, act=K.ACT_RELU, tune=_lib.TUNE3D_NO_PAIR)))
        del ref
        print(json.dumps({"diag": "pair3d", "layer": name, "paired_us": round(timeit(lambda: o.conv3d(pc3, v, act=K.ACT_RELU), iters=10), 1),
                          "no_pair_us": round(timeit(lambda: o.conv3d(pc3, v, act=K.ACT_RELU, tune=_lib.TUNE3D_NO_PAIR), iters=10), 1),
                          "bit_identical": same}), flush=True)
        del v


def mtsweep():
    """(not yet run: prepared for the next round) every multi-tap conv2d shape of the B = 96 step (profiles/r4_conv2d_layers_b96_final.txt)
    with each tile shape forced -- 16 x 4 / 16 x 8 / 16 x 16 (DMVS_TUNE_TILE_MT), 16 x 32 (DMVS_TUNE_TALL(2)), 32-wide (DMVS_TUNE_TILE_WX(2))
    -- against the dispatcher's choice: the thresholds in launch_conv2d were set at B = 16 in round 2.  ReLU'd random inputs, plain
    ReLU layers (the fused variants of a shape share its tile shape)."""
    o = _ops()
    g = _gen()
    shapes = []
    for line in open(os.path.join(ROOT, "profiles", "r4_conv2d_layers_b96_final.txt")):
        if "|" not in line or line.startswith("B"):
            continue
        a, b = line.split("|")
        B, cin, cout, kh, kw, s, Ho, Wo, mode, gated = map(int, a.split())
        if kh * kw > 1 and mode == 0:
            shapes.append((B, cin, cout, kh, kw, s, Ho, Wo, int(b.split()[0])))
    variants = [("auto", 0), ("mt1", _lib.tune_tile_mt(1) | _lib.TUNE_NO_TALL), ("mt2", _lib.tune_tile_mt(2) | _lib.TUNE_NO_TALL),
                ("mt4", _lib.tune_tile_mt(4) | _lib.TUNE_NO_TALL), ("tall", _lib.TUNE_TALL), ("wx2", _lib.tune_tile_wx(2) | _lib.TUNE_NO_TALL),
                ("wx1", _lib.tune_tile_wx(1) | _lib.TUNE_NO_TALL), ("nowalk", _lib.TUNE_NO_WALK | _lib.TUNE_NO_TALL)]
    for B, cin, cout, kh, kw, s, Ho, Wo, launches in (shapes[::6] if DRY else shapes):
        B, Ho, Wo = _n(B, 2), _hw(Ho, 4), _hw(Wo, 4)
        xx = torch.relu(torch.randn(B, cin, Ho * s, Wo * s, generator=g, device=DEV))
        ww = torch.randn(cout, cin, kh, kw, generator=g, device=DEV) * 0.1
        pc = K.pack_conv2d(ww, None, stride=s, pad=(kh // 2, kw // 2))
        ref = o.conv2d(pc, xx, act=K.ACT_RELU)
        row = {"diag": "mtsweep", "shape": [B, cin, cout, kh, kw, s, Ho, Wo], "launches_per_step": launches}
        for name, tune in variants:
            try:
                y = o.conv2d(pc, xx, act=K.ACT_RELU, tune=tune)
            except _lib.DmvsError:
                continue
            if not torch.equal(ref, y):
                row[name + "_differs"] = True
            del y
            row[name + "_us"] = round(timeit(lambda: o.conv2d(pc, xx, act=K.ACT_RELU, tune=tune), iters=8, warm=2), 1)
        best = min((v, k) for k, v in row.items() if k.endswith("_us"))
        row["best"], row["gain_ms_per_step"] = best[1][:-3], round((row["auto_us"] - best[0]) * launches / 1e3, 3)
        print(json.dumps(row), flush=True)
        del xx, ref


def stem():
    """the fused FeatureNet stem at the bench size (96 images of 512 x 640), 16-byte and 4-byte halo pieces"""
    o = _ops()
    g = _gen()
    x = torch.randn(_n(96, 2), 3, _hw(512), _hw(640), generator=g, device=DEV)
    w0, w1 = torch.randn(8, 3, 3, 3, generator=g, device=DEV) * 0.4, torch.randn(8, 8, 3, 3, generator=g, device=DEV) * 0.3
    b0, b1 = torch.randn(8, generator=g, device=DEV), torch.randn(8, generator=g, device=DEV)
    pc0, pc1 = K.pack_conv2d(w0, b0, pad=1), K.pack_conv2d(w1, b1, pad=1)
    for name, tune in (("16-byte pieces", 0), ("4-byte pieces", _lib.TUNE_PIECES4)):
        us = timeit(lambda: o.featurenet_stem(pc0, pc1, x, tune=tune), iters=10)
        print(json.dumps({"diag": "stem", "form": name, "us_per_96_images": round(us, 1)}), flush=True)


if __name__ == "__main__":
    {"getcost": getcost, "warp_init": warp_init, "getcost_pmc": getcost_pmc, "optins": optins, "convexp": convexp, "convexp2": convexp2, "stem": stem, "pair3d": pair3d, "mtsweep": mtsweep}[sys.argv[1]]()
