"""Timing of the homography-warp kernels on cuda:0: GetCost (quad per pixel; NHWC-g4 features and the training graph's plain NHWC
features) on scene geometry (hypotheses around the synthetic scene's true depth) and on noise geometry (hypotheses centred on
clamp(inv + 0.5 * randn), what the first GRU iteration of every diffusion stage and any untrained network feed the kernel),
and the stage-1 plane sweep (LDS band / texels from global memory).  One JSON line per case.  (Round 1's LDS-window and per-pixel
kernels, which this script used to time beside them, were retired in round 4: their numbers are in profiles/r2_warp_ab_b16.json.)"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffmvs_amd import synth  # noqa: E402
from diffmvs_amd.ops import Ops, g4_channels  # noqa: E402


def timeit(fn, iters):
    for _ in range(5):
        out = fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) * 1e3 / iters, out


def getcost_case(o, a, stage, C, n, geometry, conf):
    dev = o.device
    gi = synth.getcost_scene_inputs(a.H, a.W, a.src, a.batch, stage=stage, C=C, noise=0.01, conf=conf)
    h, w = gi["ref"].shape[1], gi["ref"].shape[2]
    inv = gi["inv"]
    cf = gi["conf"]
    if geometry == "noise":
        g = torch.Generator().manual_seed(5)
        inv = (inv + 0.5 * torch.randn(inv.shape, generator=g)).clamp(0, 1)
        if cf is not None:
            cf = torch.rand(cf.shape, generator=g)
    ref, src, vw = gi["ref"].to(dev), gi["src"].to(dev), gi["view_w"].to(dev)
    perm = g4_channels(C).to(dev)
    ref4, src4 = ref[..., perm].contiguous(), src[..., perm].contiguous()
    inv = inv.to(dev).contiguous()
    cf = None if cf is None else cf.to(dev).contiguous()
    rt = o.compose_proj(gi["proj"].to(dev).float().contiguous())
    kmin, kmax = gi["disp_min"].to(dev), gi["disp_max"].to(dev)
    tail = (rt, inv, cf, vw, kmin, kmax, n, gi["interval"], a.min_radius, a.max_radius, gi["vw_shift"])
    hw = h * w
    alg = 4.0 * a.batch * hw * (C + a.src * C + n + a.src + 4 * n)          # bench.py's formula
    res = {"case": "getcost", "stage": stage, "C": C, "n": n, "geometry": geometry, "conf": conf, "B": a.batch, "hw": [h, w],
           "algorithmic_MB": round(alg / 1e6, 2)}
    t_q, out_q = timeit(lambda: o.getcost_quad(ref4, src4, *tail), a.iters)
    t_p, out_p = timeit(lambda: o.getcost_quad(ref.contiguous(), src.contiguous(), *tail, plain=True), a.iters)
    for name, t, out in (("quad_g4", t_q, out_q), ("quad_plain", t_p, out_p)):
        res[name + "_us"] = round(t, 2)
        res[name + "_frac"] = round(alg / (t * 1e-6) / 8e12, 4)
    res["plain_bit_identical_to_g4"] = bool(torch.equal(out_q[0], out_p[0]))
    print(json.dumps(res), flush=True)


def init_case(o, a, C, D):
    dev = o.device
    imgs, proj, dv = synth.synth_inputs(a.H, a.W, a.src, B=a.batch, seed=0)
    h, w = a.H // 8, a.W // 8
    g = torch.Generator().manual_seed(1)
    ref = torch.randn(a.batch, h, w, C, generator=g).to(dev)
    src = torch.randn(a.src, a.batch, h, w, C, generator=g).to(dev)
    perm = g4_channels(C).to(dev)
    ref4, src4 = ref[..., perm].contiguous(), src[..., perm].contiguous()
    rt = o.compose_proj(proj["stage1"].to(dev).float().contiguous())
    kmin, kmax = dv[:, 0].contiguous().to(dev), dv[:, -1].contiguous().to(dev)
    alg = 4.0 * a.batch * h * w * (C + a.src * C + a.src * 4 * D)
    res = {"case": "warp_init", "C": C, "D": D, "B": a.batch, "hw": [h, w], "algorithmic_MB": round(alg / 1e6, 2)}
    from diffmvs_amd import _lib
    t_b, out_b = timeit(lambda: o.warp_corr_init_quad(ref4, src4, rt, kmin, kmax, D), a.iters)
    t_q, out_q = timeit(lambda: o.warp_corr_init_quad(ref4, src4, rt, kmin, kmax, D, tune=_lib.TUNE_SWEEP_GLOBAL), a.iters)
    for name, t in (("band", t_b), ("global", t_q)):
        res[name + "_us"] = round(t, 2)
        res[name + "_frac"] = round(alg / (t * 1e-6) / 8e12, 4)
    res["bit_identical"] = bool(torch.equal(out_b, out_q))
    print(json.dumps(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--H", type=int, default=512)
    ap.add_argument("--W", type=int, default=640)
    ap.add_argument("--src", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--min-radius", type=float, default=0.25)
    ap.add_argument("--max-radius", type=float, default=4.0)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--init-only", action="store_true", help="only the stage-1 plane sweep")
    a = ap.parse_args()
    o = Ops.for_device("cuda:0")
    if a.init_only:
        return init_case(o, a, 48, 48)
    for geometry, conf in (("noise", None), ("noise", 0.5), ("scene", 0.5), ("scene", 0.9)):
        getcost_case(o, a, 2, 32, 6, geometry, conf)
    if not a.quick:
        getcost_case(o, a, 3, 16, 4, "scene", 0.5)
        getcost_case(o, a, 3, 16, 4, "noise", None)
    init_case(o, a, 48, 48)


if __name__ == "__main__":
    main()
