"""Round 6: would a texel-PAIR load unit pay for the 16-bit C = 16 features of the cfg3 / cfg5 stage-3 GetCost?

A 16-bit stage-3 texel is 32 bytes -- a quarter of a cache line -- and the quad kernel issues one request per texel, so halving the element
size halved the bytes but not the requests (GetCost sits at 0.10-0.14 of the HBM roof on the cfg3 / cfg5 lines).  Every bilinear footprint
touches (x0, x0 + 1), so a layout whose load unit is the pair (64 contiguous bytes, one 16-byte load per lane) would need about half the
requests.  This times, on cuda:0, for one stage-3 launch at the given size, on noise and on scene geometry:
  * the product (dmvs_getcost_quad_f32, fp16 features),
  * its loads-only probe (libdmvs_probe.so: the same address stream, nothing computed),
  * the loads-only probe with PAIR units (dmvs_probe_getcost_pair_loads_f32).
One JSON line per case.  The pair layout is worth building only if the pair probe is >= 15 % below the single-texel probe.
    python tools/pair_probe.py [--H 1056 --W 1920 --src 11 --batch 2]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (ProbeLib)
from diffmvs_amd import synth  # noqa: E402
from diffmvs_amd.ops import Ops  # noqa: E402


def timeit(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) * 1e3 / iters


def case(o, probe, a, geometry, conf, dtype):
    dev = o.device
    C, n, stage = 16, 4, 3
    gi = synth.getcost_scene_inputs(a.H, a.W, a.src, a.batch, stage=stage, C=C, noise=0.01, conf=conf)
    h, w = gi["ref"].shape[1], gi["ref"].shape[2]
    inv, cf = gi["inv"], gi["conf"]
    if geometry == "noise":
        g = torch.Generator().manual_seed(5)
        inv = (inv + 0.5 * torch.randn(inv.shape, generator=g)).clamp(0, 1)
        if cf is not None:
            cf = torch.rand(cf.shape, generator=g)
    ref = gi["ref"].to(dev).to(dtype).contiguous()
    # the pair unit of an image row's last texel reads 32 bytes of the next row: one spare row behind the stack keeps the last one in bounds
    flat = torch.zeros(a.src * a.batch * h * w * C + w * C, dtype=dtype, device=dev)
    src = flat[:a.src * a.batch * h * w * C].view(a.src, a.batch, h, w, C)
    src.copy_(gi["src"].to(dev))
    vw = gi["view_w"].to(dev)
    inv = inv.to(dev).contiguous()
    cf = None if cf is None else cf.to(dev).contiguous()
    rt = o.compose_proj(gi["proj"].to(dev).float().contiguous())
    kmin, kmax = gi["disp_min"].to(dev), gi["disp_max"].to(dev)
    tail = (rt, inv, cf, vw, kmin, kmax, n, gi["interval"], a.min_radius, a.max_radius, gi["vw_shift"])
    captured = {}
    o.getcost_hook = lambda d, t: captured.update(d=d, t=t)
    o.getcost_quad(ref, src, *tail)
    o.getcost_hook = None
    d = captured["d"]
    stream = o.stream()
    es = ref.element_size()
    alg = a.batch * h * w * (es * (C + a.src * C) + 4 * (n + a.src + 4 * n))
    res = {"case": "getcost stage 3", "dtype": str(dtype).replace("torch.", ""), "C": C, "n": n, "geometry": geometry, "conf": conf, "B": a.batch,
           "S": a.src, "hw": [h, w], "algorithmic_MB": round(alg / 1e6, 2)}
    for name, fn in (("product", lambda: o.getcost_quad(ref, src, *tail)), ("loads_probe", lambda: probe.getcost_loads(d, stream)),
                     ("pair_loads_probe", lambda: probe.getcost_pair_loads(d, stream)), ("product_again", lambda: o.getcost_quad(ref, src, *tail))):
        t = timeit(fn, a.iters)
        res[name + "_us"] = round(t, 2)
        res[name + "_frac_of_hbm_peak"] = round(alg / (t * 1e-6) / 8e12, 4)
    res["pair_over_single_probe"] = round(res["pair_loads_probe_us"] / res["loads_probe_us"], 3)
    print(json.dumps(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--H", type=int, default=1056)
    ap.add_argument("--W", type=int, default=1920)
    ap.add_argument("--src", type=int, default=11)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--min-radius", type=float, default=0.25)
    ap.add_argument("--max-radius", type=float, default=4.0)
    a = ap.parse_args()
    o = Ops.for_device("cuda:0")
    probe = bench.ProbeLib()
    for geometry, conf in (("noise", None), ("noise", 0.5), ("scene", 0.5), ("scene", 0.9)):
        case(o, probe, a, geometry, conf, torch.float16)
    case(o, probe, a, "noise", None, torch.bfloat16)


if __name__ == "__main__":
    main()
