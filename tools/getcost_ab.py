"""A/B timing of the two GetCost kernels (LDS-window vs per-pixel gather) on cuda:0 at a BASELINE config's stage size,
with the synthetic scene's real geometry (hypotheses around the ground-truth depth + noise)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffmvs_amd import synth  # noqa: E402
from diffmvs_amd.ops import Ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--H", type=int, default=512)
    ap.add_argument("--W", type=int, default=640)
    ap.add_argument("--src", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--stage", type=int, default=2)
    ap.add_argument("--C", type=int, default=32)
    ap.add_argument("--n", type=int, default=6)
    ap.add_argument("--conf", type=float, default=0.5)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--noise", type=float, default=0.01, help="sigma of the noise added to the true normalised inverse depth")
    ap.add_argument("--lib", default=None, help="alternative libdmvs .so (kernel experiments)")
    a = ap.parse_args()
    o = Ops.for_device("cuda:0")
    if a.lib:
        from diffmvs_amd import _lib
        o = Ops(_lib.Lib(os.path.abspath(a.lib)), "cuda:0")
    dev = o.device
    imgs, proj, dv, gt, _ = synth.synth_inputs(a.H, a.W, a.src, B=a.batch, seed=0, with_gt=True)
    name = f"stage{a.stage}"
    sc = 2 ** (4 - a.stage)
    h, w = a.H // sc, a.W // sc
    g = torch.Generator().manual_seed(1)
    ref = torch.randn(a.batch, h, w, a.C, generator=g).to(dev)
    src = torch.randn(a.src, a.batch, h, w, a.C, generator=g).to(dev)
    rt = o.compose_proj(proj[name].to(dev).float().contiguous())
    kmin, kmax = dv[:, 0].contiguous().to(dev), dv[:, -1].contiguous().to(dev)
    d = gt[name]
    d = torch.where(torch.isfinite(d) & (d > 0), d, torch.full_like(d, 600.0)).to(dev)
    inv = ((1.0 / d) - kmin.view(-1, 1, 1)) / (kmax - kmin).view(-1, 1, 1)
    inv = (inv + a.noise * torch.randn(inv.shape, generator=g).to(dev)).clamp(0, 1).unsqueeze(1).contiguous()
    conf = torch.full((a.batch, h, w), a.conf, device=dev) if a.conf >= 0 else None
    vshift = a.stage - 1
    vw = torch.rand(a.batch, a.src, h >> vshift, w >> vshift, generator=g).to(dev)
    interval = (1.0 / dv.shape[1]) * (2 if a.stage == 2 else 1)
    res = {}
    outs = {}
    for gather in (False, True):
        args = (ref, src, rt, inv, conf, vw, kmin, kmax, a.n, interval, 0.2, 2.0, vshift)
        for _ in range(5):
            outs[gather] = o.getcost(*args, gather=gather)
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(a.iters):
            o.getcost(*args, gather=gather)
        en.record()
        torch.cuda.synchronize()
        res["gather_us" if gather else "window_us"] = st.elapsed_time(en) * 1e3 / a.iters
    torch.cuda.synchronize()
    o.getcost(*args)
    res["adaptive_state"] = {str(k): {"gather": v["gather"], "last": v.get("last")} for k, v in o._getcost_state.items()}
    diff = float((outs[False][0] - outs[True][0]).abs().max() / outs[True][0].abs().max())
    hw = h * w
    alg = 4.0 * a.batch * (a.C * hw + a.src * a.C * hw + hw + (hw >> (2 * vshift)) * a.src + (hw if conf is not None else 0)
                           + 4 * a.n * hw + a.n * hw)
    res.update(config=vars(a), max_rel_diff=diff, algorithmic_MB=alg / 1e6,
               window_frac_of_8TBs=alg / (res["window_us"] * 1e-6) / 8e12, gather_frac_of_8TBs=alg / (res["gather_us"] * 1e-6) / 8e12)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
