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
    ap.add_argument("--bwd", action="store_true", help="time the backward (grad_ref / grad_src) instead of the forward")
    ap.add_argument("--lib", default=None, help="alternative libdmvs .so (kernel experiments)")
    a = ap.parse_args()
    o = Ops.for_device("cuda:0")
    if a.lib:
        from diffmvs_amd import _lib
        o = Ops(_lib.Lib(os.path.abspath(a.lib)), "cuda:0")
    dev = o.device
    gi = synth.getcost_scene_inputs(a.H, a.W, a.src, a.batch, stage=a.stage, C=a.C, noise=a.noise, conf=a.conf)
    h, w = gi["ref"].shape[1], gi["ref"].shape[2]
    ref, src, inv, vw = gi["ref"].to(dev), gi["src"].to(dev), gi["inv"].to(dev), gi["view_w"].to(dev)
    conf = None if gi["conf"] is None else gi["conf"].to(dev)
    rt = o.compose_proj(gi["proj"].to(dev).float().contiguous())
    kmin, kmax = gi["disp_min"].to(dev), gi["disp_max"].to(dev)
    interval, vshift = gi["interval"], gi["vw_shift"]
    res = {}
    outs = {}
    args = (ref, src, rt, inv, conf, vw, kmin, kmax, a.n, interval, 0.2, 2.0, vshift)
    gcost = torch.randn(a.batch, 4 * a.n, h, w, device=dev)
    gbuf = torch.zeros_like(src)
    for gather in (False, True):
        def run():
            if a.bwd:
                return o.getcost_bwd(*args, gcost, gsrc=gbuf.zero_(), gather=gather)
            return o.getcost(*args, gather=gather)
        for _ in range(5):
            outs[gather] = [t.clone() for t in run()]
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(a.iters):
            run()
        en.record()
        torch.cuda.synchronize()
        res["gather_us" if gather else "window_us"] = st.elapsed_time(en) * 1e3 / a.iters
    torch.cuda.synchronize()
    o.getcost(*args)
    res["adaptive_state"] = {str(k): {"gather": v["gather"], "last": v.get("last")} for k, v in o._getcost_state.items()}
    diff = max(float((x - y).abs().max() / y.abs().max()) for x, y in zip(outs[False], outs[True]))
    hw = h * w
    alg = 4.0 * a.batch * (a.C * hw + a.src * a.C * hw + hw + (hw >> (2 * vshift)) * a.src + (hw if conf is not None else 0)
                           + 4 * a.n * hw + a.n * hw)
    res.update(config=vars(a), max_rel_diff=diff, algorithmic_MB=alg / 1e6,
               window_frac_of_8TBs=alg / (res["window_us"] * 1e-6) / 8e12, gather_frac_of_8TBs=alg / (res["gather_us"] * 1e-6) / 8e12)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
