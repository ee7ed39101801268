"""Evaluation driver: the counterpart of the reference's test.py:92-205 (save_scene_depth) on the MI355X path.

    python -m diffmvs_amd.eval --testpath <root> --dataset dtu --testlist lists/dtu/test.txt --loadckpt model.ckpt \\
        --outdir outputs --method casdiffmvs --num_view 5 [--filter]

For every reference view of every scene: build the sample (diffmvs_amd.formats.MVSDataset, the reference's dataset
contract), run CasDiffMVS.forward timed between device synchronisations exactly like test.py:122-127, and write
depth_est/*.pfm, conf{i}/*.pfm, cams/*_cam.txt, images/*.jpg in the reference's output layout, so that the reference's
filter.py -- or this package's GPU consistency filter (--filter, diffmvs_amd.fusion) -- can fuse them.  Scenes shard over
ranks (one process per GPU, RANK / WORLD_SIZE from the launcher) with no communication.
If <testpath>/<scan>/depth_gt/%08d.pfm (+ optional mask/%08d.png) exist, the absolute and relative depth errors of
utils.py:178-187 / BASELINE.json ("DTU abs-rel") are reported per scene."""
from __future__ import annotations

import argparse
import json
import os
import time

import numpy as np
import torch

from . import formats as IO
from . import shard, synth


def build_args(a) -> object:
    """the constructor namespace of CasDiffMVS from the command line (reference test.py:20-78 flag names)"""
    over = {}
    for k in ("stage_iters", "cost_dim_stage", "CostNum", "hidden_dim", "context_dim", "unet_dim", "scale", "sampling_timesteps", "ddim_eta"):
        v = getattr(a, k)
        if v is not None:
            over[k] = list(v)
    for k in ("min_radius", "max_radius"):
        if getattr(a, k) is not None:
            over[k] = getattr(a, k)
    return synth.make_args("casdiffmvs" if a.method == "casdiffmvs" else "diffmvs", numdepth_initial=a.numdepth_initial,
                           numdepth=a.numdepth, **over)


def run_scenes(model, a, scenes, device):
    """the per-scene loop of test.py:92-205.  With the scene cache (default) every image of a scene is read, decoded and pushed through
    FeatureNet ONCE (SceneFeatureStore: the features of all views stay resident in HBM) and each reference view gathers its rows;
    --scene_cache 0 is the reference's own order of work (every sample loads and encodes its V images again).  Same depth maps bit
    for bit (tests/test_scene.py); the timed region is the model call like test.py:122-127 -- the store's one-off cost is reported
    per scene in `feature_store_s`."""
    times, report, store_s = [], {}, {}
    first_calls, seen_geometry = [], set()      # with HIP graphs the first call of an input geometry = two warm-up forwards + the capture
    for scene in scenes:
        ds = IO.MVSDataset(a.testpath, a.num_view, a.numdepth, dataset=a.dataset, scan=[scene], max_h=a.max_h, max_w=a.max_w)
        views, store, row_of = None, None, None
        if a.scene_cache and len(ds):
            ids = sorted({v for i in range(len(ds)) for v in ds.view_ids(i)})
            views = {v: ds.load_view(scene, v) for v in ids}
            if len({views[v][0].shape for v in ids}) == 1:          # (a general dataset whose images differ in size cannot be stacked)
                torch.cuda.synchronize()
                t0 = time.time()
                stack = torch.from_numpy(np.stack([np.ascontiguousarray(views[v][0].transpose(2, 0, 1)) for v in ids]))
                store = model.scene_features(stack.to(device))
                torch.cuda.synchronize()
                store_s[scene] = time.time() - t0
                row_of = {v: r for r, v in enumerate(ids)}
        errs = []
        for i0 in range(0, len(ds), a.batch_size):
            idxs = list(range(i0, min(i0 + a.batch_size, len(ds))))
            sample = IO.collate([ds.sample_from_views(i, views) if views is not None else ds[i] for i in idxs])
            imgs = [t.to(device) for t in sample["imgs"]]
            proj = {k: v.to(device) for k, v in sample["proj_matrices"].items()}
            dv = sample["depth_values"].to(device)
            if store is not None:
                feats_ids = torch.tensor([[row_of[v] for v in ds.view_ids(i)] for i in idxs])
            torch.cuda.synchronize()
            t0 = time.time()
            with torch.no_grad():
                if store is not None:
                    out = model(imgs[:1], proj, dv, feats=store.gather(feats_ids))
                else:
                    out = model(imgs, proj, dv)
            torch.cuda.synchronize()
            geom = (tuple(imgs[0].shape), len(imgs), store is not None)
            if model.hip_graphs and geom not in seen_geometry:
                first_calls.append(time.time() - t0)             # reported on its own: not comparable to test.py:122-127's per-view time
            else:
                times.append(time.time() - t0)
            seen_geometry.add(geom)
            IO.save_outputs(a.outdir, sample, out)
            for b, pattern in enumerate(sample["filename"]):
                gt_file = os.path.join(a.testpath, pattern.format("depth_gt", ".pfm"))
                if os.path.exists(gt_file):
                    gt = torch.from_numpy(np.ascontiguousarray(IO.read_pfm(gt_file)[0]))[None]
                    est = out["depth"][-1][b:b + 1].float().cpu()
                    m = (gt > 1.0 / float(dv[b, -1])) & (gt < 1.0 / float(dv[b, 0])) if gt.shape == est.shape else None
                    if m is not None and bool(m.any()):
                        errs.append((float(IO.abs_depth_error(est, gt, m)), float(IO.abs_rel_error(est, gt, m))))
        if errs:
            report[scene] = {"abs_err": float(np.mean([e[0] for e in errs])), "abs_rel": float(np.mean([e[1] for e in errs])), "views": len(errs)}
    run_scenes.first_calls = first_calls
    return times, report, store_s


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--testpath", required=True)
    ap.add_argument("--dataset", default="dtu", choices=["dtu", "tank", "eth3d", "general"])
    ap.add_argument("--testlist", default=None, help="file with one scene per line (default: the single scene '' of a general dataset)")
    ap.add_argument("--loadckpt", default=None, help="reference checkpoint ({'model': state_dict}); default: seeded random weights")
    ap.add_argument("--outdir", default="./outputs")
    ap.add_argument("--method", default="casdiffmvs", choices=["casdiffmvs", "diffmvs"])
    ap.add_argument("--batch_size", type=int, default=1)
    ap.add_argument("--num_view", type=int, default=5, help="images per sample: 1 reference + num_view-1 sources")
    ap.add_argument("--numdepth", type=int, default=384)
    ap.add_argument("--numdepth_initial", type=int, default=48)
    ap.add_argument("--max_h", type=int, default=4800)
    ap.add_argument("--max_w", type=int, default=6400)
    for k, t in (("stage_iters", int), ("cost_dim_stage", int), ("CostNum", int), ("hidden_dim", int), ("context_dim", int),
                 ("unet_dim", int), ("scale", float), ("sampling_timesteps", int), ("ddim_eta", float)):
        ap.add_argument("--" + k, type=t, nargs=3, default=None)
    ap.add_argument("--min_radius", type=float, default=None)
    ap.add_argument("--max_radius", type=float, default=None)
    ap.add_argument("--filter", action="store_true", help="fuse the written depth maps with the GPU consistency filter afterwards")
    # post-processing thresholds: the reference's flags and defaults (test.py:69-76); Tanks&Temples / ETH3D scenes take their
    # per-scene tables instead (test.py:217-293), as in the reference
    ap.add_argument("--geo_mask_thres", type=int, default=2, help="depth must be consistent in at least N source views")
    ap.add_argument("--geo_pixel_thres", type=float, default=1.0, help="reprojection error threshold in pixels")
    ap.add_argument("--geo_depth_thres", type=float, default=0.01, help="relative depth error threshold")
    ap.add_argument("--photo_thres", type=float, nargs="+", default=[0.3, 0.0, 0.0], help="confidence threshold per stage")
    ap.add_argument("--scene_cache", type=int, default=1, choices=[0, 1],
                    help="1 (default): every image of a scene through FeatureNet once, features resident in HBM; 0: the reference's per-sample order")
    ap.add_argument("--graphs", type=int, default=None, choices=[0, 1],
                    help="run the forward through a captured HIP graph (default: 1 for --batch_size <= 8, where ~280 launches of 5-25 us are "
                         "host-bound -- the reference's own harness runs batch 1, test.py:101-104 -- unless --noise_seed asks for host-generated noise)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--noise_seed", type=int, default=None,
                    help="draw the diffusion noise from synth.NoiseSource(seed) (a host generator stream: the same depth maps on any "
                         "device and against the CPU oracle) instead of the device RNG of update.py:472")
    a = ap.parse_args(argv)

    rank, world, local = shard.env_rank_world()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    torch.manual_seed(a.seed + rank)
    from models import CasDiffMVS
    model = CasDiffMVS(build_args(a), test=True).eval()
    if a.loadckpt:
        sd = torch.load(a.loadckpt, map_location="cpu")
        model.load_state_dict(sd["model"], strict=False)                     # test.py:108-109
    else:
        model.load_state_dict(synth.synth_state_dict(model.state_dict(), 123))
    model.to(device)
    if a.noise_seed is not None:
        model.noise_source = synth.NoiseSource(a.noise_seed)
    # a graph replays fixed device work: the diffusion noise must then come from the device RNG (torch keeps its generator graph-safe)
    model.hip_graphs = bool(a.graphs) if a.graphs is not None else (a.batch_size <= 8 and a.noise_seed is None)
    if model.hip_graphs and a.noise_seed is not None:
        raise SystemExit("--graphs 1 needs the device RNG: drop --noise_seed")
    scenes = [""]
    if a.testlist:
        with open(a.testlist) as f:
            scenes = [ln.strip() for ln in f if ln.strip()]
    mine = shard.shard_scenes(scenes, rank, world)
    times, report, store_s = run_scenes(model, a, mine, device)
    first = getattr(run_scenes, "first_calls", [])
    calls = len(times) + len(first)
    # avg_time_s: the model call per batch like test.py:122-127 (graph-capture calls excluded, listed in first_call_s); amortised_time_s: what a
    # scene really costs per call -- every timed call, the capture calls AND the feature store's one-off FeatureNet pass over the scene's images
    res = {"rank": rank, "scenes": mine, "views": calls, "avg_time_s": float(np.mean(times)) if times else None,
           "first_call_s": [round(t, 4) for t in first],
           "amortised_time_s": float((sum(times) + sum(first) + sum(store_s.values())) / calls) if calls else None,
           "errors": report, "feature_store_s": store_s, "hip_graphs": bool(model.hip_graphs)}
    if a.filter:
        from . import fusion
        for scene in mine:      # the per-dataset protocol of test.py:298-367
            kw = fusion.scene_protocol(a.dataset, scene, a.outdir, a.geo_mask_thres, a.geo_pixel_thres, a.geo_depth_thres, a.photo_thres)
            n = fusion.filter_depth(os.path.join(a.testpath, scene), os.path.join(a.outdir, scene), method=a.method, device=device, **kw)
            res.setdefault("fused_points", {})[scene] = n
            res.setdefault("ply", {})[scene] = kw["plyfilename"]
    print(json.dumps(res), flush=True)
    return res


if __name__ == "__main__":
    main()
