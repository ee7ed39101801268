"""Depth-map filtering and fusion on the MI355X: the reference's filter.py:95-227 (filter_depth) and
:261-441 (filter_depth_dynamic, Tanks&Temples) with the per-pixel geometry -- reprojection into every source view, bilinear
sampling of its depth map, back-projection, the distance / relative-depth tests -- in ONE kernel launch per reference view
(dmvs_geo_consistency_f32, csrc/fusion.hip) instead of ~40 NumPy passes + cv2.remap per source view.  File formats are the
reference's (diffmvs_amd/formats.py): it reads the tree test.py / diffmvs_amd.eval writes and emits the same masks and
binary PLY.  The camera matrices are composed on the host in fp32 exactly as the reference's NumPy code does."""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Sequence

import numpy as np
import torch

from . import _lib
from . import formats as IO
from .ops import Ops, _ptr

# filter.py:271-288: per-scene parameters of the dynamic check
DH_VIEW_NUM = {"Family": 2, "Francis": 9, "Horse": 2, "Lighthouse": 6, "M60": 4, "Panther": 3, "Playground": 6, "Train": 3,
               "Auditorium": 2, "Ballroom": 2, "Courtroom": 2, "Museum": 2, "Palace": 2, "Temple": 1}
DH_DIST = {"Family": 12, "Francis": 8, "Horse": 4, "Lighthouse": 8, "M60": 8, "Panther": 4, "Playground": 8, "Train": 4,
           "Auditorium": 4, "Ballroom": 4, "Courtroom": 4, "Museum": 4, "Palace": 4, "Temple": 4}
DH_REL_DIFF = {"Family": 1600, "Francis": 1600, "Horse": 1300, "Lighthouse": 1600, "M60": 1600, "Panther": 1300, "Playground": 1600,
               "Train": 1600, "Auditorium": 1300, "Ballroom": 1300, "Courtroom": 1300, "Museum": 1300, "Palace": 1300, "Temple": 1500}

# test.py:217-232: photometric thresholds per Tanks&Temples scene (filter_depth_dynamic's `photo_thres`)
TANK_PHOTO_THRES = {"Family": [0.8, 0.8, 0.95], "Francis": [0.3, 0.6, 0.6], "Horse": [0.15, 0.4, 0.8], "Lighthouse": [0.3, 0.8, 0.9],
                    "M60": [0.7, 0.8, 0.95], "Panther": [0.3, 0.3, 0.95], "Playground": [0.3, 0.8, 0.9], "Train": [0.3, 0.6, 0.95],
                    "Auditorium": [0.0, 0.0, 0.0], "Ballroom": [0.3, 0.3, 0.5], "Courtroom": [0.0, 0.2, 0.2], "Museum": [0.3, 0.3, 0.7],
                    "Palace": [0.3, 0.3, 0.4], "Temple": [0.3, 0.5, 0.5]}
# test.py:240-293: ETH3D per-scene geometric thresholds (views that must agree / reprojection error in pixels)
ETH3D_GEO_MASK_THRES = {s: 1 for s in (
    "courtyard", "delivery_area", "electro", "facade", "kicker", "meadow", "office", "pipes", "playground", "relief", "relief_2",
    "terrace", "terrains", "botanical_garden", "boulders", "door", "exhibition_hall", "lecture_room", "living_room", "lounge",
    "observatory", "old_computer", "statue", "terrace_2")}
ETH3D_GEO_MASK_THRES["bridge"] = 2
ETH3D_GEO_PIXEL_THRES = {"courtyard": 0.5, "delivery_area": 0.5, "electro": 1, "facade": 1, "kicker": 1, "meadow": 2, "office": 2,
                         "pipes": 2, "playground": 1, "relief": 1, "relief_2": 1, "terrace": 0.5, "terrains": 1, "botanical_garden": 1,
                         "boulders": 0.5, "bridge": 0.5, "door": 0.5, "exhibition_hall": 0.5, "lecture_room": 0.5, "living_room": 0.5,
                         "lounge": 2, "observatory": 1, "old_computer": 2, "statue": 1, "terrace_2": 0.5}


def scene_protocol(dataset: str, scene: str, outdir: str, geo_mask_thres=2, geo_pixel_thres=1.0, geo_depth_thres=0.01,
                   photo_thres=(0.3, 0.0, 0.0)) -> dict:
    """What the reference's test.py:298-367 passes to filter_depth / filter_depth_dynamic for one scene of a dataset: the
    keyword arguments of filter_depth() below plus `plyfilename`.  dtu: the command-line thresholds, pc/mvs<id>_l3.ply;
    tank: scan = the list entry without its 'intermediate/' / 'advanced/' prefix, that scene's photometric thresholds and
    dynamic geometric rule; eth3d: that scene's geometric thresholds; general: the command-line thresholds, pc.ply."""
    kw = dict(geo_mask_thres=geo_mask_thres, geo_pixel_thres=geo_pixel_thres, geo_depth_thres=geo_depth_thres,
              photo_thres=list(photo_thres), dataset=dataset, scan=None)
    if dataset == "dtu":
        kw["plyfilename"] = os.path.join(outdir, "pc", "mvs{:0>3}_l3.ply".format(int(scene[4:])))
    elif dataset == "tank":
        scan = scene.split("/")[-1]
        if scan not in DH_VIEW_NUM:
            raise KeyError(f"no Tanks&Temples fusion parameters for scene {scan!r} (known: {sorted(DH_VIEW_NUM)})")
        kw.update(scan=scan, photo_thres=list(TANK_PHOTO_THRES[scan]), plyfilename=os.path.join(outdir, "pc", f"{scan}.ply"))
    elif dataset == "eth3d":
        if scene not in ETH3D_GEO_MASK_THRES:
            raise KeyError(f"no ETH3D fusion parameters for scene {scene!r}")
        kw.update(geo_mask_thres=ETH3D_GEO_MASK_THRES[scene], geo_pixel_thres=ETH3D_GEO_PIXEL_THRES[scene],
                  plyfilename=os.path.join(outdir, "pc", f"{scene}.ply"))
    else:
        kw["plyfilename"] = os.path.join(outdir, "pc.ply")
    return kw


def compose_mats(ref_K: np.ndarray, ref_E: np.ndarray, src_K: np.ndarray, src_E: np.ndarray) -> np.ndarray:
    """the six matrices of one (reference, source) pair in the kernel's layout, composed in fp32 like filter.py:22-45"""
    ref_K, ref_E, src_K, src_E = (np.asarray(m, np.float32) for m in (ref_K, ref_E, src_K, src_E))
    a = np.linalg.inv(ref_K)
    b = np.matmul(src_E, np.linalg.inv(ref_E))[:3, :4]
    d = np.linalg.inv(src_K)
    e = np.matmul(ref_E, np.linalg.inv(src_E))[:3, :4]
    return np.concatenate([a.reshape(-1), b.reshape(-1), src_K.reshape(-1), d.reshape(-1), e.reshape(-1), ref_K.reshape(-1)]).astype(np.float32)


def geo_consistency(ops: Ops, depth_ref: torch.Tensor, depth_src: torch.Tensor, mats: torch.Tensor, pix_thres: Sequence[float],
                    rel_thres: Sequence[float], depth_range=None):
    """depth_ref [H,W], depth_src [S,Hs,Ws], mats [S,60] on the device -> level_counts [L,H,W] int32, depth_sum [H,W]"""
    ops._chk(depth_ref, depth_src, mats)
    H, W = depth_ref.shape
    S, Hs, Ws = depth_src.shape
    L = len(pix_thres)
    pt = torch.tensor(list(pix_thres), dtype=torch.float64, device=ops.device)
    rt = torch.tensor(list(rel_thres), dtype=torch.float32, device=ops.device)
    counts = torch.empty(L, H, W, dtype=torch.int32, device=ops.device)
    dsum = ops.empty(H, W)
    rng = (0.0, 0.0) if depth_range is None else (float(depth_range[0]), float(depth_range[1]))
    ops._call("dmvs_geo_consistency_f32", _ptr(depth_ref), _ptr(depth_src), _ptr(mats), _ptr(pt), _ptr(rt), L,
              int(depth_range is not None), rng[0], rng[1], _ptr(counts), _ptr(dsum), S, H, W, Hs, Ws, ops.stream())
    return counts, dsum


def fuse_view(ops: Ops, ref_depth, ref_K, ref_E, depth_max, depth_min, confs, srcs, photo_thres, geo_mask_thres=3, geo_pixel_thres=1.0,
              geo_depth_thres=0.01, method="casdiffmvs", dynamic=None):
    """one reference view (numpy in, numpy out): -> photo_mask, geo_mask, final_mask, depth_est_averaged (fp64).
    dynamic = [view_num, dist, rel_diff] selects filter_depth_dynamic's rule."""
    dev = ops.device
    dref = torch.from_numpy(np.ascontiguousarray(ref_depth, dtype=np.float32)).to(dev)
    dsrc = torch.from_numpy(np.stack([np.ascontiguousarray(d, dtype=np.float32) for d, _, _ in srcs])).to(dev)
    mats = torch.from_numpy(np.stack([compose_mats(ref_K, ref_E, k, e) for _, k, e in srcs])).to(dev)
    cf = [torch.from_numpy(np.ascontiguousarray(c, dtype=np.float32)).to(dev) for c in confs]
    if dynamic is None:
        photo = cf[0] > photo_thres[0]
        for c, t in zip(cf[1:], photo_thres[1:]):
            photo = photo & (c > t)
        counts, dsum = geo_consistency(ops, dref, dsrc, mats, [geo_pixel_thres], [geo_depth_thres], depth_range=(depth_min, depth_max))
        n = counts[0]
        geo = n >= geo_mask_thres
        avg = (dsum + dref).double() / (n + 1).double()
        final = photo & geo
    else:
        photo = (cf[0] > photo_thres[0]) & (cf[1] > photo_thres[1]) & (cf[2] > photo_thres[2]) if method == "casdiffmvs" else \
            (cf[0] > photo_thres[0]) & (cf[1] > photo_thres[2])
        levels = list(range(dynamic[0], 11))
        counts, dsum = geo_consistency(ops, dref, dsrc, mats, [i / dynamic[1] for i in levels], [np.float64(i / dynamic[2]) for i in levels])
        n = counts[-1]
        geo = n >= 10
        for li, i in enumerate(levels):
            geo = geo | (counts[li] >= i)
        avg = (dsum + dref).double() / (n + 1).double()
        final = photo & geo & (avg >= depth_min) & (avg <= depth_max)
    return photo.cpu().numpy(), geo.cpu().numpy(), final.cpu().numpy(), avg.cpu().numpy()


def unproject(depth: np.ndarray, K: np.ndarray, E: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """valid pixels -> world points [N,3] (filter.py:192-205)"""
    h, w = depth.shape[:2]
    x, y = np.meshgrid(np.arange(0, w), np.arange(0, h))
    x, y, d = x[mask], y[mask], depth[mask]
    xyz_ref = np.matmul(np.linalg.inv(K), np.vstack((x, y, np.ones_like(x))) * d)
    return np.matmul(np.linalg.inv(E), np.vstack((xyz_ref, np.ones_like(x))))[:3].transpose((1, 0))


def filter_depth(pair_folder, out_folder, plyfilename, geo_mask_thres=3, geo_pixel_thres=1.0, geo_depth_thres=0.01,
                 photo_thres=(0.3, 0.5, 0.5), method="casdiffmvs", dataset="dtu", scan=None, device="cuda:0", ops: Ops | None = None) -> int:
    """filter.py:95-227 (dataset != 'tank') / :261-441 (dataset == 'tank': `scan` picks the dynamic parameters) over one
    scene's output tree; writes mask/*.png and the fused point cloud.  -> number of fused points."""
    from PIL import Image
    ops = ops or Ops.for_device(device)
    if os.path.dirname(plyfilename):
        os.makedirs(os.path.dirname(plyfilename), exist_ok=True)
    pairs = IO.read_pair_file(os.path.join(pair_folder, "pair.txt"), dataset)
    nconf = 3 if method == "casdiffmvs" else 2
    dynamic = [DH_VIEW_NUM[scan], DH_DIST[scan], DH_REL_DIFF[scan]] if dataset == "tank" else None
    cache = {}

    def view(v):
        if v not in cache:
            k, e, dmax, dmin = IO.read_camera_parameters(os.path.join(out_folder, f"cams/{v:0>8}_cam.txt"))
            cache[v] = (np.ascontiguousarray(IO.read_pfm(os.path.join(out_folder, f"depth_est/{v:0>8}.pfm"))[0]), k, e, dmax, dmin)
        return cache[v]

    verts, cols = [], []
    os.makedirs(os.path.join(out_folder, "mask"), exist_ok=True)
    for ref_view, src_views in pairs:
        rd, rk, re_, dmax, dmin = view(ref_view)
        img = IO.read_img(os.path.join(out_folder, f"images/{ref_view:0>8}.jpg"))[0]
        confs = [np.ascontiguousarray(IO.read_pfm(os.path.join(out_folder, f"conf{i}/{ref_view:0>8}.pfm"))[0]) for i in range(nconf)]
        srcs = [view(s)[:3] for s in src_views]
        photo, geo, final, avg = fuse_view(ops, rd, rk, re_, dmax, dmin, confs, srcs, list(photo_thres), geo_mask_thres, geo_pixel_thres,
                                           geo_depth_thres, method, dynamic)
        for name, m in (("photo", photo), ("geo", geo), ("final", final)):
            Image.fromarray(m.astype(np.uint8) * 255).save(os.path.join(out_folder, f"mask/{ref_view:0>8}_{name}.png"))
        verts.append(unproject(avg, rk, re_, final))
        cols.append((img[final] * 255).astype(np.uint8))
    xyz, rgb = np.concatenate(verts, 0), np.concatenate(cols, 0)
    IO.write_ply(plyfilename, xyz.astype(np.float32), rgb)
    return int(xyz.shape[0])
