"""Depth-map fusion: dmvs_geo_consistency_f32 + diffmvs_amd/fusion.py against the NumPy restatement of the
reference's filter.py (oracle/fusion_oracle.py) on synthetic multi-view depth maps; the whole eval -> fuse chain on a scene
tree in the reference's on-disk layout."""
import os

import numpy as np
import pytest
import torch

from diffmvs_amd import formats as IO
from diffmvs_amd import fusion, synth
from oracle import fusion_oracle as FO


def _scene(H, W, V, seed=0, noise=0.6, outliers=0.08):
    """per-view depth maps of a slanted plane + noise + gross outliers + a few invalid pixels, cameras, confidences"""
    _, proj, dv = synth.synth_inputs(H, W, V - 1, B=1, seed=seed)
    depths = synth.synth_view_depths(H, W, V, seed=seed)
    rs = np.random.RandomState(seed + 5)
    depths = depths + rs.normal(0, noise, depths.shape).astype(np.float32)
    bad = rs.rand(*depths.shape) < outliers
    depths[bad] = rs.uniform(300, 1100, int(bad.sum())).astype(np.float32)
    depths[:, :2, :3] = 0.0
    cams = proj["stage4"][0].numpy()
    Ks, Es = [cams[v, 1, :3, :3].copy() for v in range(V)], [cams[v, 0].copy() for v in range(V)]
    confs = [rs.rand(H, W).astype(np.float32) for _ in range(3)]
    return depths, Ks, Es, confs


def test_remap_restatement_basics():
    """the cv2.remap restatement: exact at integer coordinates, 1/32-pixel quantisation, constant-0 border per tap"""
    src = np.arange(20, dtype=np.float32).reshape(4, 5)
    xs, ys = np.meshgrid(np.arange(5, dtype=np.float32), np.arange(4, dtype=np.float32))
    assert np.array_equal(FO.remap_linear(src, xs, ys), src)
    got = FO.remap_linear(src, np.array([[1.5, 1.51, -0.5, 4.5, np.nan]], np.float32), np.array([[2.0, 2.0, 0.0, 3.0, 1.0]], np.float32))
    assert got[0, 0] == 11.5 and got[0, 1] == 11.5                  # 1.51 rounds to 48/32 = 1.5
    assert got[0, 2] == 0.0 * 0.5 + 0.0 * 0.5 and got[0, 3] == 19 * 0.5 and got[0, 4] == 0.0      # border taps contribute 0


@pytest.mark.parametrize("H,W,V", [(24, 40, 4), (37, 29, 6)])
def test_geo_consistency_kernel_matches_oracle(ops, H, W, V):
    depths, Ks, Es, confs = _scene(H, W, V, seed=H)
    dev = ops.device
    mats = torch.from_numpy(np.stack([fusion.compose_mats(Ks[0], Es[0], Ks[v], Es[v]) for v in range(1, V)])).to(dev)
    dref = torch.from_numpy(depths[0]).to(dev)
    dsrc = torch.from_numpy(np.ascontiguousarray(depths[1:])).to(dev)
    counts, dsum = fusion.geo_consistency(ops, dref, dsrc, mats, [1.0], [0.01], depth_range=(425.0, 935.0))
    want_n, want_sum = 0, 0
    for v in range(1, V):
        m, dr, _, _ = FO.check_geometric_consistency(depths[0], Ks[0], Es[0], depths[v], Ks[v], Es[v], 935.0, 425.0, 1.0, 0.01)
        want_n = want_n + m.astype(np.int32)
        want_sum = want_sum + dr
    assert 0.2 < float((want_n >= 2).mean()) < 0.98                   # the test scene exercises both outcomes
    assert np.array_equal(counts[0].cpu().numpy(), want_n)
    assert np.allclose(dsum.cpu().numpy(), want_sum, rtol=1e-6, atol=1e-4)
    # dynamic (Tanks&Temples) levels
    dh = [3, 4, 1300]
    levels = list(range(dh[0], 11))
    counts, dsum = fusion.geo_consistency(ops, dref, dsrc, mats, [i / dh[1] for i in levels], [np.float64(i / dh[2]) for i in levels])
    sums = None
    for v in range(1, V):
        masks, _, _, _, _ = FO.check_geometric_consistency_dynamic(depths[0], Ks[0], Es[0], depths[v], Ks[v], Es[v], dh)
        sums = [m.astype(np.int32) for m in masks] if sums is None else [s + m.astype(np.int32) for s, m in zip(sums, masks)]
    for li in range(len(levels)):
        assert np.array_equal(counts[li].cpu().numpy(), sums[li]), li


@pytest.mark.parametrize("method,dynamic", [("casdiffmvs", None), ("diffmvs", None), ("casdiffmvs", [2, 4, 1300])])
def test_fuse_view_matches_oracle(ops, method, dynamic):
    H, W, V = 32, 48, 5
    depths, Ks, Es, confs = _scene(H, W, V, seed=3)
    srcs = [(depths[v], Ks[v], Es[v]) for v in range(1, V)]
    thr = [0.3, 0.5, 0.5]
    got = fusion.fuse_view(ops, depths[0], Ks[0], Es[0], 935.0, 425.0, confs[:3 if method == "casdiffmvs" else 2], srcs, thr,
                           geo_mask_thres=2, method=method, dynamic=dynamic)
    if dynamic is None:
        want = FO.fuse_view(depths[0], Ks[0], Es[0], 935.0, 425.0, confs, srcs, thr, geo_mask_thres=2, method=method)
    else:
        want = FO.fuse_view_dynamic(depths[0], Ks[0], Es[0], 935.0, 425.0, confs, srcs, thr, dynamic, method=method)
    for g, w in zip(got[:3], want[:3]):
        assert np.array_equal(g, w)
    assert 0.02 < float(want[2].mean()) < 0.9
    assert np.allclose(got[3], want[3], rtol=1e-6, atol=1e-4)
    pts = fusion.unproject(got[3], Ks[0], Es[0], got[2])
    assert np.allclose(pts, FO.unproject(want[3], Ks[0], Es[0], want[2]), rtol=1e-6, atol=1e-4)


def test_filter_depth_on_a_scene_tree(ops, tmp_path):
    """the on-disk chain: a tree in test.py's output layout -> filter_depth -> masks + PLY; the fused points lie on the
    scene plane"""
    from PIL import Image
    H, W, V = 32, 48, 5
    depths, Ks, Es, _ = _scene(H, W, V, seed=11, noise=0.05, outliers=0.05)
    out = tmp_path / "scan1"
    for d in ("depth_est", "cams", "images", "conf0", "conf1"):
        os.makedirs(out / d)
    rs = np.random.RandomState(0)
    for v in range(V):
        IO.save_pfm(str(out / f"depth_est/{v:08d}.pfm"), depths[v])
        cam = np.zeros((2, 4, 4), np.float32)
        cam[0], cam[1, :3, :3] = Es[v], Ks[v]
        IO.write_cam(str(out / f"cams/{v:08d}_cam.txt"), cam, np.float32(935.0), np.float32(425.0))
        Image.fromarray((rs.rand(H, W, 3) * 255).astype(np.uint8)).save(str(out / f"images/{v:08d}.jpg"))
        for i in range(2):
            IO.save_pfm(str(out / f"conf{i}/{v:08d}.pfm"), np.full((H, W), 0.9, np.float32))
    with open(out / "pair.txt", "w") as f:
        f.write(f"{V}\n")
        for v in range(V):
            o = [u for u in range(V) if u != v]
            f.write(f"{v}\n{len(o)} " + " ".join(f"{u} 1.0" for u in o) + "\n")
    n = fusion.filter_depth(str(out), str(out), str(tmp_path / "scan1.ply"), geo_mask_thres=3, method="diffmvs", dataset="dtu", ops=ops)
    assert n > 0.5 * V * H * W
    assert os.path.exists(out / "mask/00000002_final.png")
    raw = open(tmp_path / "scan1.ply", "rb").read().split(b"end_header\n")[1]
    pts = np.frombuffer(raw, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("r", "u1"), ("g", "u1"), ("b", "u1")])
    rsn = np.random.RandomState(1000003 * 11 + 17)
    d0 = rsn.uniform(560.0, 760.0)
    a, c = rsn.uniform(-0.25, 0.25, 2)
    resid = pts["z"] - a * pts["x"] - c * pts["y"] - d0                 # the plane of synth_inputs(seed=11)
    assert float(np.abs(resid).mean()) < 0.5


def test_scene_protocol_tables():
    """the per-dataset fusion protocol of test.py:298-367: thresholds and point-cloud names"""
    kw = fusion.scene_protocol("dtu", "scan114", "/o", 2, 0.125, 0.01, [0.3, 0.5, 0.5])
    assert kw["plyfilename"] == "/o/pc/mvs114_l3.ply" and kw["geo_pixel_thres"] == 0.125 and kw["photo_thres"] == [0.3, 0.5, 0.5]
    kw = fusion.scene_protocol("tank", "advanced/Temple", "/o")
    assert kw["scan"] == "Temple" and kw["photo_thres"] == [0.3, 0.5, 0.5] and kw["plyfilename"] == "/o/pc/Temple.ply"
    assert (fusion.DH_VIEW_NUM["Temple"], fusion.DH_DIST["Temple"], fusion.DH_REL_DIFF["Temple"]) == (1, 4, 1500)
    kw = fusion.scene_protocol("eth3d", "bridge", "/o")
    assert kw["geo_mask_thres"] == 2 and kw["geo_pixel_thres"] == 0.5 and kw["photo_thres"] == [0.3, 0.0, 0.0]
    assert fusion.scene_protocol("eth3d", "meadow", "/o")["geo_pixel_thres"] == 2
    assert fusion.scene_protocol("general", "", "/o")["plyfilename"] == "/o/pc.ply"
    assert set(fusion.TANK_PHOTO_THRES) == set(fusion.DH_VIEW_NUM) and set(fusion.ETH3D_GEO_MASK_THRES) == set(fusion.ETH3D_GEO_PIXEL_THRES)
    with pytest.raises(KeyError):
        fusion.scene_protocol("tank", "intermediate/Nowhere", "/o")


def _eq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b, equal_nan=(a.dtype.kind == "f"))


def test_fusion_oracle_matches_reference_filter_fixtures():
    """tests/golden/fusion.npz = what the reference's OWN filter.py functions returned (make_golden_fusion.py: filter.py
    imported with cv2.remap replaced by oracle/fusion_oracle.py:remap_linear, cv2 being absent from the image) -- pins the
    fp64 projection chain, the distance / relative-depth tests and the static + dynamic masks of the restatement against the
    reference's code, array for array, bit for bit.  (The remap step itself stays unpinned.)"""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "fusion.npz"))
    depths, K, E = g["depths"], g["K"], g["E"]
    dmin, dmax = float(g["range"][0]), float(g["range"][1])
    for s in range(1, depths.shape[0]):
        r = FO.reproject_with_depth(depths[0], K[0], E[0], depths[s], K[s], E[s])
        for name, arr in zip(("depth_reproj", "x_reproj", "y_reproj", "x_src", "y_src"), r):
            assert _eq(arr, g[f"reproj{s}_{name}"]), (s, name)
        m, d, xs, ys = FO.check_geometric_consistency(depths[0], K[0], E[0], depths[s], K[s], E[s], dmax, dmin, 0.75, 0.008)
        assert _eq(m, g[f"static{s}_mask"]) and _eq(d, g[f"static{s}_depth"]) and _eq(xs, g[f"static{s}_x"]) and _eq(ys, g[f"static{s}_y"])
        assert 0.02 < m.mean() < 0.98                      # the fixture separates passing from failing pixels
        for tag, dh in (("a", (2, 4.0, 1300.0)), ("b", (4, 8.0, 1600.0))):
            masks, mask, d, xs, ys = FO.check_geometric_consistency_dynamic(depths[0], K[0], E[0], depths[s], K[s], E[s], dh)
            assert _eq(np.stack(masks), g[f"dyn{tag}{s}_masks"]) and _eq(d, g[f"dyn{tag}{s}_depth"])


@pytest.mark.parametrize("tag,kw", [("cas", dict(method="casdiffmvs", geo_mask_thres=2, photo_thres=[0.3, 0.4, 0.5], dataset="dtu")),
                                    ("diff", dict(method="diffmvs", geo_mask_thres=3, photo_thres=[0.35, 0.45, 0.5], dataset="dtu")),
                                    ("dyn", dict(method="casdiffmvs", photo_thres=[0.3, 0.4, 0.5], dataset="tank", scan="Horse"))])
def test_filter_depth_matches_reference_on_a_scene_tree(ops, tag, kw, tmp_path):
    """tests/golden/fusion_tree.npz = the vertex tables the reference's own filter_depth / filter_depth_dynamic produced on the
    tree of tests/fusion_scene.py (make_golden_fusion_tree.py; cv2.remap replaced by the restatement, cv2 being absent).  This
    package's filter_depth on the same tree -- file readers, photometric masks, the consistency kernel, depth averaging,
    unprojection, colours, point order -- must give the same points, through the host emulation AND on the MI355X (the
    kernel's fp64 chain is written without contractions, so its threshold decisions are NumPy's)."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    import fusion_scene
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "fusion_tree.npz"))
    root = fusion_scene.build_tree(str(tmp_path / "scan"))
    ply = str(tmp_path / "out.ply")
    n = fusion.filter_depth(root, root, ply, ops=ops, **kw)
    raw = open(ply, "rb").read().split(b"end_header\n")[1]
    pts = np.frombuffer(raw, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("r", "u1"), ("g", "u1"), ("b", "u1")])
    want_xyz, want_rgb = g[f"{tag}_xyz"], g[f"{tag}_rgb"]
    assert n == want_xyz.shape[0] == pts.shape[0]
    got_xyz = np.stack([pts["x"], pts["y"], pts["z"]], 1)
    assert np.allclose(got_xyz, want_xyz, rtol=1e-6, atol=2e-4)
    assert np.array_equal(np.stack([pts["r"], pts["g"], pts["b"]], 1), want_rgb)
