"""On-disk formats and the dataset sample contract: byte-for-byte against files the reference's own writers
produced and value-for-value against its readers (tests/golden/io.npz, made by make_golden_io.py), the MVSDataset / sample
dict on a scene tree written to a temp directory, and the output tree test.py writes read back."""
import os

import numpy as np
import pytest
import torch

from diffmvs_amd import formats as IO
from diffmvs_amd import synth


@pytest.fixture(scope="module")
def io_golden():
    from conftest import GOLDEN
    return np.load(os.path.join(GOLDEN, "io.npz"), allow_pickle=False)


@pytest.mark.parametrize("name", ["grey", "colour"])
def test_pfm_matches_reference_writer_and_reader(io_golden, tmp_path, name):
    arr = io_golden[f"pfm.{name}.array"]
    p = str(tmp_path / "x.pfm")
    IO.save_pfm(p, arr)
    assert open(p, "rb").read() == io_golden[f"pfm.{name}.bytes"].tobytes()          # byte-identical file
    back, scale = IO.read_pfm(p)
    assert scale == float(io_golden[f"pfm.{name}.scale"]) and np.array_equal(back, io_golden[f"pfm.{name}.read"])
    assert np.array_equal(back, arr)
    with pytest.raises(ValueError):
        IO.save_pfm(p, arr.astype(np.float64))


def test_pfm_big_endian_and_errors(tmp_path):
    arr = np.arange(12, dtype=">f4").reshape(3, 4)
    p = str(tmp_path / "be.pfm")
    IO.save_pfm(p, arr)
    back, _ = IO.read_pfm(p)
    assert np.array_equal(back.astype(np.float32), arr.astype(np.float32))
    (tmp_path / "bad.pfm").write_bytes(b"P6\n1 1\n-1\n")
    with pytest.raises(ValueError):
        IO.read_pfm(str(tmp_path / "bad.pfm"))


@pytest.mark.parametrize("tag", ["dtu", "small"])
def test_cam_files(io_golden, tmp_path, tag):
    cam = io_golden["cam.array"]
    rng = {"dtu": (np.float32(935.0), np.float32(425.0)), "small": (np.float32(12.5), np.float32(0.75))}[tag]
    p = str(tmp_path / "c_cam.txt")
    IO.write_cam(p, cam, *rng)
    assert open(p, "rb").read() == io_golden[f"cam.{tag}.bytes"].tobytes()
    k, e, dmax, dmin = IO.read_camera_parameters(p)
    assert np.array_equal(k, io_golden[f"cam.{tag}.K"]) and np.array_equal(e, io_golden[f"cam.{tag}.E"])
    assert [dmax, dmin] == io_golden[f"cam.{tag}.range"].tolist()                     # incl. the DTU range override


def test_input_cam_and_pair_files(io_golden, tmp_path):
    p = tmp_path / "00000000_cam.txt"
    p.write_text(str(io_golden["input_cam.text"]))
    k, e, dmin, dmax = IO.read_cam_file(str(p))
    assert (dmin, dmax) == (425.0, 933.8) and k[0, 0] == np.float32(2892.33) and e[0, 3] == np.float32(-191.02)
    pp = tmp_path / "pair.txt"
    pp.write_text(str(io_golden["pair.text"]))
    for ds in ("dtu", "eth3d"):
        data = IO.read_pair_file(str(pp), ds)
        assert [r for r, _ in data] == io_golden[f"pair.{ds}.ref"].tolist()
        assert [";".join(map(str, s)) for _, s in data] == io_golden[f"pair.{ds}.src"].tolist()
    # the dataset's own parse: score > 0.1 and not the reference view itself (datasets/mvs.py:64-77)
    assert IO.read_pair_file_scored(str(pp), 0.1) == [(0, [1, 2, 3]), (1, [0, 2]), (2, [0, 1])]


def _write_scene(root, scan, H, W, n_views, seed=0):
    """a scene tree in the reference's input layout, from the synthetic generator"""
    from PIL import Image
    imgs, proj, dv = synth.synth_inputs(H, W, n_views - 1, B=1, seed=seed)
    base = os.path.join(root, scan)
    os.makedirs(os.path.join(base, "images"))
    os.makedirs(os.path.join(base, "cams_1"))
    for v in range(n_views):
        Image.fromarray((imgs[v][0].permute(1, 2, 0).numpy() * 255).astype(np.uint8)).save(os.path.join(base, f"images/{v:08d}.png"))
        os.rename(os.path.join(base, f"images/{v:08d}.png"), os.path.join(base, f"images/{v:08d}.jpg"))     # lossless pixels, .jpg name
        cam = proj["stage4"][0, v].numpy()
        with open(os.path.join(base, f"cams_1/{v:08d}_cam.txt"), "w") as f:
            f.write("extrinsic\n" + "\n".join(" ".join(repr(float(x)) for x in row) for row in cam[0]) + "\n\nintrinsic\n" +
                    "\n".join(" ".join(repr(float(x)) for x in row) for row in cam[1, :3, :3]) + "\n\n425.0 2.5 192 935.0\n")
    with open(os.path.join(base, "pair.txt"), "w") as f:
        f.write(f"{n_views}\n")
        for v in range(n_views):
            others = [u for u in range(n_views) if u != v]
            f.write(f"{v}\n{len(others)} " + " ".join(f"{u} {100.0 - u}" for u in others) + "\n")
    return imgs, proj, dv


def test_dataset_sample_contract(tmp_path):
    """MVSDataset on a 'general' scene (no fixed-size resize): the dict CasDiffMVS.forward takes (datasets/mvs.py:196-210)"""
    H, W, V = 64, 96, 4
    imgs, proj, dv = _write_scene(str(tmp_path), "", H, W, V)
    os.rename(str(tmp_path / "cams_1"), str(tmp_path / "cams"))
    ds = IO.MVSDataset(str(tmp_path), n_views=3, numdepth=48, dataset="general")
    assert len(ds) == V
    s = ds[1]
    assert len(s["imgs"]) == 3 and s["imgs"][0].shape == (3, H, W) and s["imgs"][0].dtype == np.float32
    assert np.abs(s["imgs"][0] - imgs[1][0].numpy()).max() <= 1.0 / 255 + 1e-6          # ref view = view 1; 8-bit quantisation only
    assert s["filename"] == "{}/00000001{}"
    st4, st1 = s["proj_matrices"]["stage4"], s["proj_matrices"]["stage1"]
    assert st4.shape == (3, 2, 4, 4)
    assert np.allclose(st4[0], proj["stage4"][0, 1].numpy(), atol=1e-5) and np.allclose(st4[1], proj["stage4"][0, 0].numpy(), atol=1e-5)
    assert np.allclose(st1[:, 1, :2, :], st4[:, 1, :2, :] * 0.125) and np.array_equal(st1[:, 0], st4[:, 0]) and st1[0, 1, 2, 2] == 1.0
    dvs = s["depth_values"]
    assert dvs.shape == (48,) and dvs.dtype == np.float32 and abs(dvs[0] - 1 / 935.0) < 1e-9 and abs(dvs[-1] - 1 / 425.0) < 1e-9
    b = IO.collate([ds[0], ds[1]])
    assert b["imgs"][2].shape == (2, 3, H, W) and b["proj_matrices"]["stage2"].shape == (2, 3, 2, 4, 4) and b["depth_values"].shape == (2, 48)


def test_fixed_size_resize_scales_intrinsics(tmp_path):
    _write_scene(str(tmp_path), "scan9", 36, 50, 2)
    ds = IO.MVSDataset(str(tmp_path), n_views=2, numdepth=8, dataset="dtu", scan=["scan9"])
    ds.img_wh = (100, 72)                      # the DTU size is 1600x1152; keep the test small
    s = ds[0]
    assert s["imgs"][0].shape == (3, 72, 100)
    k_in = IO.read_cam_file(str(tmp_path / "scan9/cams_1/00000000_cam.txt"))[0]
    assert np.allclose(s["proj_matrices"]["stage4"][0, 1, 0, :3], k_in[0] * 2.0) and np.allclose(s["proj_matrices"]["stage4"][0, 1, 1, :3], k_in[1] * 2.0)
    assert s["filename"] == "scan9/{}/00000000{}"


def test_output_tree_round_trip(tmp_path):
    """what test.py:149-200 writes for a batch, read back with the fusion side's readers (filter.py:110-137)"""
    B, H, W = 2, 16, 24
    sample = IO.collate([IO.make_sample([np.random.rand(H, W, 3).astype(np.float32)] * 2, [np.eye(3, dtype=np.float32) * 5] * 2,
                                        [np.eye(4, dtype=np.float32)] * 2, 425.0, 935.0, 16, "scanA/{}/%08d{}" % i) for i in range(B)])
    out = {"depth": [torch.rand(B, H // 2, W // 2), torch.rand(B, H, W) * 500 + 425],
           "photometric_confidence": [torch.rand(B, H, W), torch.rand(B, H, W)], "conf": []}
    files = IO.save_outputs(str(tmp_path), sample, out)
    assert files[1].endswith("scanA/depth_est/00000001.pfm")
    d, _ = IO.read_pfm(files[1])
    assert np.array_equal(d, out["depth"][-1][1].numpy())
    c1, _ = IO.read_pfm(str(tmp_path / "scanA/conf1/00000000.pfm"))
    assert np.array_equal(c1, out["photometric_confidence"][1][0].numpy())
    k, e, dmax, dmin = IO.read_camera_parameters(str(tmp_path / "scanA/cams/00000000_cam.txt"))
    assert (dmax, dmin) == (935, 425) and k[0, 0] == 5.0
    assert os.path.exists(tmp_path / "scanA/images/00000001.jpg")


def test_depth_metrics():
    est = torch.tensor([[[500.0, 510.0], [600.0, 0.0]]])
    gt = torch.tensor([[[505.0, 500.0], [600.0, 700.0]]])
    mask = torch.tensor([[[True, True], [True, False]]])
    assert abs(float(IO.abs_depth_error(est, gt, mask)) - 5.0) < 1e-6
    assert abs(float(IO.abs_rel_error(est, gt, mask)) - (5 / 505 + 10 / 500) / 3) < 1e-7
    assert float(IO.abs_depth_error(est, gt, mask, thres=(8, 20))) == 10.0


def test_ply_writer(tmp_path):
    xyz = np.random.rand(5, 3).astype(np.float32)
    rgb = (np.random.rand(5, 3) * 255).astype(np.uint8)
    p = str(tmp_path / "m.ply")
    IO.write_ply(p, xyz, rgb)
    raw = open(p, "rb").read()
    head, body = raw.split(b"end_header\n")
    assert b"element vertex 5" in head and len(body) == 5 * 15
    v = np.frombuffer(body, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("r", "u1"), ("g", "u1"), ("b", "u1")])
    assert np.array_equal(v["y"], xyz[:, 1]) and np.array_equal(v["b"], rgb[:, 2])


def _same_sample(prefix, s, g):
    n = sum(1 for k in g.files if k.startswith(prefix + ".img"))
    assert len(s["imgs"]) == n
    for i in range(n):
        w = g[f"{prefix}.img{i}"]
        assert s["imgs"][i].dtype == w.dtype and np.array_equal(s["imgs"][i], w)
    for k in ("stage1", "stage2", "stage3", "stage4"):
        w = g[f"{prefix}.proj.{k}"]
        assert s["proj_matrices"][k].dtype == w.dtype and np.array_equal(s["proj_matrices"][k], w), k
    assert s["depth_values"].dtype == g[f"{prefix}.depth_values"].dtype and np.array_equal(s["depth_values"], g[f"{prefix}.depth_values"])
    assert s["filename"] == str(g[f"{prefix}.filename"])


def test_dataset_samples_match_the_reference_dataset(tmp_path):
    """tests/golden/dataset.npz = the sample dicts the reference's own datasets/mvs.py:MVSDataset returned on two scene trees
    (make_golden_dataset.py; cv2 absent: its only use there, cv2.resize, is called with the source size -- a copy).  This
    package's MVSDataset on the same trees must return the same dicts, array for array and bit for bit: view selection, image
    scaling, camera parsing, per-stage projection matrices, the inverse-depth hypothesis grid, the filename pattern."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "dataset.npz"))
    a = tmp_path / "a"
    _write_scene(str(a), "", 64, 96, 4, seed=2)
    os.rename(str(a / "cams_1"), str(a / "cams"))
    ds = IO.MVSDataset(str(a), n_views=3, numdepth=48, dataset="general")
    assert len(ds) == int(g["general.len"])
    _same_sample("general.1", ds[1], g)
    _same_sample("general.3", ds[3], g)
    b = tmp_path / "b"
    _write_scene(str(b), "scan9", 64, 96, 3, seed=4)
    ds = IO.MVSDataset(str(b), n_views=3, numdepth=16, dataset="dtu", scan=["scan9"])
    ds.img_wh = (96, 64)
    assert len(ds) == int(g["dtu.len"])
    _same_sample("dtu.0", ds[0], g)
