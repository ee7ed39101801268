"""On-disk formats and the dataset sample contract of the reference, restated:

  PFM depth / confidence maps      reference datasets/data_io.py:59-122  (read_pfm / save_pfm)
  *_cam.txt camera files           datasets/data_io.py:124-158 (write_cam / read_camera_parameters), datasets/mvs.py:80-93
  pair.txt view-selection files    datasets/data_io.py:172-190, datasets/mvs.py:42-78
  the sample dict a DataLoader hands to CasDiffMVS.forward      datasets/mvs.py:129-210
  binary PLY point clouds          filter.py:208-227 (through plyfile there)

so that real DTU / Tanks&Temples / COLMAP-converted scenes can be fed to the model and what the model emits can be read
by the reference's own filter.py (and vice versa).  CPU-side I/O only: nothing here is on the timed path.  Byte-level
agreement with the reference's writers / readers is pinned by tests/golden/io.npz (made by tests/golden/make_golden_io.py
with the reference's data_io imported).
"""
from __future__ import annotations

import os
import re
import sys
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------ PFM
def read_pfm(filename: str) -> Tuple[np.ndarray, float]:
    """-> (image [H,W] or [H,W,3] float32 in the file's byte order, rows top-down; scale)   data_io.py:59-92"""
    with open(filename, "rb") as f:
        header = f.readline().decode("utf-8").rstrip()
        if header not in ("PF", "Pf"):
            raise ValueError(f"{filename}: not a PFM file")
        m = re.match(r"^(\d+)\s(\d+)\s$", f.readline().decode("utf-8"))
        if not m:
            raise ValueError(f"{filename}: malformed PFM header")
        width, height = int(m.group(1)), int(m.group(2))
        scale = float(f.readline().rstrip())
        endian = "<" if scale < 0 else ">"
        data = np.frombuffer(f.read(), dtype=endian + "f4")
    shape = (height, width, 3) if header == "PF" else (height, width)
    return np.flipud(data.reshape(shape)), abs(scale)


def save_pfm(filename: str, image: np.ndarray, scale: float = 1.0) -> None:
    """data_io.py:94-122: bottom-up rows, 'Pf' / 'PF' header, negative scale = little-endian"""
    if image.dtype.name != "float32":
        raise ValueError("PFM images must be float32")
    if image.ndim == 3 and image.shape[2] == 3:
        color = True
    elif image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 1):
        color = False
    else:
        raise ValueError("PFM image must be H x W x 3, H x W x 1 or H x W")
    image = np.flipud(image)
    endian = image.dtype.byteorder
    if endian == "<" or (endian == "=" and sys.byteorder == "little"):
        scale = -scale
    with open(filename, "wb") as f:
        f.write(b"PF\n" if color else b"Pf\n")
        f.write(f"{image.shape[1]} {image.shape[0]}\n".encode("utf-8"))
        f.write(("%f\n" % scale).encode("utf-8"))
        f.write(np.ascontiguousarray(image).tobytes())


# ------------------------------------------------------------------------------------------ cameras
def write_cam(filename: str, cam: np.ndarray, depth_max, depth_min) -> None:
    """cam [2,4,4] = (extrinsic, intrinsic in [:3,:3])   data_io.py:124-141 (same text layout, str() of each number)"""
    with open(filename, "w") as f:
        f.write("extrinsic\n")
        for i in range(4):
            f.write("".join(str(cam[0][i][j]) + " " for j in range(4)) + "\n")
        f.write("\nintrinsic\n")
        for i in range(3):
            f.write("".join(str(cam[1][i][j]) + " " for j in range(3)) + "\n")
        f.write("\n" + str(depth_max) + " " + str(depth_min) + "\n")


def _cam_matrices(lines: Sequence[str]):
    extr = np.array(" ".join(lines[1:5]).split(), dtype=np.float32).reshape(4, 4)
    intr = np.array(" ".join(lines[7:10]).split(), dtype=np.float32).reshape(3, 3)
    return intr, extr


def read_camera_parameters(filename: str):
    """the fusion side's reader (data_io.py:143-158): -> intrinsics [3,3], extrinsics [4,4], depth_max, depth_min of a file
    written by write_cam (first number = depth_max), with the reference's DTU range override"""
    with open(filename) as f:
        lines = [ln.rstrip() for ln in f.readlines()]
    intr, extr = _cam_matrices(lines)
    depth_max, depth_min = float(lines[11].split()[0]), float(lines[11].split()[1])
    if depth_max > 425:
        depth_max, depth_min = 935, 425
    return intr, extr, depth_max, depth_min


def read_cam_file(filename: str):
    """the dataset side's reader (datasets/mvs.py:80-93): -> intrinsics, extrinsics, depth_min, depth_max of an INPUT camera
    file (first number = depth_min, last = depth_max)"""
    with open(filename) as f:
        lines = [ln.rstrip() for ln in f.readlines()]
    intr, extr = _cam_matrices(lines)
    depth_min, depth_max = float(lines[11].split()[0]), float(lines[11].split()[-1])
    if depth_min < 0:
        depth_min = 1.0
    return intr, extr, depth_min, depth_max


def read_pair_file(filename: str, dataset: str = "dtu", min_score: float = 0.1) -> List[Tuple[int, List[int]]]:
    """-> [(ref_view, [src views])]   data_io.py:172-190 (dtu / tank: every listed view; eth3d: score filter)"""
    data = []
    with open(filename) as f:
        n = int(f.readline())
        for _ in range(n):
            ref = int(f.readline().rstrip())
            tok = f.readline().rstrip().split()
            if dataset != "eth3d":
                src = [int(x) for x in tok[1::2]]
            else:
                ids, scores = [int(float(x)) for x in tok[1::2]], [float(x) for x in tok[2::2]]
                src = [i for i, sc in zip(ids, scores) if sc > min_score and i != ref]
            if src:
                data.append((ref, src))
    return data


def read_pair_file_scored(filename: str, min_score: float) -> List[Tuple[int, List[int]]]:
    """the dataset's own parse (datasets/mvs.py:46-77): score-filtered for every dataset kind"""
    data = []
    with open(filename) as f:
        n = int(f.readline())
        for _ in range(n):
            ref = int(f.readline().rstrip())
            tok = [float(x) for x in f.readline().rstrip().split()]
            ids, scores = [int(x) for x in tok[1::2]], tok[2::2]
            src = [i for i, sc in zip(ids, scores) if sc > min_score and i != ref]
            if src:
                data.append((ref, src))
    return data


# ------------------------------------------------------------------------------------------ sample contract
def multi_scale_projections(proj: np.ndarray) -> Dict[str, np.ndarray]:
    """proj [V,2,4,4] at full resolution -> stage1..stage4 with the intrinsics' first two rows scaled by 1/8, 1/4, 1/2, 1
    (datasets/mvs.py:170-185)"""
    out = {}
    for name, sc in (("stage1", 0.125), ("stage2", 0.25), ("stage3", 0.5), ("stage4", 1.0)):
        p = proj.copy()
        p[:, 1, :2, :] = proj[:, 1, :2, :] * sc
        out[name] = p
    return out


def make_sample(imgs: Sequence[np.ndarray], intrinsics: Sequence[np.ndarray], extrinsics: Sequence[np.ndarray], depth_min: float,
                depth_max: float, numdepth: int = 384, filename: str = "{}/00000000{}"):
    """images [H,W,3] float32 in [0,1] (index 0 = reference view) + cameras -> the dict of datasets/mvs.py:196-210:
    imgs: V arrays [3,H,W]; proj_matrices: stage1..4 -> [V,2,4,4]; depth_values: numdepth inverse depths ascending from
    1/depth_max to 1/depth_min (mvs.py:163-166)"""
    proj = np.zeros((len(imgs), 2, 4, 4), np.float32)
    for i, (k, e) in enumerate(zip(intrinsics, extrinsics)):
        proj[i, 0] = e
        proj[i, 1, :3, :3] = k
    return {"imgs": [np.ascontiguousarray(im.transpose(2, 0, 1)) for im in imgs], "proj_matrices": multi_scale_projections(proj),
            "depth_values": np.linspace(1.0 / depth_max, 1.0 / depth_min, numdepth, dtype=np.float32), "filename": filename}


def collate(samples: Sequence[dict]) -> dict:
    """what torch's default_collate makes of a list of samples (test.py:101-104): tensors with a leading batch axis"""
    V = len(samples[0]["imgs"])
    return {"imgs": [torch.from_numpy(np.stack([s["imgs"][v] for s in samples])) for v in range(V)],
            "proj_matrices": {k: torch.from_numpy(np.stack([s["proj_matrices"][k] for s in samples])) for k in samples[0]["proj_matrices"]},
            "depth_values": torch.from_numpy(np.stack([s["depth_values"] for s in samples])),
            "filename": [s["filename"] for s in samples]}


def read_img(filename: str):
    """-> [H,W,3] float32 in [0,1], original height, width   (datasets/mvs.py:95-99)"""
    from PIL import Image
    im = np.array(Image.open(filename).convert("RGB"), dtype=np.float32) / 255.0
    return im, im.shape[0], im.shape[1]


def resize_bilinear(img: np.ndarray, wh: Tuple[int, int]) -> np.ndarray:
    """cv2.resize(img, wh, INTER_LINEAR) of datasets/mvs.py:101-104: half-pixel-centre bilinear without antialiasing"""
    t = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1)))[None]
    out = F.interpolate(t, size=(wh[1], wh[0]), mode="bilinear", align_corners=False)
    return np.ascontiguousarray(out[0].numpy().transpose(1, 2, 0))


class MVSDataset:
    """The reference's evaluation dataset (datasets/mvs.py:9-210): same constructor arguments, same sample dict.
    dataset in {'dtu', 'tank', 'eth3d'}: <datapath>/<scan>/{images/%08d.jpg, cams_1/%08d_cam.txt, pair.txt}, images resized to
    a fixed size; 'general': <datapath>/{images, cams, pair.txt}, sizes rounded down to multiples of 32."""
    FIXED = {"dtu": (1600, 1152), "tank": (1920, 1056), "eth3d": (1920, 1280)}

    def __init__(self, datapath, n_views=3, numdepth=384, dataset="dtu", scan=("scan1",), max_h=4800, max_w=6400):
        self.datapath, self.n_views, self.numdepth, self.dataset = datapath, n_views, numdepth, dataset
        self.max_h, self.max_w = max_h, max_w
        self.img_wh = self.FIXED.get(dataset)
        self.cam_folder = "cams" if dataset == "general" else "cams_1"
        self.metas = []
        if dataset == "general":
            self.metas = [("", r, s) for r, s in read_pair_file_scored(os.path.join(datapath, "pair.txt"), 0.01)]
        else:
            for sc in scan:
                self.metas += [(sc, r, s) for r, s in read_pair_file_scored(os.path.join(datapath, sc, "pair.txt"), 0.1)]

    def __len__(self):
        return len(self.metas)

    def _adaptive(self, img, intr, base=32):
        h, w = img.shape[:2]
        if h > self.max_h or w > self.max_w:
            new_w, new_h = (1.0 * self.max_w / w) * w // base * base, (1.0 * self.max_h / h) * h // base * base
        else:
            new_w, new_h = 1.0 * w // base * base, 1.0 * h // base * base
        intr[0, :] *= new_w / w
        intr[1, :] *= new_h / h
        return resize_bilinear(img, (int(new_w), int(new_h))), intr

    def view_ids(self, idx) -> List[int]:
        """[reference view, source views ...] of sample idx (datasets/mvs.py:131-134)"""
        _, ref_view, src_views = self.metas[idx]
        return [ref_view] + src_views[:self.n_views - 1]

    def load_view(self, scan: str, vid: int):
        """one view as EVERY sample that contains it sees it: image [H,W,3] resized / cropped, intrinsics scaled with it, extrinsics,
        depth range (datasets/mvs.py:136-156).  Depends on the view alone -- which is what lets a scene loop load, decode and
        encode each image once (diffmvs_amd.eval: scene cache)."""
        root = os.path.join(self.datapath, scan) if self.dataset != "general" else self.datapath
        img, oh, ow = read_img(os.path.join(root, f"images/{vid:08d}.jpg"))
        k, e, d0, d1 = read_cam_file(os.path.join(root, self.cam_folder, f"{vid:08d}_cam.txt"))
        if self.dataset != "general":
            img = resize_bilinear(img, self.img_wh)
            k[0] *= self.img_wh[0] / ow
            k[1] *= self.img_wh[1] / oh
        else:
            img, k = self._adaptive(img, k)
        return img, k, e, d0, d1

    def sample_from_views(self, idx, views: dict):
        """the sample of __getitem__(idx) assembled from already loaded views {view id: load_view(...)}"""
        scan = self.metas[idx][0]
        ids = self.view_ids(idx)
        loaded = [views[v] for v in ids]
        prefix = (scan + "/") if self.dataset != "general" else ""
        return make_sample([x[0] for x in loaded], [x[1] for x in loaded], [x[2] for x in loaded], loaded[0][3], loaded[0][4], self.numdepth,
                           prefix + "{}/" + f"{ids[0]:0>8}" + "{}")

    def __getitem__(self, idx):
        scan = self.metas[idx][0]
        return self.sample_from_views(idx, {v: self.load_view(scan, v) for v in self.view_ids(idx)})


# ------------------------------------------------------------------------------------------ what test.py writes
def save_outputs(outdir: str, sample: dict, outputs: dict, write_images: bool = True) -> List[str]:
    """test.py:131-200 for one batch: <outdir>/<scan>/depth_est/%08d.pfm, conf{i}/%08d.pfm (one per stage that emits a
    confidence: 2 for DiffMVS, 3 for CasDiffMVS), cams/%08d_cam.txt (the full-resolution reference camera and the depth
    range), images/%08d.jpg.  -> the depth files written."""
    depth = outputs["depth"][-1].detach().float().cpu().numpy()
    confs = [c.detach().float().cpu().numpy() for c in outputs["photometric_confidence"]]
    cams = sample["proj_matrices"]["stage4"].cpu().numpy()
    imgs = sample["imgs"][0].cpu().numpy()
    dv = sample["depth_values"].cpu().numpy()
    written = []
    for b, pattern in enumerate(sample["filename"]):
        paths = {k: os.path.join(outdir, pattern.format(k, ext)) for k, ext in (("depth_est", ".pfm"), ("cams", "_cam.txt"), ("images", ".jpg"))}
        for p in paths.values():
            os.makedirs(os.path.dirname(p), exist_ok=True)
        save_pfm(paths["depth_est"], np.ascontiguousarray(depth[b]))
        write_cam(paths["cams"], cams[b, 0], 1.0 / dv[b, 0], 1.0 / dv[b, -1])
        if write_images:
            from PIL import Image
            Image.fromarray(np.clip(imgs[b].transpose(1, 2, 0) * 255, 0, 255).astype(np.uint8)).save(paths["images"], quality=95)
        for i, c in enumerate(confs):
            p = os.path.join(outdir, pattern.format(f"conf{i}", ".pfm"))
            os.makedirs(os.path.dirname(p), exist_ok=True)
            save_pfm(p, np.ascontiguousarray(c[b]))
        written.append(paths["depth_est"])
    return written


# ------------------------------------------------------------------------------------------ metrics
def abs_depth_error(depth_est: torch.Tensor, depth_gt: torch.Tensor, mask: torch.Tensor, thres=None) -> torch.Tensor:
    """mean |est - gt| over the mask, averaged over the batch items (utils.py:150-187 AbsDepthError_metrics)"""
    vals = []
    for e, g, m in zip(depth_est, depth_gt, mask):
        err = (e[m] - g[m]).abs()
        if thres is not None:
            err = err[(err >= float(thres[0])) & (err <= float(thres[1]))]
            if err.numel() == 0:
                vals.append(torch.zeros((), dtype=e.dtype, device=e.device))
                continue
        vals.append(err.mean())
    return torch.stack(vals).mean()


def abs_rel_error(depth_est: torch.Tensor, depth_gt: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """mean |est - gt| / gt over the mask, per batch item then averaged: BASELINE.json's "DTU abs-rel" (the
    reference itself only defines the absolute error)"""
    return torch.stack([((e[m] - g[m]).abs() / g[m]).mean() for e, g, m in zip(depth_est, depth_gt, mask)]).mean()


# ------------------------------------------------------------------------------------------ PLY
def write_ply(filename: str, xyz: np.ndarray, rgb: np.ndarray) -> None:
    """binary little-endian PLY with x y z (float32) red green blue (uint8) per vertex: what plyfile writes for
    filter.py:208-227"""
    n = xyz.shape[0]
    v = np.empty(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    v["x"], v["y"], v["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    v["red"], v["green"], v["blue"] = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    with open(filename, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                 "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % n).encode("ascii"))
        f.write(v.tobytes())
