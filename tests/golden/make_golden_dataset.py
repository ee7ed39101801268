#!/usr/bin/env python3
"""DEV-ONLY fixture generator for the dataset sample contract.  Runs only where /root/reference is mounted.

Runs the reference's own datasets/mvs.py:MVSDataset on two small scene trees (tests/test_formats.py:_write_scene) and
records the sample dicts it returns.  cv2 is absent from the image; mvs.py only uses it for cv2.resize (and two thread
settings).  The stand-in's resize REFUSES to change the size (OpenCV's resize is a copy when dsize equals the source size),
so the fixtures cover what MVSDataset computes when no resampling is needed -- view selection from pair.txt, image loading
and scaling to [0,1], camera parsing, per-stage projection matrices, intrinsics scaling bookkeeping, the inverse-depth
hypothesis grid, the filename pattern; the resampling itself stays unpinned.  Only data is written."""
import importlib
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def import_ref_mvs():
    cv2 = types.ModuleType("cv2")
    cv2.INTER_LINEAR = 1

    def resize(img, dsize, interpolation=1):
        assert (img.shape[1], img.shape[0]) == tuple(int(d) for d in dsize), "the stand-in only supports the identity resize"
        return img.copy()
    cv2.resize = resize
    cv2.setNumThreads = lambda n: None
    cv2.ocl = types.SimpleNamespace(setUseOpenCL=lambda b: None)
    tv = types.ModuleType("torchvision")
    tv.transforms = types.ModuleType("torchvision.transforms")
    for name, mod in (("cv2", cv2), ("torchvision", tv), ("torchvision.transforms", tv.transforms)):
        sys.modules.setdefault(name, mod)
    sys.path.insert(0, REF)
    return importlib.import_module("datasets.mvs")


def flat(prefix, s, out):
    for i, im in enumerate(s["imgs"]):
        out[f"{prefix}.img{i}"] = np.asarray(im)
    for k, v in s["proj_matrices"].items():
        out[f"{prefix}.proj.{k}"] = np.asarray(v)
    out[f"{prefix}.depth_values"] = np.asarray(s["depth_values"])
    out[f"{prefix}.filename"] = np.array(s["filename"])


def main():
    import test_formats as TF
    ref = import_ref_mvs()
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        TF._write_scene(tmp, "", 64, 96, 4, seed=2)
        os.rename(os.path.join(tmp, "cams_1"), os.path.join(tmp, "cams"))
        ds = ref.MVSDataset(tmp, n_views=3, numdepth=48, dataset="general")
        out["general.len"] = np.array(len(ds))
        flat("general.1", ds[1], out)
        flat("general.3", ds[3], out)
    with tempfile.TemporaryDirectory() as tmp:
        TF._write_scene(tmp, "scan9", 64, 96, 3, seed=4)
        ds = ref.MVSDataset(tmp, n_views=3, numdepth=16, dataset="dtu", scan=["scan9"])
        ds.img_wh = (96, 64)                     # the DTU size is 1600x1152; the tree's own size = identity resize
        out["dtu.len"] = np.array(len(ds))
        flat("dtu.0", ds[0], out)
    np.savez_compressed(os.path.join(HERE, "dataset.npz"), **out)
    print({k: getattr(v, "shape", v) for k, v in list(out.items())[:12]})


if __name__ == "__main__":
    main()
