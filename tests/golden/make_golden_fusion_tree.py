#!/usr/bin/env python3
"""DEV-ONLY fixture generator: the reference's filter_depth / filter_depth_dynamic (filter.py:90-227, :262-440) run on the
scene tree of tests/fusion_scene.py.  Runs only where /root/reference is mounted.

cv2 and plyfile are absent from the image: filter.py is imported with cv2.remap replaced by oracle/fusion_oracle.py:remap_linear
(see make_golden_fusion.py) and with a plyfile stand-in that CAPTURES the vertex table instead of writing it.  Everything
else -- file readers, photometric masks, consistency tests, depth averaging, unprojection, colours, the order of the
points -- is the reference's own code.  Only data is written: the fused vertex tables (tests/golden/fusion_tree.npz)."""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
from make_golden_fusion import import_filter  # noqa: E402
import fusion_scene  # noqa: E402

CAPTURED = []


class _El:
    @staticmethod
    def describe(vertex_all, name):
        CAPTURED.append(np.array(vertex_all))
        return None


class _Ply:
    def __init__(self, els):
        pass

    def write(self, fn):
        pass


def table(v):
    return np.stack([v["x"], v["y"], v["z"]], 1).astype(np.float32), np.stack([v["red"], v["green"], v["blue"]], 1).astype(np.uint8)


def main():
    ref = import_filter()
    ref.PlyElement, ref.PlyData = _El, _Ply
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        root = fusion_scene.build_tree(os.path.join(tmp, "scan"))
        cases = {"cas": dict(method="casdiffmvs", geo_mask_thres=2, photo_thres=[0.3, 0.4, 0.5]),
                 "diff": dict(method="diffmvs", geo_mask_thres=3, photo_thres=[0.35, 0.45, 0.5])}
        for tag, kw in cases.items():
            ref.filter_depth(root, root, os.path.join(tmp, "x.ply"), geo_pixel_thres=1.0, geo_depth_thres=0.01, dataset="dtu", **kw)
            out[f"{tag}_xyz"], out[f"{tag}_rgb"] = table(CAPTURED.pop())
        ref.filter_depth_dynamic("Horse", root, root, os.path.join(tmp, "x.ply"), photo_thres=[0.3, 0.4, 0.5], method="casdiffmvs", dataset="tank")
        out["dyn_xyz"], out["dyn_rgb"] = table(CAPTURED.pop())
    np.savez_compressed(os.path.join(HERE, "fusion_tree.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
