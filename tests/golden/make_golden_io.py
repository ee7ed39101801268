#!/usr/bin/env python3
"""DEV-ONLY fixture generator for the on-disk formats.  Runs only where /root/reference is mounted.

Imports the reference's datasets/data_io.py (its unused `torchvision.transforms` import is satisfied by an empty module:
only get_transform() touches it, which is not called) and records, as plain bytes / arrays in tests/golden/io.npz:
  * the files its WRITERS produce (save_pfm grey + colour, write_cam) for seeded arrays,
  * what its READERS (read_pfm, read_camera_parameters, read_pair_file) return for those files and for a hand-written
    input camera / pair file in the dataset's layout.
Only data is written: file bytes and parsed numbers."""
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def import_data_io():
    tv = types.ModuleType("torchvision")
    tv.transforms = types.ModuleType("torchvision.transforms")
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.transforms", tv.transforms)
    spec = importlib.util.spec_from_file_location("ref_data_io", os.path.join(REF, "datasets", "data_io.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


PAIR_TXT = """4
0
3 1 2036.53 2 1980.10 3 30.5
1
3 0 2036.53 2 1800.00 3 0.05
2
2 0 1980.10 1 1800.00
3
1 3 5.0
"""

INPUT_CAM = """extrinsic
0.970263 0.00747983 0.241939 -191.02
-0.0147429 0.999493 0.0282234 3.28832
-0.241605 -0.030951 0.969881 22.5401
0.0 0.0 0.0 1.0

intrinsic
2892.33 0 823.205
0 2883.18 619.071
0 0 1

425.0 2.5 192 933.8
"""


def main():
    dio = import_data_io()
    rs = np.random.RandomState(3)
    out = {}
    with tempfile.TemporaryDirectory() as td:
        grey = (rs.rand(5, 7).astype(np.float32) * 500 + 425)
        col = rs.rand(4, 6, 3).astype(np.float32)
        for name, arr in (("grey", grey), ("colour", col)):
            p = os.path.join(td, name + ".pfm")
            dio.save_pfm(p, arr)
            out[f"pfm.{name}.array"] = arr
            out[f"pfm.{name}.bytes"] = np.frombuffer(open(p, "rb").read(), dtype=np.uint8)
            back, scale = dio.read_pfm(p)
            out[f"pfm.{name}.read"] = np.ascontiguousarray(back)
            out[f"pfm.{name}.scale"] = np.array(scale)
        cam = np.zeros((2, 4, 4), np.float32)
        cam[0] = np.eye(4)
        cam[0, :3, :4] = rs.randn(3, 4).astype(np.float32)
        cam[1, :3, :3] = np.array([[361.54, 0, 82.9], [0, 360.39, 66.38], [0, 0, 1]], np.float32)
        for tag, (dmax, dmin) in (("dtu", (np.float32(935.0), np.float32(425.0))), ("small", (np.float32(12.5), np.float32(0.75)))):
            p = os.path.join(td, f"cam_{tag}.txt")
            dio.write_cam(p, cam, dmax, dmin)
            out[f"cam.{tag}.bytes"] = np.frombuffer(open(p, "rb").read(), dtype=np.uint8)
            k, e, a, b = dio.read_camera_parameters(p)
            out[f"cam.{tag}.K"], out[f"cam.{tag}.E"], out[f"cam.{tag}.range"] = k, e, np.array([a, b], np.float64)
        out["cam.array"] = cam
        p = os.path.join(td, "pair.txt")
        open(p, "w").write(PAIR_TXT)
        for ds in ("dtu", "eth3d"):
            data = dio.read_pair_file(p, ds)
            out[f"pair.{ds}.ref"] = np.array([r for r, _ in data], np.int64)
            out[f"pair.{ds}.src"] = np.array([";".join(map(str, s)) for _, s in data])
        out["pair.text"] = np.array(PAIR_TXT)
        out["input_cam.text"] = np.array(INPUT_CAM)
    np.savez_compressed(os.path.join(HERE, "io.npz"), **out)
    print("io.npz:", sorted(out))


if __name__ == "__main__":
    main()
