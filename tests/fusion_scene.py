"""A small scene tree in the layout the reference's test.py leaves on disk (depth_est / conf{i} / cams / images + pair.txt),
built from seeded synthetic depth maps: shared by tests/golden/make_golden_fusion_tree.py (which runs the reference's
filter_depth on it) and tests/test_fusion.py (which runs this package's filter_depth on the same tree)."""
import os

import numpy as np

from diffmvs_amd import formats as IO
from diffmvs_amd import synth

H, W, V = 40, 56, 5


def build_tree(root, seed=21, noise=0.08, outliers=0.06, n_conf=3):
    from PIL import Image
    depths = np.asarray(synth.synth_view_depths(H, W, V, seed=seed), np.float32)
    cams = synth.synth_cameras(H, W, V - 1)[0]["stage4"][0]
    cams = cams.numpy() if hasattr(cams, "numpy") else np.asarray(cams)
    rs = np.random.RandomState(seed + 5)
    depths = depths + rs.normal(0, noise, depths.shape).astype(np.float32)
    bad = rs.rand(*depths.shape) < outliers
    depths[bad] = rs.uniform(300, 1100, int(bad.sum())).astype(np.float32)
    depths[:, :2, :3] = 0.0
    for d in ["depth_est", "cams", "images"] + [f"conf{i}" for i in range(n_conf)]:
        os.makedirs(os.path.join(root, d), exist_ok=True)
    for v in range(V):
        IO.save_pfm(os.path.join(root, f"depth_est/{v:08d}.pfm"), depths[v])
        cam = np.zeros((2, 4, 4), np.float32)
        cam[0], cam[1, :3, :3] = cams[v, 0], cams[v, 1, :3, :3]
        IO.write_cam(os.path.join(root, f"cams/{v:08d}_cam.txt"), cam, np.float32(935.0), np.float32(425.0))
        # smooth image content: JPEG decoding of noise differs more between libjpeg builds than of gradients
        yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        img = np.stack([(xx * 4 + v * 9) % 256, (yy * 5 + v * 17) % 256, ((xx + yy) * 3) % 256], -1).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(root, f"images/{v:08d}.jpg"), quality=95)
        for i in range(n_conf):
            IO.save_pfm(os.path.join(root, f"conf{i}/{v:08d}.pfm"), rs.uniform(0.2, 1.0, (H, W)).astype(np.float32))
    with open(os.path.join(root, "pair.txt"), "w") as f:
        f.write(f"{V}\n")
        for v in range(V):
            o = [u for u in range(V) if u != v]
            f.write(f"{v}\n{len(o)} " + " ".join(f"{u} 1.0" for u in o) + "\n")
    return root
