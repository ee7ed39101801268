"""GPU: the evaluation driver end to end on a scene tree in the reference's input layout (datasets/mvs.py 'general'):
dataset -> CasDiffMVS on the MI355X -> depth_est / conf / cams / images tree (test.py:149-200) -> GPU consistency filter ->
masks + PLY, with the depth errors computed against depth_gt PFMs."""
import json
import os

import numpy as np
import pytest

from diffmvs_amd import formats as IO
from diffmvs_amd import synth

pytestmark = pytest.mark.gpu


def test_eval_driver_and_fusion(tmp_path, capsys):
    from PIL import Image
    from diffmvs_amd import eval as EV
    H, W, V = 64, 96, 4
    imgs, proj, dv = synth.synth_inputs(H, W, V - 1, B=1, seed=4)
    depths = synth.synth_view_depths(H, W, V, seed=4)
    root = tmp_path / "scene"
    for d in ("images", "cams", "depth_gt"):
        os.makedirs(root / d)
    for v in range(V):
        Image.fromarray((imgs[v][0].permute(1, 2, 0).numpy() * 255).astype(np.uint8)).save(str(root / f"images/{v:08d}.png"))
        os.rename(root / f"images/{v:08d}.png", root / f"images/{v:08d}.jpg")
        cam = proj["stage4"][0, v].numpy()
        with open(root / f"cams/{v:08d}_cam.txt", "w") as f:
            f.write("extrinsic\n" + "\n".join(" ".join(repr(float(x)) for x in row) for row in cam[0]) + "\n\nintrinsic\n" +
                    "\n".join(" ".join(repr(float(x)) for x in row) for row in cam[1, :3, :3]) + "\n\n425.0 2.5 192 935.0\n")
        IO.save_pfm(str(root / f"depth_gt/{v:08d}.pfm"), depths[v])
    with open(root / "pair.txt", "w") as f:
        f.write(f"{V}\n")
        for v in range(V):
            o = [u for u in range(V) if u != v]
            f.write(f"{v}\n{len(o)} " + " ".join(f"{u} {10.0 - u}" for u in o) + "\n")
    out = tmp_path / "out"
    res = EV.main(["--testpath", str(root), "--dataset", "general", "--outdir", str(out), "--method", "diffmvs", "--num_view", "4",
                   "--numdepth_initial", "16", "--batch_size", "2", "--filter"])
    assert res["views"] == 2 and res["avg_time_s"] > 0
    for v in range(V):
        d, _ = IO.read_pfm(str(out / f"depth_est/{v:08d}.pfm"))
        assert d.shape == (H, W) and np.isfinite(d).all() and d.min() >= 424.9 and d.max() <= 935.1
        assert os.path.exists(out / f"conf1/{v:08d}.pfm") and os.path.exists(out / f"cams/{v:08d}_cam.txt")
    assert "" in res["errors"] and res["errors"][""]["views"] == V and res["errors"][""]["abs_rel"] > 0
    assert os.path.exists(out / "mask/00000000_final.png") and os.path.exists(out / "scene.ply")
    json.dumps(res)
