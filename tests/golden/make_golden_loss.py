#!/usr/bin/env python3
"""DEV-ONLY (needs /root/reference): records compute_inverse_loss (reference models/loss.py:6-73) on seeded
random inputs for both variants -> tests/golden/loss.npz.  Only data is written."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from make_golden import import_reference  # noqa: E402


def case(rs, iters, B=2, H=32, W=48):
    cas = iters[2] != 0
    stage_id = [1] * iters[0] + [2] * (iters[1] + 1) + ([3] * (iters[2] + 1) if cas else []) + [4]
    res = {1: (H // 8, W // 8), 2: (H // 4, W // 4), 3: (H // 2, W // 2), 4: (H, W)}
    inputs = [torch.from_numpy(rs.uniform(430, 930, (B,) + res[s]).astype(np.float32)) for s in stage_id]
    n_conf = iters[1] + (iters[2] if cas else 0)
    conf_res = [res[2]] * iters[1] + ([res[3]] * iters[2] if cas else [])
    confs = [torch.from_numpy(rs.uniform(0.0, 1.0, (B,) + r).astype(np.float32)) for r in conf_res]
    gt, mask = {}, {}
    for s in (1, 2, 3, 4):
        g = rs.uniform(430, 930, (B,) + res[s]).astype(np.float32)
        g[rs.uniform(size=g.shape) < 0.1] = 0.0                 # invalid pixels (depth 0)
        gt[f"stage{s}"] = torch.from_numpy(g)
        mask[f"stage{s}"] = torch.from_numpy((rs.uniform(size=g.shape) < 0.8).astype(np.float32))
    dv = torch.from_numpy(np.tile(np.linspace(1 / 935.0, 1 / 425.0, 384, dtype=np.float32)[None], (B, 1)))
    return inputs, confs, gt, mask, dv, n_conf


def main():
    ref_models, _, _ = import_reference()
    rs = np.random.RandomState(5)
    out = {}
    for name, iters in (("diffmvs", [1, 4, 0]), ("casdiffmvs", [1, 3, 3])):
        inputs, confs, gt, mask, dv, _ = case(rs, iters)
        args = SimpleNamespace(conf_weight=0.05)
        loss, parts = ref_models.compute_inverse_loss(args, inputs, confs, gt, mask, dv, loss_rate=0.9, iters=iters)
        out[f"{name}.loss"] = np.array(float(loss))
        for k, v in parts.items():
            out[f"{name}.part.{k}"] = np.array(float(v))
        out[f"{name}.n_in"] = np.array(len(inputs))
        for i, t in enumerate(inputs):
            out[f"{name}.in.{i}"] = t.numpy()
        out[f"{name}.n_conf"] = np.array(len(confs))
        for i, t in enumerate(confs):
            out[f"{name}.conf.{i}"] = t.numpy()
        for s in (1, 2, 3, 4):
            out[f"{name}.gt.stage{s}"] = gt[f"stage{s}"].numpy()
            out[f"{name}.mask.stage{s}"] = mask[f"stage{s}"].numpy()
        out[f"{name}.dv"] = dv.numpy()
        print(name, float(loss))
    np.savez_compressed(os.path.join(HERE, "loss.npz"), **out)


if __name__ == "__main__":
    main()
