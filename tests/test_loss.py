"""CPU: compute_inverse_loss against values recorded from the reference's models/loss.py."""
from types import SimpleNamespace

import pytest
import torch

from models import compute_inverse_loss


@pytest.mark.parametrize("variant,iters", [("diffmvs", [1, 4, 0]), ("casdiffmvs", [1, 3, 3])])
def test_loss_matches_reference(golden, variant, iters):
    g = golden("loss.npz")
    inputs = [g.t(f"{variant}.in.{i}") for i in range(int(g.np(f"{variant}.n_in")))]
    confs = [g.t(f"{variant}.conf.{i}") for i in range(int(g.np(f"{variant}.n_conf")))]
    gt = {f"stage{s}": g.t(f"{variant}.gt.stage{s}") for s in (1, 2, 3, 4)}
    mask = {f"stage{s}": g.t(f"{variant}.mask.stage{s}") for s in (1, 2, 3, 4)}
    loss, parts = compute_inverse_loss(SimpleNamespace(conf_weight=0.05), inputs, confs, gt, mask, g.t(f"{variant}.dv"),
                                       loss_rate=0.9, iters=iters)
    assert abs(float(loss) - float(g.np(f"{variant}.loss"))) <= 1e-5 * abs(float(g.np(f"{variant}.loss")))
    for k, v in parts.items():
        assert abs(float(v) - float(g.np(f"{variant}.part.{k}"))) <= 1e-5 * max(1e-6, abs(float(g.np(f"{variant}.part.{k}"))))
    with pytest.raises(AssertionError):
        compute_inverse_loss(SimpleNamespace(conf_weight=0.05), inputs[:-1], confs, gt, mask, g.t(f"{variant}.dv"), iters=iters)
