"""One training step (forward in train mode + compute_inverse_loss + backward) against the reference's own step on
the same weights, scene, t and noise draws (tests/golden/train_<variant>.npz, made by make_golden_train.py).
CPU: through the host emulation of the kernel sources; -m gpu: the product library on the MI355X."""
import json

import numpy as np
import pytest
import torch

from conftest import rel_l1
from diffmvs_amd import synth


def _t_source():
    calls = []

    def draw(B, T, device):
        rs = np.random.RandomState(1234 + len(calls))
        v = torch.from_numpy(rs.randint(0, T, size=(B,)).astype(np.int64))
        calls.append(v)
        return v.to(device)
    return draw


def _run_step(variant, g, ops):
    from models import CasDiffMVS, compute_inverse_loss
    meta = g.meta()
    args = synth.make_args(variant, numdepth_initial=meta["nd_init"])
    model = CasDiffMVS(args, test=False)
    model.load_state_dict(synth.synth_state_dict(model.state_dict(), meta["weight_seed"]), strict=True)
    model.to(ops.device).train()
    model.noise_source = synth.NoiseSource(meta["noise_seed"])
    model.t_source = _t_source()
    imgs, proj, dv, gt, mask = synth.synth_inputs(meta["H"], meta["W"], meta["S"], B=meta["B"], seed=meta["scene_seed"], with_gt=True)
    dev = ops.device
    gt = {k: v.to(dev) for k, v in gt.items()}
    mask = {k: v.to(dev) for k, v in mask.items()}
    out = model([i.to(dev) for i in imgs], {k: v.to(dev) for k, v in proj.items()}, dv.to(dev), gt)
    loss, _ = compute_inverse_loss(args, out["depth"], out["conf"], gt, mask, dv.to(dev), loss_rate=0.9, iters=args.stage_iters)
    loss.backward()
    return model, out, loss


@pytest.mark.parametrize("variant", ["diffmvs", "casdiffmvs"])
def test_training_step_matches_reference(golden, ops, variant):
    g = golden(f"train_{variant}.npz")
    model, out, loss = _run_step(variant, g, ops)
    assert len(out["depth"]) == int(g.np("n_depth")) and len(out["conf"]) == int(g.np("n_conf"))
    for i, d in enumerate(out["depth"]):
        assert rel_l1(d.detach().cpu(), g.t(f"depth.{i}")) < 1e-4, ("depth", i)
    for i, c in enumerate(out["conf"]):
        assert rel_l1(c.detach().cpu(), g.t(f"conf.{i}")) < 1e-4, ("conf", i)
    assert abs(float(loss.detach()) - float(g.np("loss"))) < 1e-4 * abs(float(g.np("loss")))
    # gradients: per-parameter norms, per-subsystem norms, a few full tensors
    named = dict(model.named_parameters())
    bad = []
    for k in g.files:
        if k.startswith("gnorm."):
            name, want = k[6:], float(g.np(k))
            p = named[name]
            assert p.grad is not None, name
            got = float(p.grad.double().norm())
            # per-tensor: sums with heavy cancellation (BN biases at full resolution) carry fp32 ordering noise of a few 1e-3
            if abs(got - want) > 1e-2 * want + 1e-7:
                bad.append((name, got, want))
    assert not bad, bad[:8]
    pre = {}
    seen_p = set()
    for name, p in model.named_parameters():
        if p.grad is not None and id(p) not in seen_p:
            seen_p.add(id(p))
            pre[name.split(".")[0]] = pre.get(name.split(".")[0], 0.0) + float(p.grad.double().pow(2).sum())
    for k, want in json.loads(str(g.np("prefix_norms"))).items():
        assert abs(pre[k] ** 0.5 - want) < 2e-3 * want, (k, pre[k] ** 0.5, want)
    for k in g.files:
        if k.startswith("grad."):
            got, want = named[k[5:]].grad.cpu(), g.t(k)
            assert rel_l1(got, want) < 2e-3, (k, rel_l1(got, want))
    # every parameter the reference trains gets a gradient here, and nothing else does
    want_names = {k[6:] for k in g.files if k.startswith("gnorm.")}
    seen, got_names = set(), set()
    for name, p in model.named_parameters():
        if p.grad is not None and id(p) not in seen:
            seen.add(id(p))
            got_names.add(name)
    assert got_names == want_names
    # BatchNorm running statistics (per-view FeatureNet calls, momentum 0.1)
    sd = model.state_dict()
    for k in g.files:
        if k.startswith("buf."):
            a, b = sd[k[4:]].cpu().double(), g.t(k).double()
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-6), k
