"""Module-level drop-in parity: the sub-modules and free functions of the reference's models/module.py and
models/update.py, called with the reference's own signatures, against the inputs/outputs that forward hooks
recorded from the imported reference (tests/golden/make_golden.py).  One test per SURVEY section 8a row.
Runs on the host emulation (CPU) and, with -m gpu, on the MI355X."""
import pytest
import torch

from conftest import emu_ops, hip_ops, pin_ops
from diffmvs_amd import synth


@pytest.fixture(params=[pytest.param("emu"), pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request, monkeypatch):
    ops = emu_ops() if request.param == "emu" else hip_ops()
    if request.param == "emu":
        pin_ops(monkeypatch, ops)
    return ops


def build(variant, ops, nd=32):
    from models import CasDiffMVS
    model = CasDiffMVS(synth.make_args(variant, numdepth_initial=nd), test=True).eval()
    model.load_state_dict(synth.synth_state_dict(model.state_dict(), 123))
    return model.to(ops.device)


def close(a, b, tol=5e-5):
    a = a.detach().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    err, scale = float((a - b).abs().max()), max(1.0, float(b.abs().max()))
    assert err <= tol * scale, f"max abs err {err:.3e} (scale {scale:.3e})"


def D(ops, t):
    return None if t is None else t.to(ops.device)


@pytest.mark.parametrize("variant", ["diffmvs", "casdiffmvs"])
def test_a15_feature_and_context_nets(backend, golden, variant):
    g, model = golden(f"ops_{variant}.npz"), build(variant, backend)
    f = model.feature(D(backend, g.t("feature#0.in.0")))
    c = model.context(D(backend, g.t("context#0.in.0")))
    assert set(f) == {k.split(".")[-1] for k in g.files if k.startswith("feature#0.out.")}
    for k in f:
        close(f[k], g.t(f"feature#0.out.{k}"))
    for k in c:
        close(c[k], g.t(f"context#0.out.{k}"))


def test_a1_differentiable_warping_edge_cases(backend, golden):
    """big rotations (all taps out of bounds), points behind the camera, z == 0, source grid != hypothesis grid"""
    from models.module import differentiable_warping
    g = golden("warp_edge.npz")
    for ci in range(int(g.np("n_cases"))):
        out = differentiable_warping(D(backend, g.t(f"c{ci}.src")), D(backend, g.t(f"c{ci}.src_proj")),
                                     D(backend, g.t(f"c{ci}.ref_proj")), D(backend, g.t(f"c{ci}.depth")))
        assert torch.isfinite(out).all()
        close(out, g.t(f"c{ci}.out"), 1e-4)


def test_a3_a4_a5_initial_cost(backend, golden):
    g, model = golden("ops_diffmvs.npz"), build("diffmvs", backend)
    dn = model.depthnet
    for n in (0, 1):
        close(dn.pixel_view_weight(D(backend, g.t(f"depthnet.pixel_view_weight#{n}.in.0"))),
              g.t(f"depthnet.pixel_view_weight#{n}.out"))
    close(dn.cost_regularization(D(backend, g.t("depthnet.cost_regularization#0.in.0"))),
          g.t("depthnet.cost_regularization#0.out"))
    feats = [D(backend, f) for f in g.seq("depthnet#0.in.0")]
    mask, nd, depth, vw, conf = dn(feats, D(backend, g.t("depthnet#0.in.1")), D(backend, g.t("depthnet#0.in.2")),
                                   depth_values=D(backend, g.t("depthnet#0.kw.depth_values")))
    close(mask, g.t("depthnet#0.out.0"))
    close(nd, g.t("depthnet#0.out.1"))
    close(depth, g.t("depthnet#0.out.2"))
    close(vw, g.t("depthnet#0.out.3"))
    ref = g.t("depthnet#0.out.4")
    assert float(((conf.cpu() - ref).abs() > 1e-4).float().mean()) < 0.01      # floor(index) bin flips


@pytest.mark.parametrize("variant,calls,stage", [("diffmvs", (0, 1), 2), ("casdiffmvs", (3, 4), 3)])
def test_a6_a7_get_cost(backend, golden, variant, calls, stage):
    g, e = golden(f"ops_{variant}.npz"), golden(f"e2e_{variant}_cfg1.npz")
    model = build(variant, backend)
    meta = e.meta()
    imgs, proj, dv = synth.synth_inputs(meta["H"], meta["W"], meta["n_src"], B=meta["B"], seed=meta["scene_seed"])
    feats = [model.feature(D(backend, im))[f"stage{stage}"] for im in imgs]
    dmax, dmin = torch.tensor(935.0).view(1, 1, 1, 1), torch.tensor(425.0).view(1, 1, 1, 1)
    for n in calls:
        k = f"GetCost#{n}"
        conf = D(backend, g.t(k + ".kw.confidence")) if (k + ".kw.confidence") in g else None
        cost, samples = model.GetCost(D(backend, g.t(k + ".in.0")), features=feats,
                                      proj_matrices=D(backend, proj[f"stage{stage}"]),
                                      depth_interval=float(g.np(k + ".kw.depth_interval")), depth_max=dmax, depth_min=dmin,
                                      CostNum=model.CostNum[stage - 1], view_weights=D(backend, g.t(k + ".kw.view_weights")),
                                      confidence=conf)
        close(samples, g.t(k + ".out.1"), 1e-6)
        close(cost, g.t(k + ".out.0"), 1e-4)


@pytest.mark.parametrize("variant,i", [("diffmvs", 0), ("casdiffmvs", 1)])
def test_a8_a9_a10_update_nets(backend, golden, variant, i):
    g, model = golden(f"ops_{variant}.npz"), build(variant, backend)
    k, ub = f"update_block.{i}", build(variant, backend).update_block[i]
    close(ub.encoder(*[D(backend, g.t(f"{k}.encoder#0.in.{j}")) for j in range(3)]), g.t(f"{k}.encoder#0.out"))
    close(ub.unet.gru(D(backend, g.t(f"{k}.unet.gru#0.in.0")), D(backend, g.t(f"{k}.unet.gru#0.in.1"))),
          g.t(f"{k}.unet.gru#0.out"))
    hid, delta, conf = ub.unet(*[D(backend, g.t(f"{k}.unet#0.in.{j}")) for j in range(3)])
    close(hid, g.t(f"{k}.unet#0.out.0"))
    close(delta, g.t(f"{k}.unet#0.out.1"))
    close(conf, g.t(f"{k}.unet#0.out.2"))
    hi = model.hidden_init[i]
    x = D(backend, g.t(f"hidden_init.{i}#0.in.0"))
    for layer in hi:
        x = layer(x) if hasattr(layer, "packed") else None
        if x is None:
            break
    if x is not None:
        close(x, g.t(f"hidden_init.{i}#0.out"))


def test_a11_update_block_with_callable(backend, golden):
    """DiffusionUpdateBlockDepth.forward(depth_cost_func, ...) with the reference's closure convention: the
    recorded costs of the reference run are replayed through the callable."""
    g, model = golden("ops_diffmvs.npz"), build("diffmvs", backend)
    e = golden("e2e_diffmvs_cfg1.npz")
    ub = model.update_block[0]
    k = "update_block.0"
    ub.noise_source = lambda shape, device: e.t("noise.0").to(device)
    imgs, proj, dv = synth.synth_inputs(128, 160, 5, B=1, seed=1)
    feats = [model.feature(D(backend, im))["stage2"] for im in imgs]
    dmax, dmin = torch.tensor(935.0).view(1, 1, 1, 1), torch.tensor(425.0).view(1, 1, 1, 1)
    vw = D(backend, g.t("GetCost#0.kw.view_weights"))

    def cost_fn(inv, confidence=None):
        return model.GetCost(inv, features=feats, proj_matrices=D(backend, proj["stage2"]), depth_interval=2.0 / 384,
                             depth_max=dmax, depth_min=dmin, CostNum=6, view_weights=vw, confidence=confidence)
    mask, hidden, inv_list, conf_list = ub(cost_fn, D(backend, g.t(f"{k}#0.in.1")), D(backend, g.t(f"{k}#0.in.2")),
                                           D(backend, g.t(f"{k}#0.in.3")))
    close(mask, g.t(f"{k}#0.out.0"))
    close(hidden, g.t(f"{k}#0.out.1"), 2e-4)
    assert len(inv_list) == 4 and len(conf_list) == 4
    close(inv_list[-1], g.t(f"{k}#0.out.2.3"), 2e-4)
    close(conf_list[-1], g.t(f"{k}#0.out.3.3"), 2e-4)


def test_a13_upsample_depth(backend):
    from models.module import upsample_depth
    from oracle import diffmvs_oracle as O
    gen = torch.Generator().manual_seed(3)
    for r in (2, 4):
        depth, mask = torch.rand(2, 1, 6, 9, generator=gen), torch.randn(2, 9 * r * r, 6, 9, generator=gen)
        close(upsample_depth(D(backend, depth), D(backend, mask), r), O.upsample_depth(depth, mask, r), 1e-5)
