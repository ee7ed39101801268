"""CPU: the oracle (oracle/diffmvs_oracle.py) against vectors recorded from the imported
reference (tests/golden/make_golden.py).  This is what "parity pinned" rests on."""
import numpy as np
import pytest
import torch

from conftest import conf_close, rel_l1, state_keys
from diffmvs_amd import synth
from oracle import diffmvs_oracle as O


def make_sd(variant, seed=123):
    keys = state_keys(variant)["keys"]
    dt = {"float32": torch.float32, "int64": torch.int64}
    tmpl = {k: torch.zeros(shape, dtype=dt[d]) for k, (shape, d) in keys.items()}
    sched = O.cosine_schedule(1000)
    for k in tmpl:
        leaf = k.rsplit(".", 1)[-1]
        if leaf in sched:
            tmpl[k] = sched[leaf].clone()
    return synth.synth_state_dict(tmpl, seed=seed)


def close(a, b, tol=2e-5):
    scale = float(b.abs().max()) + 1e-12
    err = float((a - b).abs().max())
    assert a.shape == b.shape, (a.shape, b.shape)
    assert err <= tol * max(scale, 1.0), f"max abs err {err:.3e} vs scale {scale:.3e}"


def test_schedule_buffers(golden):
    g = golden("ops_diffmvs.npz")
    sched = O.cosine_schedule(1000)
    for name, v in sched.items():
        ref = g.t(f"update_block.0.buf.{name}")
        assert torch.allclose(v, ref, rtol=1e-6, atol=1e-9), name


def test_warp_edge_cases(golden):
    g = golden("warp_edge.npz")
    for ci in range(int(g.np("n_cases"))):
        out = O.warp(g.t(f"c{ci}.src"), g.t(f"c{ci}.src_proj"), g.t(f"c{ci}.ref_proj"), g.t(f"c{ci}.depth"))
        ref = g.t(f"c{ci}.out")
        assert torch.isfinite(out).all()
        close(out, ref, 1e-4)


def test_warp_impls_agree(golden):
    """the grid_sample-based warp that bench.py times as the CPU baseline == the spelled-out checker == the reference"""
    g = golden("warp_edge.npz")
    for ci in range(int(g.np("n_cases"))):
        a = (g.t(f"c{ci}.src"), g.t(f"c{ci}.src_proj"), g.t(f"c{ci}.ref_proj"), g.t(f"c{ci}.depth"))
        close(O.warp_grid_sample(*a), g.t(f"c{ci}.out"), 1e-5)
        close(O.warp_grid_sample(*a), O.warp(*a), 1e-4)
    e = golden("e2e_diffmvs_b2.npz")
    meta = e.meta()
    sd = make_sd("diffmvs", meta["weight_seed"])
    args = synth.make_args("diffmvs", numdepth_initial=meta["nd_init"])
    imgs, proj, dv = synth.synth_inputs(meta["H"], meta["W"], meta["n_src"], B=meta["B"], seed=meta["scene_seed"])
    src = synth.NoiseSource(meta["noise_seed"])
    O.use_grid_sample_warp(True)
    try:
        with torch.no_grad():
            out = O.forward(sd, args, imgs, proj, dv, noise_fn=lambda shape: src(shape, "cpu"))
    finally:
        O.use_grid_sample_warp(False)
    for a, b in zip(out["depth"], e.seq("out.depth")):
        assert rel_l1(a, b) < 1e-5


@pytest.mark.parametrize("variant", ["diffmvs", "casdiffmvs"])
def test_feature_context(golden, variant):
    g = golden(f"ops_{variant}.npz")
    sd = make_sd(variant)
    f = O.feature_net(sd, g.t("feature#0.in.0"))
    for s in f:
        close(f[s], g.t(f"feature#0.out.{s}"))
    c = O.context_net(sd, g.t("context#0.in.0"))
    for s in c:
        close(c[s], g.t(f"context#0.out.{s}"))


def test_initial_cost_pieces(golden):
    g = golden("ops_diffmvs.npz")
    sd = make_sd("diffmvs")
    for n in (0, 1):
        w = O.pixel_view_weight(sd, g.t(f"depthnet.pixel_view_weight#{n}.in.0"))
        close(w, g.t(f"depthnet.pixel_view_weight#{n}.out"))
    pre = O.cost_reg(sd, g.t("depthnet.cost_regularization#0.in.0"))
    close(pre, g.t("depthnet.cost_regularization#0.out"), 5e-5)


def test_initial_cost_full(golden):
    g = golden("ops_diffmvs.npz")
    e = golden("e2e_diffmvs_cfg1.npz")
    sd = make_sd("diffmvs")
    feats = g.seq("depthnet#0.in.0")
    context, proj = g.t("depthnet#0.in.1"), g.t("depthnet#0.in.2")
    hyp = g.t("depthnet#0.kw.depth_values")
    dmin, dmax = torch.tensor(425.0).view(1, 1, 1, 1), torch.tensor(935.0).view(1, 1, 1, 1)
    dbg = {}
    mask, nd, depth, vw, conf = O.initial_cost(sd, feats, context, proj, hyp, dmin, dmax, 4, debug=dbg)
    close(dbg["cor"][0], g.t("depthnet.pixel_view_weight#0.in.0"))
    close(dbg["agg"], g.t("depthnet.cost_regularization#0.in.0"))
    close(mask, g.t("depthnet#0.out.0"))
    close(nd, g.t("depthnet#0.out.1"), 5e-5)
    close(depth, g.t("depthnet#0.out.2"), 5e-5)
    close(vw, g.t("depthnet#0.out.3"))
    # floor(index) is discontinuous: allow a few bin flips
    ref = g.t("depthnet#0.out.4")
    bad = ((conf - ref).abs() > 1e-4).float().mean()
    assert bad < 0.01
    close(depth, e.t("out.depth.0"), 5e-5)


@pytest.mark.parametrize("variant,calls", [("diffmvs", (0, 1)), ("casdiffmvs", (3, 4))])
def test_get_cost(golden, variant, calls):
    g = golden(f"ops_{variant}.npz")
    e = golden(f"e2e_{variant}_cfg1.npz")
    sd = make_sd(variant)
    args = synth.make_args(variant, numdepth_initial=32)
    meta = e.meta()
    imgs, proj, dv = synth.synth_inputs(meta["H"], meta["W"], meta["n_src"], B=meta["B"], seed=meta["scene_seed"])
    stage = 2 if variant == "diffmvs" else 3
    feats = [O.feature_net(sd, im)[f"stage{stage}"] for im in imgs]
    dmin, dmax = torch.tensor(425.0).view(1, 1, 1, 1), torch.tensor(935.0).view(1, 1, 1, 1)
    for n in calls:
        inv = g.t(f"GetCost#{n}.in.0")
        conf = g.t(f"GetCost#{n}.kw.confidence") if f"GetCost#{n}.kw.confidence" in g else None
        vw = g.t(f"GetCost#{n}.kw.view_weights")
        interval = float(g.np(f"GetCost#{n}.kw.depth_interval"))
        cost, samples = O.get_cost(feats, proj[f"stage{stage}"], inv, interval, dmax, dmin,
                                   args.CostNum[stage - 1], vw, conf, 4, args.min_radius, args.max_radius)
        close(samples, g.t(f"GetCost#{n}.out.1"))
        close(cost, g.t(f"GetCost#{n}.out.0"), 1e-4)
    assert any(f"GetCost#{n}.kw.confidence" in g for n in calls)


@pytest.mark.parametrize("variant,i", [("diffmvs", 0), ("casdiffmvs", 1)])
def test_update_nets(golden, variant, i):
    g = golden(f"ops_{variant}.npz")
    sd = make_sd(variant)
    args = synth.make_args(variant, numdepth_initial=32)
    p = f"update_block_depth{i + 2}"
    k = f"update_block.{i}"
    enc = O.condition_encoder(sd, p + ".encoder", g.t(f"{k}.encoder#0.in.0"), g.t(f"{k}.encoder#0.in.1"),
                              g.t(f"{k}.encoder#0.in.2"))
    close(enc, g.t(f"{k}.encoder#0.out"))
    h = O.sep_conv_gru(sd, p + ".unet.gru", g.t(f"{k}.unet.gru#0.in.0"), g.t(f"{k}.unet.gru#0.in.1"))
    close(h, g.t(f"{k}.unet.gru#0.out"))
    t = g.t(f"{k}.unet#0.in.2")
    te = O.time_mlp(sd, p + ".unet.time_mlp", t, args.unet_dim[i + 1])
    close(te, g.t(f"{k}.unet.time_mlp#0.out"))
    rb = O.resnet_block(sd, p + ".unet.downs.0.0", g.t(f"{k}.unet.downs.0.0#0.in.0"), te)
    close(rb, g.t(f"{k}.unet.downs.0.0#0.out"), 5e-5)
    hid, delta, conf = O.unet(sd, p + ".unet", g.t(f"{k}.unet#0.in.0"), g.t(f"{k}.unet#0.in.1"), t,
                              args.unet_dim[i + 1], i + 2)
    close(hid, g.t(f"{k}.unet#0.out.0"), 5e-5)
    close(delta, g.t(f"{k}.unet#0.out.1"), 5e-5)
    close(conf, g.t(f"{k}.unet#0.out.2"), 5e-5)
    m = O.mask_head(sd, g.t(f"{k}.mask#0.in.0"), p + ".mask")
    close(m, 0.25 * g.t(f"{k}.mask#0.out"))


@pytest.mark.parametrize("variant", ["diffmvs", "casdiffmvs"])
@pytest.mark.parametrize("cfg", ["cfg1", "b2"])
def test_end_to_end(golden, variant, cfg):
    e = golden(f"e2e_{variant}_{cfg}.npz")
    meta = e.meta()
    sd = make_sd(variant, meta["weight_seed"])
    args = synth.make_args(variant, numdepth_initial=meta["nd_init"])
    imgs, proj, dv = synth.synth_inputs(meta["H"], meta["W"], meta["n_src"], B=meta["B"], seed=meta["scene_seed"])
    src = synth.NoiseSource(meta["noise_seed"])
    with torch.no_grad():
        out = O.forward(sd, args, imgs, proj, dv, noise_fn=lambda shape: src(shape, "cpu"))
    ref = e.seq("out.depth")
    assert len(out["depth"]) == len(ref)
    for a, b in zip(out["depth"], ref):
        assert a.shape == b.shape
        assert rel_l1(a, b) < 1e-5, rel_l1(a, b)
    refc = e.seq("out.photometric_confidence")
    assert len(out["photometric_confidence"]) == len(refc)
    assert conf_close(out["photometric_confidence"][0], refc[0])       # stage 1: floor(index) bin flips allowed
    for a, b in zip(out["photometric_confidence"][1:], refc[1:]):
        assert rel_l1(a, b) < 1e-4
    for i, n in enumerate(e.seq("noise")):
        assert torch.equal(n, synth.synth_noise(n.shape, meta["noise_seed"], i))


@pytest.mark.parametrize("variant", ["diffmvs", "casdiffmvs"])
@pytest.mark.parametrize("tag", ["ms2", "evalall"])
def test_end_to_end_multistep_and_all_iterates(golden, variant, tag):
    """ms2: two DDIM sampling steps per refinement stage (reference update.py:504-519, second noise draw);
    evalall: test=False in eval mode -- every iterate and the Unet confidences (diffusion.py:264-270)."""
    e = golden(f"e2e_{variant}_b2_{tag}.npz")
    meta = e.meta()
    sd = make_sd(variant, meta["weight_seed"])
    args = synth.make_args(variant, numdepth_initial=meta["nd_init"], sampling_timesteps=meta["sampling_timesteps"])
    imgs, proj, dv = synth.synth_inputs(meta["H"], meta["W"], meta["n_src"], B=meta["B"], seed=meta["scene_seed"])
    src = synth.NoiseSource(meta["noise_seed"])
    with torch.no_grad():
        out = O.forward(sd, args, imgs, proj, dv, noise_fn=lambda shape: src(shape, "cpu"), test=meta["test"])
    check_outputs(out, e, 1e-5, 1e-4)


def check_outputs(out, e, tol_depth, tol_conf):
    ref = e.seq("out.depth")
    assert len(out["depth"]) == len(ref)
    for a, b in zip(out["depth"], ref):
        assert a.shape == b.shape
        assert rel_l1(a, b) < tol_depth, rel_l1(a, b)
    refc = e.seq("out.conf") if "out.conf.len" in e else []
    assert len(out["conf"]) == len(refc)
    for a, b in zip(out["conf"], refc):
        assert a.shape == b.shape
        assert rel_l1(a, b) < tol_conf, rel_l1(a, b)
    refp = e.seq("out.photometric_confidence")
    assert len(out["photometric_confidence"]) == len(refp)
    assert conf_close(out["photometric_confidence"][0], refp[0])
    for a, b in zip(out["photometric_confidence"][1:], refp[1:]):
        assert rel_l1(a, b) < tol_conf
