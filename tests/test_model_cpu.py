"""CPU: the drop-in model boundary -- checkpoint layout, loud failure without a HIP device, and the
whole engine driven end to end through the host emulation of the kernel sources against the
reference-generated goldens."""
import pytest
import torch

from conftest import conf_close, emu_ops, rel_l1, state_keys
from diffmvs_amd import synth
from diffmvs_amd._lib import DmvsError


@pytest.mark.parametrize("variant", ["diffmvs", "casdiffmvs"])
def test_state_dict_layout(variant):
    from models import CasDiffMVS
    model = CasDiffMVS(synth.make_args(variant, numdepth_initial=32), test=True)
    want = state_keys(variant)
    got = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in model.state_dict().items()}
    assert set(got) == set(want["keys"])
    for k, v in want["keys"].items():
        assert got[k] == v, k
    assert sum(p.numel() for p in model.parameters()) == want["n_params"]


def test_schedule_buffers_match_reference(golden):
    from models import CasDiffMVS
    g = golden("ops_diffmvs.npz")
    model = CasDiffMVS(synth.make_args("diffmvs"), test=True)
    for name in synth.SCHEDULE_BUFFERS:
        assert torch.allclose(getattr(model.update_block_depth2, name), g.t(f"update_block.0.buf.{name}"),
                              rtol=1e-6, atol=1e-9), name


def test_no_cpu_path():
    """The product refuses to run anywhere but on a HIP device: no silent fallback."""
    from models import CasDiffMVS
    model = CasDiffMVS(synth.make_args("diffmvs", numdepth_initial=8), test=True).eval()
    imgs, proj, dv = synth.synth_inputs(32, 32, 1, B=1, seed=0)
    with pytest.raises(DmvsError):
        model(imgs, proj, dv)
    model.train()
    with pytest.raises(ValueError):            # train mode needs the ground truth (reference diffusion.py:169)
        model(imgs, proj, dv)
    imgs, proj, dv, gt, _ = synth.synth_inputs(32, 32, 1, B=1, seed=0, with_gt=True)
    with pytest.raises(DmvsError):
        model(imgs, proj, dv, gt)


@pytest.mark.parametrize("variant", ["diffmvs", "casdiffmvs"])
def test_end_to_end_emulated(golden, variant):
    """B=2 golden through the full engine with the kernels running on the host emulation."""
    from models import CasDiffMVS
    e = golden(f"e2e_{variant}_b2.npz")
    meta = e.meta()
    model = CasDiffMVS(synth.make_args(variant, numdepth_initial=meta["nd_init"]), test=True).eval()
    model.load_state_dict(synth.synth_state_dict(model.state_dict(), meta["weight_seed"]), strict=True)
    imgs, proj, dv = synth.synth_inputs(meta["H"], meta["W"], meta["n_src"], B=meta["B"], seed=meta["scene_seed"])
    eng = model.engine(emu_ops())
    out = eng.forward(imgs, proj, dv, noise_fn=synth.NoiseSource(meta["noise_seed"]))
    ref = e.seq("out.depth")
    assert len(out["depth"]) == len(ref)
    for i, (a, b) in enumerate(zip(out["depth"], ref)):
        assert a.shape == b.shape
        assert rel_l1(a, b) < 1e-4, (i, rel_l1(a, b))
    refc = e.seq("out.photometric_confidence")
    assert len(out["photometric_confidence"]) == len(refc)
    for a, b in zip(out["photometric_confidence"], refc):
        assert a.shape == b.shape
    assert conf_close(out["photometric_confidence"][0], refc[0])      # stage 1: floor(index) bin flips allowed
    for a, b in zip(out["photometric_confidence"][1:], refc[1:]):
        assert rel_l1(a, b) < 1e-3


@pytest.mark.parametrize("variant,tag", [("diffmvs", "ms2"), ("casdiffmvs", "evalall")])
def test_multistep_and_all_iterates_emulated(golden, variant, tag):
    """ms2: two DDIM steps per refinement stage (reference update.py:504-519); evalall: a test=False model in eval mode
    returns every iterate + the Unet confidences (diffusion.py:264-270).  Through CasDiffMVS.forward's own dispatch."""
    from models import CasDiffMVS
    e = golden(f"e2e_{variant}_b2_{tag}.npz")
    meta = e.meta()
    args = synth.make_args(variant, numdepth_initial=meta["nd_init"], sampling_timesteps=meta["sampling_timesteps"])
    model = CasDiffMVS(args, test=meta["test"]).eval()
    model.load_state_dict(synth.synth_state_dict(model.state_dict(), meta["weight_seed"]), strict=True)
    imgs, proj, dv = synth.synth_inputs(meta["H"], meta["W"], meta["n_src"], B=meta["B"], seed=meta["scene_seed"])
    model.engine(emu_ops())
    model.noise_source = synth.NoiseSource(meta["noise_seed"])
    out = model(imgs, proj, dv)
    ref = e.seq("out.depth")
    assert len(out["depth"]) == len(ref)
    for i, (a, b) in enumerate(zip(out["depth"], ref)):
        assert a.shape == b.shape
        assert rel_l1(a, b) < 1e-4, (i, rel_l1(a, b))
    refc = e.seq("out.conf") if "out.conf.len" in e else []
    assert len(out["conf"]) == len(refc)
    for a, b in zip(out["conf"], refc):
        assert a.shape == b.shape and rel_l1(a, b) < 1e-3
    refp = e.seq("out.photometric_confidence")
    assert len(out["photometric_confidence"]) == len(refp)
    assert conf_close(out["photometric_confidence"][0], refp[0])
    for a, b in zip(out["photometric_confidence"][1:], refp[1:]):
        assert rel_l1(a, b) < 1e-3


def test_validation_loss_on_eval_outputs(golden):
    """the reference's validation loop (train.py test_sample_depth) feeds a test=False model's eval outputs to
    compute_inverse_loss, which asserts one depth map per supervised iterate and indexes conf[-1]"""
    from models import CasDiffMVS, compute_inverse_loss
    args = synth.make_args("diffmvs", numdepth_initial=8)
    model = CasDiffMVS(args, test=False).eval()
    model.load_state_dict(synth.synth_state_dict(model.state_dict(), 5), strict=True)
    imgs, proj, dv, gt, mask = synth.synth_inputs(32, 64, 2, B=1, seed=0, with_gt=True)
    model.engine(emu_ops())
    model.noise_source = synth.NoiseSource(1)
    out = model(imgs, proj, dv)
    loss, parts = compute_inverse_loss(args, out["depth"], out["conf"], gt, mask, dv, loss_rate=0.9, iters=args.stage_iters)
    assert torch.isfinite(loss)


@pytest.mark.parametrize("variant,prec", [("casdiffmvs", "bf16"), ("diffmvs", "fp16")])
def test_reduced_precision_feature_storage_emulated(golden, variant, prec):
    """BASELINE.json configs[2] / [4] name bf16 / fp16, which the reference itself never runs.  The build's
    reduced-precision mode stores the image FEATURES in 16 bits and keeps every computation fp32, so it must agree tightly
    with the oracle run on the same rounded features, and loosely with the fp32 reference golden."""
    import torch as _t
    from models import CasDiffMVS
    from oracle import diffmvs_oracle as O
    e = golden(f"e2e_{variant}_b2.npz")
    meta = e.meta()
    args = synth.make_args(variant, numdepth_initial=meta["nd_init"], precision=prec)
    model = CasDiffMVS(args, test=True).eval()
    sd = synth.synth_state_dict(model.state_dict(), meta["weight_seed"])
    model.load_state_dict(sd, strict=True)
    imgs, proj, dv = synth.synth_inputs(meta["H"], meta["W"], meta["n_src"], B=meta["B"], seed=meta["scene_seed"])
    eng = model.engine(emu_ops())
    assert eng.precision == prec
    out = eng.forward(imgs, proj, dv, noise_fn=synth.NoiseSource(meta["noise_seed"]))
    src = synth.NoiseSource(meta["noise_seed"])
    with _t.no_grad():
        want = O.forward(sd, args, imgs, proj, dv, noise_fn=lambda shape: src(shape, "cpu"),
                         feature_dtype={"bf16": _t.bfloat16, "fp16": _t.float16}[prec])
    errs = [rel_l1(a, b) for a, b in zip(out["depth"], want["depth"])]
    assert max(errs) < 1e-4, errs
    loose = [rel_l1(a, b) for a, b in zip(out["depth"], e.seq("out.depth"))]
    print(variant, prec, "vs fp32 reference:", ["%.2e" % x for x in loose])
    assert max(loose) < (2e-3 if prec == "bf16" else 2e-4), loose


@pytest.mark.parametrize("variant", ["casdiffmvs", "diffmvs"])
def test_bf16_matrix_arithmetic_emulated(golden, variant):
    """conv_arith = "bf16" (+ bf16 feature storage: BASELINE.json configs[2] as stated): the multi-tap 2-D convolutions round
    their inputs and weights to bf16 on the way into the matrix cores and accumulate in fp32.  Against the oracle doing the
    same rounding (O.forward(conv_dtype=, feature_dtype=)) the depth maps agree to rounding-order noise amplified by the
    network (a flipped bf16 rounding of an activation is a 4e-3 relative step); against the fp32 reference they stay inside a
    stated band -- and outside the fp32 path's own agreement, i.e. the mode is really on."""
    import torch as _t
    from models import CasDiffMVS
    from oracle import diffmvs_oracle as O
    e = golden(f"e2e_{variant}_b2.npz")
    meta = e.meta()
    args = synth.make_args(variant, numdepth_initial=meta["nd_init"], precision="bf16", conv_arith="bf16")
    model = CasDiffMVS(args, test=True).eval()
    sd = synth.synth_state_dict(model.state_dict(), meta["weight_seed"])
    model.load_state_dict(sd, strict=True)
    imgs, proj, dv = synth.synth_inputs(meta["H"], meta["W"], meta["n_src"], B=meta["B"], seed=meta["scene_seed"])
    eng = model.engine(emu_ops())
    assert eng.conv_arith == "bf16" and eng.precision == "bf16"
    out = eng.forward(imgs, proj, dv, noise_fn=synth.NoiseSource(meta["noise_seed"]))
    src = synth.NoiseSource(meta["noise_seed"])
    with _t.no_grad():
        want = O.forward(sd, args, imgs, proj, dv, noise_fn=lambda shape: src(shape, "cpu"), feature_dtype=_t.bfloat16,
                         conv_dtype=_t.bfloat16)
    errs = [rel_l1(a, b) for a, b in zip(out["depth"], want["depth"])]
    loose = [rel_l1(a, b) for a, b in zip(out["depth"], e.seq("out.depth"))]
    print(variant, "bf16 arithmetic vs the bf16 oracle:", ["%.2e" % x for x in errs], "vs the fp32 reference:", ["%.2e" % x for x in loose])
    assert max(errs) < 2e-3, errs
    assert 1e-5 < max(loose) < 2e-2, loose


@pytest.mark.parametrize("variant", ["casdiffmvs", "diffmvs"])
def test_split_bf16_arithmetic_emulated(golden, variant):
    """conv_arith = "split": the multi-tap 2-D convolutions form every product from bf16 triples of their fp32 operands (six partial
    products on the bf16 matrix cores, fp32 accumulation).  fp32 accuracy: against the REFERENCE's recorded fp32 outputs the depth maps
    stay where the exact-fp32 path's are (1e-5 relative L1; the plain bf16 arithmetic is 1e-3) -- no rounded-operand oracle is involved."""
    from models import CasDiffMVS
    e = golden(f"e2e_{variant}_b2.npz")
    meta = e.meta()
    args = synth.make_args(variant, numdepth_initial=meta["nd_init"], conv_arith="split")
    model = CasDiffMVS(args, test=True).eval()
    sd = synth.synth_state_dict(model.state_dict(), meta["weight_seed"])
    model.load_state_dict(sd, strict=True)
    imgs, proj, dv = synth.synth_inputs(meta["H"], meta["W"], meta["n_src"], B=meta["B"], seed=meta["scene_seed"])
    eng = model.engine(emu_ops())
    assert eng.conv_arith == "split" and eng.precision == "fp32"
    out = eng.forward(imgs, proj, dv, noise_fn=synth.NoiseSource(meta["noise_seed"]))
    errs = [rel_l1(a, b) for a, b in zip(out["depth"], e.seq("out.depth"))]
    print(variant, "split arithmetic vs the fp32 reference:", ["%.2e" % x for x in errs])
    assert max(errs) < 1e-5, errs


def test_image_sizes_that_are_not_multiples_of_32_are_rejected():
    """the reference's own skip connections only line up on multiples of 32 (module.py:444-445); the engine says so before any launch instead of
    reading past its tensors (the transposed 3-D convolution's residual on a 72 x 104 input: found under ASan on the host emulation)"""
    from diffmvs_amd import _lib
    from models import CasDiffMVS
    args = synth.make_args("diffmvs", numdepth_initial=8)
    model = CasDiffMVS(args, test=True).eval()
    model.load_state_dict(synth.synth_state_dict(model.state_dict(), 3), strict=True)
    imgs, proj, dv = synth.synth_inputs(72, 104, 2, B=1, seed=1)
    with pytest.raises(_lib.DmvsError, match="multiples of 32"):
        model.engine(emu_ops()).forward(imgs, proj, dv, noise_fn=synth.NoiseSource(1))
