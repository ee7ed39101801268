"""Scene-mode evaluation: every image of a scene through FeatureNet ONCE (diffmvs_amd.engine.SceneFeatureStore), each reference view
gathering its rows -- against the per-sample forward of the reference's harness (test.py:92-127: FeatureNet on all V images of every
reference view), which it must reproduce BIT FOR BIT, and against the CPU oracle."""
import pytest
import torch

from conftest import rel_l1
from diffmvs_amd import synth


def _model(variant, ops_device, nd=8):
    from models import CasDiffMVS
    args = synth.make_args(variant, numdepth_initial=nd)
    model = CasDiffMVS(args, test=True).eval()
    sd = synth.synth_state_dict(model.state_dict(), 123)
    model.load_state_dict(sd, strict=True)
    return model.to(ops_device), sd, args


def _run_both(model, scene, ref_ids, device, noise_seed=3):
    imgs, proj, dv, view_ids = synth.scene_batch(scene, ref_ids)
    imgs, proj, dv = [i.to(device) for i in imgs], {k: v.to(device) for k, v in proj.items()}, dv.to(device)
    model.noise_source = synth.NoiseSource(noise_seed)
    with torch.no_grad():
        per_sample = model(imgs, proj, dv)
    store = model.scene_features(scene["images"].to(device), chunk=4)
    model.noise_source = synth.NoiseSource(noise_seed)
    with torch.no_grad():
        cached = model(imgs[:1], proj, dv, feats=store.gather(view_ids))
    return per_sample, cached, (imgs, proj, dv)


@pytest.mark.parametrize("variant", ["diffmvs", "casdiffmvs"])
def test_scene_store_reproduces_the_per_sample_forward(ops, variant):
    dev = ops.device
    model, sd, args = _model(variant, dev)
    scene = synth.synth_scene(64, 96, n_views=9, n_src=3, seed=5, grid_w=3)
    assert scene["pairs"].shape == (9, 3) and all(int(v) not in scene["pairs"][v].tolist() for v in range(9))
    per_sample, cached, (imgs, proj, dv) = _run_both(model, scene, [4, 0, 7], dev)
    for key in ("depth", "photometric_confidence"):
        assert len(per_sample[key]) == len(cached[key])
        for a, b in zip(per_sample[key], cached[key]):
            assert torch.equal(a, b)
    if variant == "diffmvs":          # and the per-sample forward on these general (non-rig) cameras is the reference's: CPU oracle
        from oracle import diffmvs_oracle as O
        src = synth.NoiseSource(3)
        with torch.no_grad():
            want = O.forward(sd, args, [i.cpu() for i in imgs], {k: v.cpu() for k, v in proj.items()}, dv.cpu(), noise_fn=lambda shape: src(shape, "cpu"))
        errs = [rel_l1(a.cpu(), b) for a, b in zip(cached["depth"], want["depth"])]
        assert max(errs) < 1e-3, errs


def test_scene_features_are_validated(ops):
    """a feature stack that does not belong to the batch (wrong row count, wrong stage size, a missing stage) is refused before any launch"""
    from diffmvs_amd import _lib
    dev = ops.device
    model, _, _ = _model("diffmvs", dev)
    scene = synth.synth_scene(64, 96, n_views=4, n_src=2, seed=1, grid_w=2)
    imgs, proj, dv, ids = synth.scene_batch(scene, [0, 1])
    imgs, proj, dv = [i.to(dev) for i in imgs], {k: v.to(dev) for k, v in proj.items()}, dv.to(dev)
    store = model.scene_features(scene["images"].to(dev))
    good = store.gather(ids)
    bad = [{k: v[:-1] for k, v in good.items()}, {"stage1": good["stage1"]}, {"stage1": good["stage1"], "stage2": good["stage1"]},
           {k: v.double() for k, v in good.items()}]
    for feats in bad:
        with pytest.raises(_lib.DmvsError, match="feats"):
            with torch.no_grad():
                model(imgs[:1], proj, dv, feats=feats)
    # a camera tensor with fewer views than the gathered rows (the kernels would index the composed projections out of bounds), a wrong channel count
    short = {k: v[:, :2].contiguous() for k, v in proj.items()}
    with pytest.raises(_lib.DmvsError, match="proj_matrices"):
        with torch.no_grad():
            model(imgs[:1], short, dv, feats=good)
    with pytest.raises(_lib.DmvsError, match="feats"):
        with torch.no_grad():
            model(imgs[:1], proj, dv, feats={k: v[..., :16].contiguous() for k, v in good.items()})
    with torch.no_grad():
        out = model(imgs[:1], proj, dv, feats=good)
        # gather(out=...): into caller-owned buffers (a captured graph's static inputs), same rows
        bufs = {k: torch.empty_like(v) for k, v in good.items()}
        assert store.gather(ids, out=bufs) is bufs and all(torch.equal(bufs[k], good[k]) for k in good)
        out2 = model(imgs[:1], proj, dv, feats=bufs)
    assert torch.equal(out["depth"][0], out2["depth"][0])
    model.train()
    with pytest.raises(ValueError, match="eval-mode"):
        model(imgs, proj, dv, {"stage1": None}, feats=good)
    model.eval()


@pytest.mark.gpu
def test_scene_mode_full_size_bit_identical():
    """BASELINE.json configs[1] geometry: a 49-view scene at 640x512, 12 reference views of it through the store == per sample"""
    model, _, _ = _model("diffmvs", "cuda:0", nd=48)
    scene = synth.synth_scene(512, 640, n_views=14, n_src=5, seed=2)
    per_sample, cached, _ = _run_both(model, scene, list(range(12)), "cuda:0")
    for key in ("depth", "photometric_confidence"):
        for a, b in zip(per_sample[key], cached[key]):
            assert torch.equal(a, b)


@pytest.mark.gpu
def test_scene_mode_through_the_captured_graph():
    """the batch-1 operating point of the evaluation driver: store rows gathered into the graph's static feature buffers, replayed for
    two different reference views == the eager scene-mode forward"""
    class FixedNoise:
        def __init__(self):
            self.bufs, self.i = [], 0

        def rewind(self):
            self.i = 0

        def __call__(self, shape, device):
            if self.i == len(self.bufs):
                self.bufs.append(synth.synth_noise(shape, 7, self.i).to(device))
            self.i += 1
            return self.bufs[self.i - 1]

    model, _, _ = _model("casdiffmvs", "cuda:0", nd=16)
    scene = synth.synth_scene(128, 192, n_views=6, n_src=3, seed=4, grid_w=3)
    store = model.scene_features(scene["images"].cuda())
    noise = FixedNoise()
    model.noise_source = noise
    for ref in (2, 5):
        imgs, proj, dv, ids = synth.scene_batch(scene, [ref])
        args = ([imgs[0].cuda()], {k: v.cuda() for k, v in proj.items()}, dv.cuda())
        outs = []
        for graphs in (False, True):
            model.hip_graphs = graphs
            noise.rewind()
            with torch.no_grad():
                o = model(*args, feats=store.gather(ids))
            torch.cuda.synchronize()
            outs.append([d.clone() for d in o["depth"]])
        for a, b in zip(*outs):
            assert torch.equal(a, b)
