#!/usr/bin/env python3
"""DEV-ONLY fixture generator.  Runs only where /root/reference is mounted (the build
container); the GPU box never sees the reference, it only sees the .npz / .json files
this script wrote.  Nothing here is imported by the product or by the tests.

What it does: imports the reference's `models` package on CPU (with the one import-time
patch SURVEY F5 describes: models/module.py:7 touches cuda:0), loads the deterministic
weights of diffmvs_amd.synth into it, feeds it the deterministic scenes of
diffmvs_amd.synth with the deterministic diffusion noise of diffmvs_amd.synth, and records
 * end-to-end outputs  (test.py:124-125 call)            -> e2e_<variant>_<cfg>.npz
 * inputs/outputs of every hot-path sub-module (hooks)   -> ops_<variant>.npz
 * stand-alone differentiable_warping edge cases         -> warp_edge.npz
 * state-dict key/shape listing                          -> state_keys_<variant>.json
Only data is written: tensors in, tensors out.
"""
import hashlib
import json
import os
import re
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from diffmvs_amd import synth  # noqa: E402


def import_reference():
    """module.py:7 runs torch.inverse(torch.ones((1,1), device='cuda:0')) at import."""
    real_ones = torch.ones

    def ones_cpu(*a, **k):
        k.pop("device", None)
        return real_ones(*a, **k)

    torch.ones = ones_cpu
    sys.path.insert(0, REF)
    try:
        import models as ref_models  # noqa
        import models.module as ref_module  # noqa
        import models.update as ref_update  # noqa
    finally:
        torch.ones = real_ones
        sys.path.remove(REF)
    return ref_models, ref_module, ref_update


def flat(prefix, obj, out):
    if obj is None:
        return
    if torch.is_tensor(obj):
        out[prefix] = obj.detach().cpu().numpy().copy()
    elif isinstance(obj, (list, tuple)):
        out[prefix + ".len"] = np.array(len(obj))
        for i, o in enumerate(obj):
            flat(f"{prefix}.{i}", o, out)
    elif isinstance(obj, dict):
        for k, o in obj.items():
            flat(f"{prefix}.{k}", o, out)
    elif isinstance(obj, (int, float)):
        out[prefix] = np.array(obj)
    elif callable(obj):
        return
    else:
        raise TypeError(type(obj))


def dedupe(d):
    """identical arrays (a module's output fed to the next one) are stored once: '@key' alias."""
    seen, out = {}, {}
    for k, v in d.items():
        if v.nbytes < 4096:
            out[k] = v
            continue
        h = (v.shape, str(v.dtype), hashlib.sha1(np.ascontiguousarray(v).tobytes()).hexdigest())
        if h in seen:
            out[k] = np.array("@" + seen[h])
        else:
            seen[h] = k
            out[k] = v
    return out


class Recorder:
    """forward hooks -> flat dict; keeps the first `keep` calls of each module."""

    def __init__(self, keep=2):
        self.data = {}
        self.count = {}
        self.keep = keep
        self.handles = []

    def hook(self, name, module, keep=None):
        keep = self.keep if keep is None else keep

        def fn(mod, args, kwargs, output):
            n = self.count.get(name, 0)
            self.count[name] = n + 1
            if n >= keep:
                return
            flat(f"{name}#{n}.in", list(args), self.data)
            kw = {k: v for k, v in kwargs.items()
                  if k not in ("features", "proj_matrices", "scale_inv_depth")}
            flat(f"{name}#{n}.kw", kw, self.data)
            flat(f"{name}#{n}.out", output, self.data)
        self.handles.append(module.register_forward_hook(fn, with_kwargs=True))

    def remove(self):
        for h in self.handles:
            h.remove()


def build_ref_model(ref_models, variant, nd_init, seed=123):
    args = synth.make_args(variant, numdepth_initial=nd_init)
    model = ref_models.CasDiffMVS(args, test=True)
    sd = synth.synth_state_dict(model.state_dict(), seed=seed)
    model.load_state_dict(sd, strict=True)
    model.eval()
    return model, args


def run_forward(model, imgs, proj, dv, noise_seed):
    """Forward with torch.randn_like replaced by the deterministic synth noise stream."""
    src = synth.NoiseSource(noise_seed)
    drawn = []
    real = torch.randn_like

    def fake(t, *a, **k):
        n = src(t.shape, t.device).to(t.dtype)
        drawn.append(n.clone())
        return n

    torch.randn_like = fake
    try:
        with torch.no_grad():
            out = model(imgs, proj, dv)
    finally:
        torch.randn_like = real
    return out, drawn


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_models, ref_module, ref_update = import_reference()
    meta = {"torch": torch.__version__, "numpy": np.__version__}

    for variant in ("diffmvs", "casdiffmvs"):
        n_src = 5
        # ---- state-dict listing (full-size model; key set does not depend on nd_init)
        model, args = build_ref_model(ref_models, variant, nd_init=32)
        keys = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in model.state_dict().items()}
        with open(os.path.join(HERE, f"state_keys_{variant}.json"), "w") as f:
            json.dump({"n_params": sum(p.numel() for p in model.parameters()), "keys": keys}, f, indent=0)

        # ---- cfg1: B=1, 128x160, 5 src, numdepth_initial=32 (BASELINE.json configs[0]) with hooks
        H, W, B = 128, 160, 1
        imgs, proj, dv = synth.synth_inputs(H, W, n_src, B=B, seed=1)
        rec = Recorder(keep=1)
        rec.hook("feature", model.feature)
        rec.hook("context", model.context)
        cas = variant == "casdiffmvs"
        if not cas:     # stage-1/2 modules are identical in structure for both variants: record once
            rec.hook("depthnet", model.depthnet)
            rec.hook("depthnet.pixel_view_weight", model.depthnet.pixel_view_weight, keep=2)
            rec.hook("depthnet.cost_regularization", model.depthnet.cost_regularization)
        rec.hook("GetCost", model.GetCost, keep=(2 if not cas else 5))
        for i, hi in enumerate(model.hidden_init):
            if cas and i == 0:
                continue
            rec.hook(f"hidden_init.{i}", hi)
        for i, ub in enumerate(model.update_block):
            if cas and i == 0:
                continue
            rec.hook(f"update_block.{i}", ub)
            rec.hook(f"update_block.{i}.encoder", ub.encoder, keep=2)
            rec.hook(f"update_block.{i}.unet", ub.unet)
            rec.hook(f"update_block.{i}.unet.gru", ub.unet.gru)
            rec.hook(f"update_block.{i}.unet.init_conv", ub.unet.init_conv)
            rec.hook(f"update_block.{i}.unet.time_mlp", ub.unet.time_mlp)
            rec.hook(f"update_block.{i}.unet.downs.0.0", ub.unet.downs[0][0])
            rec.hook(f"update_block.{i}.unet.downs.0.1", ub.unet.downs[0][1])
            rec.hook(f"update_block.{i}.unet.mid", ub.unet.mid)
            rec.hook(f"update_block.{i}.unet.ups.0.1", ub.unet.ups[0][1])
            rec.hook(f"update_block.{i}.mask", ub.mask)
            for bname in SCHEDULE:
                rec.data[f"update_block.{i}.buf.{bname}"] = getattr(ub, bname).numpy().copy()
        out, drawn = run_forward(model, imgs, proj, dv, noise_seed=5)
        rec.remove()
        e2e = {}
        flat("out", out, e2e)
        flat("noise", drawn, e2e)
        e2e["in.imgs_sum"] = np.array([float(i.double().sum()) for i in imgs])
        e2e["in.proj_stage1"] = proj["stage1"].numpy()
        e2e["meta"] = np.array(json.dumps(dict(meta, H=H, W=W, B=B, n_src=n_src, nd_init=32,
                                               scene_seed=1, noise_seed=5, weight_seed=123)))
        np.savez_compressed(os.path.join(HERE, f"e2e_{variant}_cfg1.npz"), **e2e)
        if cas:   # GetCost calls 0-2 are stage 2 (recorded for diffmvs); keep the stage-3 calls 3,4
            rec.data = {k: v for k, v in rec.data.items() if not re.match(r"GetCost#[012]\.", k)}
        ops = dedupe({k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in rec.data.items()})
        ops["meta"] = e2e["meta"]
        np.savez_compressed(os.path.join(HERE, f"ops_{variant}.npz"), **ops)
        fin = out["depth"][-1]
        print(variant, "cfg1 final depth: min %.2f max %.2f mean %.2f std %.2f" %
              (fin.min(), fin.max(), fin.mean(), fin.std()),
              "| saturated frac", float(((fin < 426) | (fin > 934)).float().mean()),
              "| n_out", len(out["depth"]), "| ops MB %.2f" % (sum(v.nbytes for v in ops.values()) / 1e6))
        for i, d in enumerate(out["depth"]):
            print("   depth[%d] %s mean %.2f std %.2f" % (i, tuple(d.shape), d.mean(), d.std()))

        # ---- B=2, 64x96, nd_init=16, 3 src: batch-index coverage
        model2, _ = build_ref_model(ref_models, variant, nd_init=16)
        imgs, proj, dv = synth.synth_inputs(64, 96, 3, B=2, seed=2)
        out, drawn = run_forward(model2, imgs, proj, dv, noise_seed=9)
        e2e = {}
        flat("out", out, e2e)
        flat("noise", drawn, e2e)
        e2e["meta"] = np.array(json.dumps(dict(meta, H=64, W=96, B=2, n_src=3, nd_init=16,
                                               scene_seed=2, noise_seed=9, weight_seed=123)))
        np.savez_compressed(os.path.join(HERE, f"e2e_{variant}_b2.npz"), **e2e)

    # ---- stand-alone warping edge cases: strong rotation, points behind the camera, z == 0,
    #      source resolution different from the hypothesis grid (module.py:181-218)
    rs = np.random.RandomState(77)
    warp = {}
    cases = []
    for ci, (C, D, H0, W0, Hs, Ws, ang, tz) in enumerate([
        (8, 5, 12, 16, 12, 16, 0.05, 0.0),     # mild
        (8, 4, 10, 14, 12, 18, 0.9, -300.0),   # large rotation: big out-of-bounds regions
        (4, 6, 8, 8, 8, 8, 2.6, -900.0),       # camera looks backwards: negative z everywhere/partially
        (12, 3, 9, 11, 7, 13, 0.3, 50.0),      # source grid != hypothesis grid
    ]):
        B = 2
        src = torch.from_numpy(rs.standard_normal((B, C, Hs, Ws)).astype(np.float32))
        K = np.array([[20.0, 0, W0 / 2], [0, 20.0, H0 / 2], [0, 0, 1]], np.float32)
        ref_proj = np.tile(np.eye(4, dtype=np.float32)[None], (B, 1, 1))
        src_proj = np.tile(np.eye(4, dtype=np.float32)[None], (B, 1, 1))
        for b in range(B):
            a = ang * (1 + 0.5 * b)
            R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)
            E = np.eye(4, dtype=np.float32)
            E[:3, :3] = R
            E[:3, 3] = [-40.0, 10.0, tz]
            src_proj[b, :3, :4] = K @ E[:3, :4]
            ref_proj[b, :3, :4] = K @ np.eye(4, dtype=np.float32)[:3, :4]
        depth = torch.from_numpy(rs.uniform(300, 1000, (B, D, H0, W0)).astype(np.float32))
        out = ref_module.differentiable_warping(src, torch.from_numpy(src_proj), torch.from_numpy(ref_proj), depth)
        warp[f"c{ci}.src"] = src.numpy()
        warp[f"c{ci}.src_proj"] = src_proj
        warp[f"c{ci}.ref_proj"] = ref_proj
        warp[f"c{ci}.depth"] = depth.numpy()
        warp[f"c{ci}.out"] = out.numpy()
        cases.append(ci)
        print("warp case", ci, "nonzero frac %.3f" % float((out != 0).float().mean()))
    # exact z == 0 case (module.py:206): identity rotation, t_z = -depth
    B, C, D, H0, W0 = 1, 4, 2, 6, 6
    src = torch.from_numpy(rs.standard_normal((B, C, H0, W0)).astype(np.float32))
    ref_proj = np.eye(4, dtype=np.float32)[None].copy()
    src_proj = np.eye(4, dtype=np.float32)[None].copy()
    src_proj[0, 2, 3] = -512.0
    depth = torch.full((B, D, H0, W0), 512.0)
    depth[:, 1] = 640.0
    out = ref_module.differentiable_warping(src, torch.from_numpy(src_proj), torch.from_numpy(ref_proj), depth)
    warp["c4.src"], warp["c4.src_proj"], warp["c4.ref_proj"] = src.numpy(), src_proj, ref_proj
    warp["c4.depth"], warp["c4.out"] = depth.numpy(), out.numpy()
    warp["n_cases"] = np.array(5)
    np.savez_compressed(os.path.join(HERE, "warp_edge.npz"), **warp)
    print("done")


SCHEDULE = synth.SCHEDULE_BUFFERS

if __name__ == "__main__":
    main()
