#!/usr/bin/env python3
"""DEV-ONLY (needs /root/reference): one reference training step in train mode (BatchNorm batch statistics, the
train branch of the update block, models/update.py:423-464) on deterministic inputs, with torch.randint /
torch.randn_like replaced by recorded draws -> tests/golden/train_<variant>.npz: loss, every output depth /
confidence map, per-prefix gradient norms, a few full gradients, updated BN running statistics."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from make_golden import import_reference  # noqa: E402
from diffmvs_amd import synth  # noqa: E402

FULL_GRADS = ["feature.conv0.0.conv.weight", "feature.out2.weight", "context.output2.weight",
              "depthnet.pixel_view_weight.conv.0.conv.weight", "depthnet.cost_regularization.conv6.conv.weight",
              "depthnet.cost_regularization.prob.weight", "update_block_depth2.unet.gru.convq1.weight",
              "update_block_depth2.unet.downs.0.0.block1.proj.weight", "update_block_depth2.unet.downs.0.0.mlp.1.weight",
              "update_block_depth2.unet.time_mlp.1.weight", "update_block_depth2.encoder.output.bias",
              "hidden_init.0.1.weight", "depthnet.mask.2.weight"]


def main():
    ref_models, _, _ = import_reference()
    torch.set_num_threads(8)
    for variant in ("diffmvs", "casdiffmvs"):
        H, W, S, B, nd = 64, 96, 2, 2, 8
        args = synth.make_args(variant, numdepth_initial=nd)
        model = ref_models.CasDiffMVS(args, test=False)
        model.load_state_dict(synth.synth_state_dict(model.state_dict(), 123), strict=True)
        model.train()
        imgs, proj, dv, gt, mask = synth.synth_inputs(H, W, S, B=B, seed=21, with_gt=True)
        src = synth.NoiseSource(31)
        draws_t, draws_n = [], []
        real_randn, real_randint = torch.randn_like, torch.randint

        def fake_randn(t, *a, **k):
            n = src(t.shape, t.device).to(t.dtype)
            draws_n.append(n.clone())
            return n

        def fake_randint(lo, hi, size, *a, **k):
            rs = np.random.RandomState(1234 + len(draws_t))
            v = torch.from_numpy(rs.randint(lo, hi, size=tuple(size)).astype(np.int64))
            draws_t.append(v.clone())
            return v

        torch.randn_like, torch.randint = fake_randn, fake_randint
        try:
            out = model(imgs, proj, dv, gt)
            loss, parts = ref_models.compute_inverse_loss(args, out["depth"], out["conf"], gt, mask, dv, loss_rate=0.9,
                                                          iters=args.stage_iters)
            loss.backward()
        finally:
            torch.randn_like, torch.randint = real_randn, real_randint
        res = {"loss": np.array(float(loss))}
        for i, d in enumerate(out["depth"]):
            res[f"depth.{i}"] = d.detach().numpy()
        for i, c in enumerate(out["conf"]):
            res[f"conf.{i}"] = c.detach().numpy()
        res["n_depth"], res["n_conf"] = np.array(len(out["depth"])), np.array(len(out["conf"]))
        for i, t in enumerate(draws_t):
            res[f"t.{i}"] = t.numpy()
        for i, n in enumerate(draws_n):
            res[f"noise.{i}"] = n.numpy()
        res["n_t"], res["n_noise"] = np.array(len(draws_t)), np.array(len(draws_n))
        norms = {}
        seen = set()
        for name, p in model.named_parameters():
            if p.grad is None or id(p) in seen:
                continue
            seen.add(id(p))
            pre = name.split(".")[0]
            norms[pre] = norms.get(pre, 0.0) + float(p.grad.double().pow(2).sum())
            res["gnorm." + name] = np.array(float(p.grad.double().norm()))
        res["prefix_norms"] = np.array(json.dumps({k: v ** 0.5 for k, v in norms.items()}))
        named = dict(model.named_parameters())
        for k in FULL_GRADS:
            if k in named and named[k].grad is not None:
                res["grad." + k] = named[k].grad.numpy().copy()
        # optimisation step of train.py:200-203 with the optimizer of train.py:321-326 (constant lr: first step)
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-3, eps=1e-8)
        before = {k: named[k].detach().clone() for k in FULL_GRADS if k in named}
        total_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 2.0)
        opt.step()
        res["total_grad_norm"] = np.array(float(total_norm))
        for k, b in before.items():
            res["post." + k] = named[k].detach().numpy()
        res["post_sum"] = np.array(sum(float(p.detach().double().sum()) for p in model.parameters()))
        res["post_abs"] = np.array(sum(float(p.detach().double().abs().sum()) for p in model.parameters()))
        sd = model.state_dict()
        for k in ("feature.conv0.0.bn.running_mean", "feature.conv0.0.bn.running_var", "context.conv1.bn.running_mean",
                  "depthnet.pixel_view_weight.conv.0.bn.running_var", "depthnet.cost_regularization.conv5.bn.running_mean",
                  "feature.conv0.0.bn.num_batches_tracked"):
            res["buf." + k] = sd[k].numpy()
        res["meta"] = np.array(json.dumps(dict(H=H, W=W, S=S, B=B, nd_init=nd, scene_seed=21, noise_seed=31, weight_seed=123,
                                               torch=torch.__version__)))
        np.savez_compressed(os.path.join(HERE, f"train_{variant}.npz"), **res)
        print(variant, "loss", float(loss), "n_depth", len(out["depth"]), "n_conf", len(out["conf"]), "t", [t.tolist() for t in draws_t],
              {k: round(v ** 0.5, 4) for k, v in norms.items()})


if __name__ == "__main__":
    main()
