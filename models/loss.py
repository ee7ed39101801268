"""compute_inverse_loss with the reference's signature (models/loss.py:6-73): per-output L1 in
normalised inverse depth, confidence-weighted for the diffusion iterates, weights
loss_rate**(n-i-1).  Training-step only; plain tensor math."""
import torch

from .module import depth_to_disp


def compute_inverse_loss(args, inputs, confs, depth_gt_ms, mask_ms, depth_values, loss_rate=0.8, iters=[1, 3, 3]):
    n_out = len(inputs)
    if iters[2] == 0:       # DiffMVS
        stage_id = [1] * iters[0] + [2] * (iters[1] + 1) + [4]
        conf_flag = [False] * (iters[0] + 1) + [True] * iters[1] + [False]
    else:                   # CasDiffMVS
        stage_id = [1] * iters[0] + [2] * (iters[1] + 1) + [3] * (iters[2] + 1) + [4]
        conf_flag = [False] * (iters[0] + 1) + [True] * iters[1] + [False] + [True] * iters[2] + [False]
    assert n_out == len(stage_id), "input depths need to have the same number as stage_id."
    depth_max = 1.0 / depth_values[:, 0, None, None]
    depth_min = 1.0 / depth_values[:, -1, None, None]
    total, parts, ci = 0.0, {}, 0
    for i, est in enumerate(inputs):
        key = f"stage{stage_id[i]}"
        gt = depth_gt_ms[key]
        gt = torch.where(gt > 1e-4, gt, depth_max.view(-1, 1, 1).expand_as(gt))
        gt = depth_to_disp(gt, depth_min, depth_max)
        pred = depth_to_disp(est, depth_min, depth_max)
        valid = mask_ms[key] > 0.5
        l1 = (pred[valid] - gt[valid]).abs().mean()
        if conf_flag[i]:
            unc = (1 - confs[ci]).clamp(min=1e-6)
            ci += 1
            term = ((pred - gt).abs() / unc + args.conf_weight * torch.log(unc))[valid].mean()
        else:
            term = l1
        parts[f"l{i}"] = l1
        total = total + loss_rate ** (n_out - i - 1) * term
    return total, parts
