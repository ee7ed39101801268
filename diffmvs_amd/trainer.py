"""The reference's optimisation step, data-parallel over MI355Xs (reference train.py:179-209, :321-326, :372-376).

One process per GPU.  All trainable parameters live in ONE flat fp32 bucket (FlatParams): parameters and their
.grad tensors are views into two contiguous buffers, so a training step's tail is
    backward  ->  ONE all-reduce of the gradient bucket (RCCL over xGMI; 3.7 MB for CasDiffMVS, latency-bound,
                  so no bucketing / overlap games -- SURVEY 8e)
              ->  dmvs_sumsq_f32 (global grad norm)  ->  dmvs_adamw_step_f32 (clip + 1/world average + AdamW)
i.e. three launches that read each array once, instead of ~600 per-tensor optimizer launches.
The LR schedule is the reference's OneCycle (linear anneal, pct_start 0.05, no momentum cycling); checkpoints use
the reference's layout {'epoch', 'model', 'optimizer'} with a torch.optim.AdamW-compatible optimizer state.
The reference's default `--lr_sche mslr` (MultiStepLR over epochs, `--lrepochs "10,12,14:2"`, train.py:34-36, :367-371) is
`milestones` / `lr_gamma` / `steps_per_epoch`; a constant lr is total_steps=None without milestones."""
from __future__ import annotations

import torch
import torch.distributed as dist

from .ops import Ops


def parse_lrepochs(spec: str = "10,12,14:2"):
    """the reference's `--lrepochs` string -> (milestone epochs, gamma): "10,12,14:2" = halve at epochs 10, 12, 14 (train.py:368-369)"""
    ms, rate = spec.split(":")
    return [int(e) for e in ms.split(",")], 1.0 / float(rate)


def multi_step_lr(step: int, base_lr: float, milestones, gamma: float, steps_per_epoch: int) -> float:
    """lr of optimisation step `step` (0-based) under torch's MultiStepLR stepped once per EPOCH (train.py:133, :370)"""
    epoch = step // steps_per_epoch
    return base_lr * gamma ** sum(1 for m in milestones if m <= epoch)


def one_cycle_lr(step: int, max_lr: float, total_steps: int, pct_start: float = 0.05, div_factor: float = 25.0,
                 final_div_factor: float = 1e4) -> float:
    """lr used by optimisation step number `step` (0-based) under torch's OneCycleLR(anneal_strategy='linear',
    three_phase=False) as configured at train.py:372-376"""
    if step >= total_steps:
        raise ValueError(f"OneCycle schedule exhausted: step {step} of {total_steps}")
    initial, final = max_lr / div_factor, max_lr / div_factor / final_div_factor
    end1 = float(pct_start * total_steps) - 1.0
    end2 = float(total_steps - 1)
    if step <= end1:
        return initial + (step / end1) * (max_lr - initial) if end1 > 0 else max_lr
    return max_lr + ((step - end1) / (end2 - end1)) * (final - max_lr)


class FlatParams:
    """Re-homes every (unique) parameter of `model` and its gradient into two flat fp32 buffers."""

    def __init__(self, model: torch.nn.Module):
        self.params, self.names, seen = [], [], set()
        for name, p in model.named_parameters():            # named_parameters() already skips the aliased block copies
            if id(p) not in seen and p.requires_grad:
                seen.add(id(p))
                self.params.append(p)
                self.names.append(name)
        dev = self.params[0].device
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4                     # keep every view 16-byte aligned
        self.numel = n
        self.data = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, self.offsets):
            self.data[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.data[o:o + p.numel()].view(p.shape)
            p.grad = self.grad[o:o + p.numel()].view(p.shape)

    def zero_grad(self):
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):           # someone may have set .grad = None (set_to_none)
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)

    def check_views(self):
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o or \
                    p.data.data_ptr() != self.data.data_ptr() + 4 * o:
                raise RuntimeError("a parameter or its .grad was re-allocated outside the flat bucket "
                                   "(use Trainer.zero_grad(), not zero_grad(set_to_none=True))")


class Trainer:
    milestones, lr_gamma, steps_per_epoch = None, 0.5, None      # MultiStepLR off unless the constructor sets it

    def __init__(self, model, args, ops: Ops | None = None, lr=1e-3, wd=1e-3, betas=(0.9, 0.999), eps=1e-8, max_norm=2.0,
                 total_steps: int | None = None, loss_rate=0.9, distributed: bool = True, milestones=None, lr_gamma: float = 0.5,
                 steps_per_epoch: int | None = None):
        """`total_steps` = len(loader) * epochs + 100 for the reference's onecycle schedule (train.py:374); `milestones` (+
        `lr_gamma`, `steps_per_epoch` = len(loader)) for its default MultiStepLR (parse_lrepochs); neither = constant lr"""
        if milestones is not None and (total_steps is not None or not steps_per_epoch):
            raise ValueError("MultiStepLR needs steps_per_epoch and excludes the onecycle schedule (total_steps)")
        self.milestones, self.lr_gamma, self.steps_per_epoch = milestones, lr_gamma, steps_per_epoch
        from models import compute_inverse_loss
        self.model, self.args, self.loss_fn = model, args, compute_inverse_loss
        dev = next(model.parameters()).device
        self.ops = ops if ops is not None else Ops.for_device(dev)
        self.flat = FlatParams(model)
        # every step goes zero_grad() -> backward: the convolution nodes may add their weight gradients straight into the bucket's views
        # (diffmvs_amd.autograd._Conv2dFn) instead of handing autograd one tensor + one add kernel per use
        model.grad_into_flat_bucket = True
        self.exp_avg = torch.zeros_like(self.flat.data)
        self.exp_avg_sq = torch.zeros_like(self.flat.data)
        self.sumsq = torch.zeros(1, dtype=torch.float64, device=dev)
        self.lr, self.wd, self.betas, self.eps, self.max_norm = lr, wd, betas, eps, max_norm
        self.total_steps, self.loss_rate = total_steps, loss_rate
        self.step_count = 0
        self.time_allreduce, self.allreduce_events = False, []        # bench: an event pair around the collective
        self.world = dist.get_world_size() if distributed and dist.is_available() and dist.is_initialized() else 1
        if self.world > 1:                                    # identical start on every rank (DDP's initial broadcast)
            dist.broadcast(self.flat.data, src=0)
            for b in model.buffers():
                dist.broadcast(b, src=0)

    def current_lr(self):
        if self.milestones is not None:
            return multi_step_lr(self.step_count, self.lr, self.milestones, self.lr_gamma, self.steps_per_epoch)
        return self.lr if self.total_steps is None else one_cycle_lr(self.step_count, self.lr, self.total_steps)

    def zero_grad(self):
        self.flat.zero_grad()

    def backward_and_step(self, loss):
        """loss.backward() -> all-reduce -> clip -> AdamW.  Returns the global (averaged) gradient norm as a device tensor."""
        o, f = self.ops, self.flat
        loss.backward()
        f.check_views()
        if self.world > 1:
            if self.time_allreduce:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            dist.all_reduce(f.grad, op=dist.ReduceOp.SUM)     # the one collective of the training path
            if self.time_allreduce:
                ev[1].record()
                self.allreduce_events.append(ev)
        o.sumsq(f.grad, self.sumsq)
        lr = self.current_lr()
        self.step_count += 1
        o.adamw_step(f.data, f.grad, self.exp_avg, self.exp_avg_sq, lr, self.betas[0], self.betas[1], self.eps, self.wd,
                     self.step_count, grad_scale=1.0 / self.world, sumsq=self.sumsq, max_norm=self.max_norm)
        return self.sumsq.sqrt() / self.world

    def train_sample(self, sample):
        """train.py:179-209 for one batch dict {imgs, proj_matrices, depth_values, depth, mask} already on the device"""
        self.model.train()
        self.zero_grad()
        out = self.model(sample["imgs"], sample["proj_matrices"], sample["depth_values"], sample["depth"])
        loss, parts = self.loss_fn(self.args, out["depth"], out["conf"], sample["depth"], sample["mask"],
                                   sample["depth_values"], loss_rate=self.loss_rate, iters=self.args.stage_iters)
        gnorm = self.backward_and_step(loss)
        return loss.detach(), parts, gnorm, out

    # ------------------------------------------------------------------ checkpoints (train.py:137-141, :339-343)
    def optimizer_state_dict(self):
        """torch.optim.AdamW-compatible: loads into the reference's optimizer on resume"""
        state = {}
        for i, (p, o) in enumerate(zip(self.flat.params, self.flat.offsets)):
            n = p.numel()
            state[i] = {"step": torch.tensor(float(self.step_count)),
                        "exp_avg": self.exp_avg[o:o + n].view(p.shape).clone(),
                        "exp_avg_sq": self.exp_avg_sq[o:o + n].view(p.shape).clone()}
        group = {"lr": self.current_lr() if self.total_steps is None or self.step_count < self.total_steps else 0.0,
                 "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(self.flat.params)))}
        if self.milestones is not None:
            group["initial_lr"] = self.lr                      # what MultiStepLR(last_epoch = start_epoch - 1) reads on resume
        if self.total_steps is not None:
            # what torch's OneCycleLR(last_epoch != -1) reads on resume (reference train.py:372-376; div_factor 25,
            # final_div_factor 1e4, cycle_momentum=False)
            group.update(initial_lr=self.lr / 25.0, max_lr=self.lr, min_lr=self.lr / 25.0 / 1e4)
        return {"state": state if self.step_count else {}, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd):
        for i, (p, o) in enumerate(zip(self.flat.params, self.flat.offsets)):
            st = sd["state"].get(i)
            if st is None:
                continue
            n = p.numel()
            self.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            self.step_count = int(float(st["step"]))

    def checkpoint(self, epoch):
        return {"epoch": epoch, "model": self.model.state_dict(), "optimizer": self.optimizer_state_dict()}

    def load_checkpoint(self, ck):
        sd = ck["model"]
        with torch.no_grad():
            own = self.model.state_dict()
            for k, v in sd.items():
                if k in own:
                    own[k].copy_(v)                            # in place: the flat views stay valid
        self.load_optimizer_state_dict(ck["optimizer"])
        return ck["epoch"] + 1
