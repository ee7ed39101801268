"""The optimisation-step tail and the data-parallel harness (reference train.py:179-209, :321-326, :372-376):
OneCycle schedule vs torch's, the flat-bucket clip + AdamW kernels vs torch.optim.AdamW, one full reference
training step including optimizer.step(), checkpoint layout, and a world-size-2 gloo run."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, rel_l1
from diffmvs_amd import synth
from diffmvs_amd.trainer import FlatParams, Trainer, one_cycle_lr
from test_train import _t_source


def test_one_cycle_matches_torch():
    for total, max_lr in ((250, 1e-3), (1337, 4e-4)):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([p], lr=max_lr)
        sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr, total, pct_start=0.05, cycle_momentum=False, anneal_strategy="linear")
        for k in range(total):
            assert abs(one_cycle_lr(k, max_lr, total) - opt.param_groups[0]["lr"]) < 1e-9 * max_lr + 1e-12, k
            opt.step()
            if k < total - 1:
                sch.step()
    with pytest.raises(ValueError):
        one_cycle_lr(10, 1e-3, 10)


@pytest.mark.parametrize("n,clip", [(1, 2.0), (1027, 2.0), (70001, 0.5), (4096, 1e9)])
def test_adamw_kernel_matches_torch(ops, n, clip):
    g = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=1e-3, weight_decay=1e-3, eps=1e-8)
    dev = ops.device
    p, m, v = p0.clone().to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    for step in range(1, 4):
        grad = torch.randn(n, generator=g) * (3.0 if step == 2 else 0.01)
        ref.grad = grad.clone()
        want_norm = torch.nn.utils.clip_grad_norm_([ref], clip)
        opt.step()
        gd = grad.to(dev)
        ss = ops.sumsq(gd)
        assert abs(float(ss.sqrt()) - float(want_norm)) < 1e-5 * float(want_norm)
        ops.adamw_step(p, gd, m, v, 1e-3, 0.9, 0.999, 1e-8, 1e-3, step, sumsq=ss, max_norm=clip)
        assert torch.allclose(p.cpu(), ref.detach(), rtol=1e-5, atol=1e-6), step
    # data-parallel averaging folded in: grad_scale = 1/world on a summed gradient == the averaged gradient
    p2, m2, v2 = p0.clone().to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    p3, m3, v3 = p0.clone().to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    gsum = (torch.randn(n, generator=g) * 5).to(dev)
    ops.adamw_step(p2, gsum, m2, v2, 1e-3, 0.9, 0.999, 1e-8, 1e-3, 1, grad_scale=0.25, sumsq=ops.sumsq(gsum), max_norm=clip)
    gavg = (gsum * 0.25).contiguous()
    ops.adamw_step(p3, gavg, m3, v3, 1e-3, 0.9, 0.999, 1e-8, 1e-3, 1, sumsq=ops.sumsq(gavg), max_norm=clip)
    assert torch.allclose(p2, p3, rtol=1e-6, atol=1e-7)


def _golden_trainer(variant, g, ops):
    from models import CasDiffMVS
    meta = g.meta()
    args = synth.make_args(variant, numdepth_initial=meta["nd_init"])
    model = CasDiffMVS(args, test=False)
    model.load_state_dict(synth.synth_state_dict(model.state_dict(), meta["weight_seed"]), strict=True)
    model.to(ops.device)
    model.noise_source = synth.NoiseSource(meta["noise_seed"])
    model.t_source = _t_source()
    imgs, proj, dv, gt, mask = synth.synth_inputs(meta["H"], meta["W"], meta["S"], B=meta["B"], seed=meta["scene_seed"], with_gt=True)
    dev = ops.device
    sample = {"imgs": [i.to(dev) for i in imgs], "proj_matrices": {k: v.to(dev) for k, v in proj.items()},
              "depth_values": dv.to(dev), "depth": {k: v.to(dev) for k, v in gt.items()},
              "mask": {k: v.to(dev) for k, v in mask.items()}}
    return Trainer(model, args, ops=ops, lr=1e-3, wd=1e-3), model, sample


def test_full_step_matches_reference(golden, ops):
    """forward(train) -> loss -> backward -> clip_grad_norm_(2.0) -> AdamW: parameters after the step vs the reference's"""
    variant = "diffmvs"
    g = golden(f"train_{variant}.npz")
    tr, model, sample = _golden_trainer(variant, g, ops)
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    loss, parts, gnorm, _ = tr.train_sample(sample)
    assert abs(float(loss) - float(g.np("loss"))) < 1e-4 * float(g.np("loss"))
    assert abs(float(gnorm) - float(g.np("total_grad_norm"))) < 2e-3 * float(g.np("total_grad_norm"))
    named = dict(model.named_parameters())
    for k in g.files:
        if not k.startswith("post."):
            continue
        name = k[5:]
        got, want, b4 = named[name].detach().cpu(), g.t(k), before[name].cpu()
        # compare the update itself (|update| <= lr per element on the first Adam step), where the gradient is not fp32 noise
        gr = g.t("grad." + name)
        sig = gr.abs() > 1e-3 * gr.abs().max()
        assert rel_l1((got - b4)[sig], (want - b4)[sig]) < 2e-2, name
        assert float((got - want).abs().max()) <= 2.1e-3                   # never more than the two opposite +-lr moves
    s_sum = sum(float(p.detach().double().sum()) for p in model.parameters())
    s_abs = sum(float(p.detach().double().abs().sum()) for p in model.parameters())
    assert abs(s_abs - float(g.np("post_abs"))) < 1e-5 * float(g.np("post_abs"))
    assert abs(s_sum - float(g.np("post_sum"))) < 1e-5 * float(g.np("post_abs"))
    # a second step runs (views into the flat bucket survived the first) and the loss moves
    loss2, _, _, _ = tr.train_sample(sample)
    assert torch.isfinite(loss2) and float(loss2) != float(loss)


def test_flat_bucket_and_checkpoint_layout(golden):
    from models import CasDiffMVS
    args = synth.make_args("casdiffmvs", numdepth_initial=8)
    model = CasDiffMVS(args, test=False)
    flat = FlatParams(model)
    n_unique = sum(p.numel() for p in model.parameters())
    assert n_unique == 925435                                  # the all-reduce payload
    assert flat.numel >= n_unique and flat.numel - n_unique < 4 * len(flat.params)
    for p, o in zip(flat.params, flat.offsets):
        assert p.data_ptr() == flat.data.data_ptr() + 4 * o and p.grad.data_ptr() == flat.grad.data_ptr() + 4 * o
        assert o % 4 == 0
    # aliased blocks share storage inside the bucket
    assert model.update_block[0].unet.final_conv.weight.data_ptr() == model.update_block_depth2.unet.final_conv.weight.data_ptr()
    # the optimizer state loads into the reference's optimizer class (train.py:321-326, :339-343)
    tr = Trainer.__new__(Trainer)
    tr.flat, tr.step_count, tr.total_steps = flat, 3, None
    tr.lr, tr.wd, tr.betas, tr.eps = 1e-3, 1e-3, (0.9, 0.999), 1e-8
    tr.exp_avg = torch.randn(flat.numel)
    tr.exp_avg_sq = torch.rand(flat.numel)
    tr.model = model
    ck = tr.checkpoint(7)
    assert set(ck) == {"epoch", "model", "optimizer"}
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-3, eps=1e-8)
    opt.load_state_dict(ck["optimizer"])
    st = opt.state[flat.params[5]]
    o, n = flat.offsets[5], flat.params[5].numel()
    assert torch.equal(st["exp_avg"].reshape(-1), tr.exp_avg[o:o + n]) and float(st["step"]) == 3.0
    # and back
    tr2 = Trainer.__new__(Trainer)
    tr2.flat, tr2.step_count = flat, 0
    tr2.exp_avg, tr2.exp_avg_sq = torch.zeros(flat.numel), torch.zeros(flat.numel)
    tr2.model = model
    assert tr2.load_checkpoint({"epoch": 7, "model": ck["model"], "optimizer": opt.state_dict()}) == 8
    used = torch.zeros(flat.numel, dtype=torch.bool)
    for p, o in zip(flat.params, flat.offsets):
        used[o:o + p.numel()] = True
    assert torch.equal(tr2.exp_avg[used], tr.exp_avg[used]) and tr2.step_count == 3


_WORKER = r"""
import os, sys, json, torch, torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
from conftest import emu_ops
from diffmvs_amd import synth
from diffmvs_amd.trainer import Trainer
from diffmvs_amd.shard import init_distributed
from models import CasDiffMVS
from test_train import _t_source
from diffmvs_amd.ops import Ops
Ops.for_device = classmethod(lambda cls, device: emu_ops())      # this CPU worker runs the package on the host emulation

def make(seed_w):
    args = synth.make_args("diffmvs", numdepth_initial=8)
    model = CasDiffMVS(args, test=False)
    model.load_state_dict(synth.synth_state_dict(model.state_dict(), seed_w), strict=True)
    return args, model

def sample(rank):
    imgs, proj, dv, gt, mask = synth.synth_inputs(32, 64, 1, B=1, seed=40 + rank, with_gt=True)
    return dict(imgs=imgs, proj_matrices=proj, depth_values=dv, depth=gt, mask=mask)

def set_draws(model, rank):
    model.noise_source = synth.NoiseSource(90 + rank)
    ts = _t_source()
    model.t_source = lambda B, T, dev: (ts(B, T, dev) + 37 * rank) % T

init_distributed("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.set_num_threads(2)
args, model = make(7 + rank)                 # deliberately different initial weights: the Trainer must broadcast rank 0's
set_draws(model, rank)
tr = Trainer(model, args, ops=emu_ops(), lr=1e-3, wd=1e-3)
loss, parts, gnorm, _ = tr.train_sample(sample(rank))
flat = tr.flat.data.clone()
gathered = [torch.zeros_like(flat) for _ in range(world)]
dist.all_gather(gathered, flat)
res = dict(rank=rank, loss=float(loss), gnorm=float(gnorm), same=bool(all(torch.equal(gathered[0], x) for x in gathered)))
if rank == 0:
    # serial restatement: both samples on rank-0 weights, gradients averaged, one clip + AdamW
    grads, losses = [], []
    for r in range(world):
        a2, m2 = make(7)
        set_draws(m2, r)
        t2 = Trainer(m2, a2, ops=emu_ops(), lr=1e-3, wd=1e-3, distributed=False)
        m2.train(); t2.zero_grad()
        s = sample(r)
        out = m2(s["imgs"], s["proj_matrices"], s["depth_values"], s["depth"])
        l, _ = t2.loss_fn(a2, out["depth"], out["conf"], s["depth"], s["mask"], s["depth_values"], loss_rate=0.9, iters=a2.stage_iters)
        l.backward()
        grads.append(t2.flat.grad.clone()); losses.append(float(l))
    gavg = sum(grads) / world
    a3, m3 = make(7)
    t3 = Trainer(m3, a3, ops=emu_ops(), lr=1e-3, wd=1e-3, distributed=False)
    p = torch.nn.Parameter(t3.flat.data.clone())
    p.grad = gavg.clone()
    opt = torch.optim.AdamW([p], lr=1e-3, weight_decay=1e-3, eps=1e-8)
    n = torch.nn.utils.clip_grad_norm_([p], 2.0)
    opt.step()
    used = torch.zeros(t3.flat.numel, dtype=torch.bool)
    for q, o in zip(t3.flat.params, t3.flat.offsets):
        used[o:o + q.numel()] = True
    d = (p.detach() - flat)[used].abs()
    moved = (flat - t3.flat.data)[used].abs()
    res.update(serial_norm=float(n), max_diff=float(d.max()), mean_diff=float(d.mean()), mean_move=float(moved.mean()),
               loss_serial=losses[0])
print("RESULT " + json.dumps(res), flush=True)
dist.barrier()
dist.destroy_process_group()
"""


def test_data_parallel_step_gloo(tmp_path):
    """world_size 2 over gloo: rank-0 weights broadcast, ONE all-reduce of the flat gradient bucket, identical
    parameters on both ranks afterwards, equal to the serial average-of-gradients step."""
    import json
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    port = 29500 + (os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, out[-3000:]
        outs.append(out)
    res = {}
    for out in outs:
        line = [ln for ln in out.splitlines() if ln.startswith("RESULT ")][-1]
        r = json.loads(line[7:])
        res[r["rank"]] = r
    assert res[0]["same"] and res[1]["same"]
    assert res[0]["loss"] != res[1]["loss"]                     # different samples per rank
    assert abs(res[0]["gnorm"] - res[1]["gnorm"]) < 1e-6 * res[0]["gnorm"]
    r0 = res[0]
    assert abs(r0["loss"] - r0["loss_serial"]) < 1e-5 * abs(r0["loss_serial"])
    assert abs(r0["gnorm"] - r0["serial_norm"]) < 1e-3 * r0["serial_norm"]
    assert r0["mean_diff"] < 0.02 * r0["mean_move"], r0          # first Adam step: |move| ~ lr everywhere


def test_onecycle_resume_roundtrip():
    """the reference's resume path (train.py:372-376) builds OneCycleLR(last_epoch = k) on the loaded optimizer, which
    reads initial_lr / max_lr / min_lr from the checkpoint's param group"""
    from models import CasDiffMVS
    args = synth.make_args("diffmvs", numdepth_initial=8)
    model = CasDiffMVS(args, test=False)
    flat = FlatParams(model)
    tr = Trainer.__new__(Trainer)
    tr.flat, tr.step_count, tr.total_steps = flat, 40, 500
    tr.lr, tr.wd, tr.betas, tr.eps = 1e-3, 1e-3, (0.9, 0.999), 1e-8
    tr.exp_avg, tr.exp_avg_sq = torch.zeros(flat.numel), torch.zeros(flat.numel)
    tr.model = model
    ck = tr.checkpoint(3)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-3, eps=1e-8)
    opt.load_state_dict(ck["optimizer"])
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, 1e-3, 500, pct_start=0.05, cycle_momentum=False, anneal_strategy="linear",
                                              last_epoch=tr.step_count - 1)
    # the scheduler resumes on the same curve the Trainer is on
    assert abs(opt.param_groups[0]["lr"] - one_cycle_lr(tr.step_count, 1e-3, 500)) < 1e-9
    opt.step()
    sch.step()
    assert abs(opt.param_groups[0]["lr"] - one_cycle_lr(tr.step_count + 1, 1e-3, 500)) < 1e-9


def test_submodule_eval_after_train_step(golden, ops):
    """packed-weight caches must notice parameters rewritten through raw pointers (dmvs_adamw_step_f32 does not bump
    torch's version counters): a sub-module forward and the whole-model engine after a training step use the NEW weights"""
    import models.module as M
    variant = "diffmvs"
    g = golden(f"train_{variant}.npz")
    tr, model, sample = _golden_trainer(variant, g, ops)
    enc = model.update_block_depth2.encoder
    B, H, W = 1, 8, 12
    dev = ops.device
    depth, samples, cost = (torch.rand(B, 1, H, W).to(dev), torch.rand(B, 6, H, W).to(dev), torch.rand(B, 24, H, W).to(dev))
    try:
        model.eval()
        y0 = enc(depth, samples, cost).clone()
        e0 = model.engine(ops)
        tr.train_sample(sample)
        model.eval()
        y1 = enc(depth, samples, cost).clone()
        assert model.engine(ops) is not e0                       # whole-model engine re-packed too
        # fresh module with the post-step weights = what a cold pack gives
        from models import CasDiffMVS
        fresh = CasDiffMVS(tr.args, test=False).eval()
        fresh.load_state_dict({k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, strict=True)
        fresh.to(dev)
        y2 = fresh.update_block_depth2.encoder(depth, samples, cost)
    finally:
        pass
    assert float((y1 - y0).abs().max()) > 0                      # the step moved the encoder
    assert torch.allclose(y1, y2, rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
def test_cfg4_full_size_training_step():
    """BASELINE.json configs[3] at its stated size on one GPU's share: CasDiffMVS, 768x576, 9 views (8 source), batch 4,
    fp32 -- forward(train) -> compute_inverse_loss -> backward -> clip(2.0) -> AdamW through Trainer.train_sample.
    Too large for the CPU oracle; size-independent properties: finite loss, every parameter receives a gradient, the
    global gradient norm is reproducible from identical weights + draws (the only atomics are the fp32 scatter-adds of
    the warp backward), the step moves the weights, the 10 + 6 outputs have the reference's shapes."""
    from models import CasDiffMVS
    dev = torch.device("cuda:0")
    H, W, S, B = 576, 768, 8, 4
    imgs, proj, dv, gt, mask = synth.synth_inputs(H, W, S, B=B, seed=3, with_gt=True)
    sample = {"imgs": [i.to(dev) for i in imgs], "proj_matrices": {k: v.to(dev) for k, v in proj.items()},
              "depth_values": dv.to(dev), "depth": {k: v.to(dev) for k, v in gt.items()},
              "mask": {k: v.to(dev) for k, v in mask.items()}}
    runs = []
    for _ in range(2):
        args = synth.make_args("casdiffmvs", numdepth_initial=48)
        model = CasDiffMVS(args, test=False)
        model.load_state_dict(synth.synth_state_dict(model.state_dict(), 123), strict=True)
        model.to(dev)
        model.noise_source = synth.NoiseSource(21)
        model.t_source = _t_source()
        tr = Trainer(model, args, lr=1e-3, wd=1e-3, total_steps=1000)
        before = tr.flat.data.clone()
        torch.cuda.reset_peak_memory_stats()
        loss, parts, gnorm, out = tr.train_sample(sample)
        torch.cuda.synchronize()
        assert torch.isfinite(loss) and torch.isfinite(gnorm) and float(gnorm) > 0
        assert len(out["depth"]) == 10 and len(out["conf"]) == 6
        assert out["depth"][-1].shape == (B, H, W) and out["depth"][0].shape == (B, H // 8, W // 8)
        dead = [n for n, p in zip(tr.flat.names, tr.flat.params) if not bool((p.grad != 0).any())]
        assert not dead, dead
        assert torch.isfinite(tr.flat.grad).all()
        moved = (tr.flat.data - before).abs()
        assert 0 < float(moved.mean()) and float(moved.max()) <= 1e-3      # first Adam step: |move| ~ lr (OneCycle start lr/25)
        runs.append((float(loss), float(gnorm)))
        print(f"cfg4 step: loss {float(loss):.6f} grad-norm {float(gnorm):.5f} peak memory "
              f"{torch.cuda.max_memory_allocated() / 2 ** 30:.2f} GiB")
    assert abs(runs[0][0] - runs[1][0]) <= 1e-5 * abs(runs[0][0])
    assert abs(runs[0][1] - runs[1][1]) <= 1e-3 * runs[0][1]


def test_multi_step_lr_matches_torch():
    """the reference's DEFAULT schedule (--lr_sche mslr, --lrepochs "10,12,14:2", train.py:34-36, :367-371): torch's
    MultiStepLR stepped once per epoch"""
    from diffmvs_amd.trainer import multi_step_lr, parse_lrepochs
    ms, gamma = parse_lrepochs("10,12,14:2")
    assert ms == [10, 12, 14] and gamma == 0.5
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=1e-3)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, ms, gamma=gamma)
    spe = 7
    for epoch in range(17):
        for i in range(spe):
            assert multi_step_lr(epoch * spe + i, 1e-3, ms, gamma, spe) == pytest.approx(opt.param_groups[0]["lr"], rel=1e-12)
        opt.step()
        sch.step()


_NCCL_WORKER = r"""
import os, sys, json, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from diffmvs_amd import synth
from diffmvs_amd.trainer import Trainer
from diffmvs_amd.shard import init_distributed
from models import CasDiffMVS
rank, local = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
init_distributed("nccl", dev)
args = synth.make_args("diffmvs", numdepth_initial=8)
model = CasDiffMVS(args, test=False)
model.load_state_dict(synth.synth_state_dict(model.state_dict(), 7 + rank), strict=True)      # the Trainer broadcasts rank 0's
model.to(dev).train()
model.noise_source = synth.NoiseSource(90 + rank)
tr = Trainer(model, args, lr=1e-3, wd=1e-3)
imgs, proj, dv, gt, mask = synth.synth_inputs(64, 96, 2, B=1, seed=40 + rank, with_gt=True)
mv = lambda d: {{k: v.to(dev) for k, v in d.items()}}
sample = dict(imgs=[i.to(dev) for i in imgs], proj_matrices=mv(proj), depth_values=dv.to(dev), depth=mv(gt), mask=mv(mask))
tr.zero_grad()
out = model(sample["imgs"], sample["proj_matrices"], sample["depth_values"], sample["depth"])
loss, _ = tr.loss_fn(args, out["depth"], out["conf"], sample["depth"], sample["mask"], sample["depth_values"], loss_rate=0.9, iters=args.stage_iters)
loss.backward()
local_grad = tr.flat.grad.clone()
grads = [torch.zeros_like(local_grad) for _ in range(dist.get_world_size())]
dist.all_gather(grads, local_grad)                     # what the all-reduce of the step must produce: the sum over ranks
tr.zero_grad()
loss2, parts, gnorm, _ = tr.train_sample(sample)       # (same draws are not needed: only the collective is under test)
flat = tr.flat.data.clone()
gathered = [torch.zeros_like(flat) for _ in range(dist.get_world_size())]
dist.all_gather(gathered, flat)
t = torch.ones(1 << 20, device=dev) * (rank + 1)
dist.all_reduce(t)
if rank == 0:
    print(json.dumps(dict(world=dist.get_world_size(), same=bool(all(torch.equal(gathered[0], x) for x in gathered)),
                          sum_ok=bool((t == 3.0).all()), finite=bool(torch.isfinite(flat).all()), loss=float(loss2),
                          grads_differ=bool(not torch.equal(grads[0], grads[1])))), flush=True)
dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_data_parallel_step_rccl_two_gpus(tmp_path):
    """The training path's one collective on the real transport: 2 ranks, backend "nccl" (= RCCL over xGMI), one process per GPU.
    Ranks start from different weights and train on different samples; after one Trainer.train_sample the parameters are
    bit-identical on both ranks (broadcast + all-reduce of the flat gradient bucket + identical clip / AdamW), and a plain
    all-reduce sums.  Skipped on a 1-GPU box (the arithmetic of the step is pinned by test_data_parallel_step_gloo)."""
    import json
    import socket
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs on the box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(_NCCL_WORKER.format(root=root))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    res = json.loads([ln for ln in outs[0].splitlines() if ln.startswith("{")][-1])
    assert res["world"] == 2 and res["same"] and res["sum_ok"] and res["finite"] and res["grads_differ"]
