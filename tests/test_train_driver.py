"""The training driver (diffmvs_amd.train_driver; reference train.py:98-162, :330-362, datasets/dtu.py:125): the rank-strided sampler, the
per-rank random streams, and a world-size-2 run over gloo on the host emulation -- disjoint samples, different draws, identical weights,
the reference's checkpoint files, --resume."""
import json
import os
import random
import subprocess
import sys

import torch

from conftest import ROOT
from diffmvs_amd import train_driver as TD
from diffmvs_amd.trainer import one_cycle_lr


def test_rank_strided_sampler_partitions_every_epoch():
    n, world = 23, 4
    for epoch in (0, 1, 5):
        parts = [TD.RankStridedSampler(n, r, world, seed=3).indices(epoch) for r in range(world)]
        flat = [i for p in parts for i in p]
        assert len({len(p) for p in parts}) == 1 and len(parts[0]) == n // world          # equally many, drop_last
        assert len(set(flat)) == len(flat) and set(flat) <= set(range(n))                 # disjoint
    a, b = TD.RankStridedSampler(n, 0, world, seed=3), TD.RankStridedSampler(n, 0, world, seed=3)
    assert a.indices(2) == b.indices(2) and a.indices(2) != a.indices(3)                  # a function of (seed, epoch)
    assert TD.RankStridedSampler(n, 1, world, seed=3, shuffle=False).indices(0) == [1, 5, 9, 13, 17]
    bs = TD.RankStridedSampler(10, 0, 2, seed=0).batches(0, 2)
    assert [len(x) for x in bs] == [2, 2]                                                 # 5 indices -> two full batches, the rest dropped


def test_rank_streams_differ_per_rank_and_epoch_and_reproduce():
    def draws(rank, epoch):
        s = TD.RankStreams(7, rank, epoch, "cpu")
        return (s.views.sample(range(1, 30), 4), s.t_source(3, 1000, "cpu").tolist(), s.noise_source((2, 3), "cpu").flatten().tolist())
    assert draws(0, 0) == draws(0, 0)
    for other in (draws(1, 0), draws(0, 1)):
        assert all(x != y for x, y in zip(draws(0, 0), other))


def test_synthetic_train_set_follows_the_sample_contract():
    ds = TD.SyntheticTrainSet(3, 32, 64, nviews=3, pool=4, seed=5, numdepth=16)
    s = ds.get(1, random.Random(0))
    assert len(s["imgs"]) == 3 and s["imgs"][0].shape == (3, 32, 64) and s["view_ids"][0] == 0 and len(set(s["view_ids"])) == 3
    assert s["proj_matrices"]["stage2"].shape == (3, 2, 4, 4) and s["depth_values"].shape == (16,)
    assert s["depth"]["stage1"].shape == (4, 8) and s["mask"]["stage4"].shape == (32, 64)
    b = TD.collate_train([s, ds.get(2, random.Random(1))], "cpu")
    assert b["imgs"][1].shape == (2, 3, 32, 64) and b["depth"]["stage3"].shape == (2, 16, 32) and b["proj_matrices"]["stage1"].shape == (2, 3, 2, 4, 4)


def test_tree_train_set_reads_an_mvs_tree_with_ground_truth(tmp_path):
    """a `general` tree written by this package's writers + depth_gt/*.pfm: reference view first, source views drawn from pair.txt
    (datasets/dtu.py:125), nearest-subsampled multi-scale ground truth (datasets/dtu.py:100-112)"""
    import numpy as np
    from PIL import Image
    from diffmvs_amd import formats as IO, synth
    H, W, NV = 32, 64, 4
    sc = synth.synth_scene(H, W, n_views=NV, n_src=3, seed=1, numdepth=8)
    for d in ("images", "cams", "depth_gt"):
        os.makedirs(tmp_path / d)
    depth = np.full((H, W), 600.0, np.float32)
    depth[:4] = 0.0
    with open(tmp_path / "pair.txt", "w") as f:
        f.write(f"{NV}\n")
        for v in range(NV):
            Image.fromarray((sc["images"][v].permute(1, 2, 0).numpy() * 255).astype("uint8")).save(tmp_path / "images" / f"{v:08d}.jpg")
            cam = np.zeros((2, 4, 4), np.float32)
            cam[0], cam[1, :3, :3] = sc["E"][v].numpy(), sc["K"][v].numpy()
            IO.write_cam(str(tmp_path / "cams" / f"{v:08d}_cam.txt"), cam, 425.0, 935.0)      # an INPUT camera file: depth_min first (datasets/mvs.py:80-93)
            IO.save_pfm(str(tmp_path / "depth_gt" / f"{v:08d}.pfm"), depth)
            f.write(f"{v}\n3 " + " ".join(f"{int(s)} 1.0" for s in sc["pairs"][v]) + "\n")
    ds = TD.TreeTrainSet(str(tmp_path), [""], nviews=3, numdepth=8, dataset="general")
    assert len(ds) == NV
    s = ds.get(2, random.Random(3))
    assert s["view_ids"][0] == 2 and set(s["view_ids"][1:]) <= set(sc["pairs"][2].tolist()) and len(s["imgs"]) == 3
    assert s["depth"]["stage1"].shape == (H // 8, W // 8) and s["depth"]["stage4"].shape == (H, W)
    assert float(s["depth"]["stage2"][0, 0]) == 0.0 and float(s["depth"]["stage2"][2, 3]) == 600.0
    assert float(s["mask"]["stage4"][0, 0]) == 0.0 and float(s["mask"]["stage4"][10, 10]) == 1.0
    assert s["depth_values"].shape == (8,) and s["proj_matrices"]["stage3"].shape == (3, 2, 4, 4)


_WORKER = r"""
import os, sys, json, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
from conftest import emu_ops
from diffmvs_amd.ops import Ops
Ops.for_device = classmethod(lambda cls, device: emu_ops())      # this CPU worker runs the package on the host emulation
from diffmvs_amd import train_driver as TD
torch.set_num_threads(2)
a = TD.parse_args({argv!r})
log = TD.run(a, device=torch.device("cpu"), ops=emu_ops())
print("RESULT " + json.dumps(log), flush=True)
import torch.distributed as dist
if dist.is_initialized():
    dist.destroy_process_group()
"""


def _spawn(tmp_path, argv, world=2):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, argv=argv))
    port = 29500 + (os.getpid() % 2000) + 7
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    res = {}
    for p in procs:
        try:
            out, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, out[-3000:]
        r = json.loads([ln for ln in out.splitlines() if ln.startswith("RESULT ")][-1][7:])
        res[r["rank"]] = r
    return res


def test_two_ranks_over_gloo_disjoint_samples_different_draws_identical_weights(tmp_path):
    """world size 2, gloo, host emulation: 4 rendered scenes, batch 1 -> 2 steps per epoch and rank.  The ranks start from DIFFERENT weights
    (the Trainer broadcasts rank 0's), see disjoint samples, draw different source views / diffusion steps, and hold bit-identical
    parameters after every all-reduced step; rank 0 writes the reference's checkpoint files; --resume continues epoch count and schedule."""
    logdir = tmp_path / "ckpt"
    base = ["--method", "diffmvs", "--synthetic", "4", "--height", "32", "--width", "64", "--trainviews", "2", "--view_pool", "4",
            "--numdepth_initial", "8", "--batch_size", "1", "--lr_sche", "onecycle", "--logdir", str(logdir), "--backend", "gloo",
            "--same_init", "0", "--quiet", "--seed", "11"]
    res = _spawn(tmp_path, base + ["--epochs", "2"])
    r0, r1 = res[0], res[1]
    assert r0["steps_per_epoch"] == r1["steps_per_epoch"] == 2 and r0["steps_done"] == r1["steps_done"] == 4
    for e in range(2):
        assert not set(r0["seen"][e]) & set(r1["seen"][e]) and len(r0["seen"][e]) == len(r1["seen"][e]) == 2
        assert sorted(r0["seen"][e] + r1["seen"][e]) == [0, 1, 2, 3]                       # together: every sample once per epoch
    assert r0["t_draws"] != r1["t_draws"]                                                  # per-rank diffusion steps (update.py:432)
    assert r0["view_draws"] != r1["view_draws"] or r0["seen"] != r1["seen"]
    assert r0["weights_identical_across_ranks"] and r1["weights_identical_across_ranks"]
    assert r0["weights_sum"] == r1["weights_sum"] and r0["loss"] != r1["loss"]
    assert all(abs(lr - one_cycle_lr(k, 1e-3, 2 * 2 + 100)) < 1e-12 for k, lr in enumerate(r0["lr"]))      # train.py:374
    files = sorted(os.listdir(logdir))
    assert files == ["model_000000.ckpt", "model_000001.ckpt"]                              # train.py:136-141
    ck = torch.load(logdir / files[-1], map_location="cpu")
    assert set(ck) == {"epoch", "model", "optimizer"} and ck["epoch"] == 1 and len(ck["model"]) == 474
    # --resume: the highest-numbered checkpoint, the next epoch, the schedule where it stopped
    res2 = _spawn(tmp_path, base + ["--epochs", "3", "--resume"])
    q0 = res2[0]
    assert q0["start_epoch"] == 2 and len(q0["loss"]) == 2 and q0["steps_done"] == 6
    assert abs(q0["lr"][0] - one_cycle_lr(4, 1e-3, 2 * 3 + 100)) < 1e-12
    assert q0["weights_identical_across_ranks"] and sorted(os.listdir(logdir))[-1] == "model_000002.ckpt"
