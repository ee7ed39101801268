"""Training driver: the counterpart of the reference's train.py on the MI355X path -- its epoch loop (train.py:98-143), its
checkpoint / resume convention (:137-141, :330-339) and its loaders (:353-362), as ONE PROCESS PER GPU instead of nn.DataParallel (:351).

    python -m diffmvs_amd.train_driver --logdir ./ckpt --epochs 16 --batch_size 4 --trainviews 9 --lr 1e-3 --lr_sche onecycle \\
        [--trainpath <tree> --trainlist scans.txt | --synthetic 256] [--resume] [--loadckpt model.ckpt] [--method casdiffmvs]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m diffmvs_amd.train_driver ...

What changes against the reference when the batch is spread over processes instead of DataParallel replicas:
  * `shuffle=True` (train.py:359-360) becomes a RANK-STRIDED sampler: one permutation of the sample indices per epoch, the same on every
    rank (seeded by --seed + epoch), rank r takes elements r, r + world, ...; drop_last like the reference's train loader -- the ranks
    see disjoint samples and the same number of steps;
  * the draws that a single process made from one global RNG must DIFFER between ranks: the source views of a training sample
    (`random.sample(src_views, nviews - 1)`, datasets/dtu.py:125) come from a per-(rank, epoch) `random.Random`, the diffusion step `t` and
    the noise of every refinement stage (models/update.py:432-433) from per-(rank, epoch) torch generators, injected through the model's
    t_source / noise_source hooks;
  * gradients: ONE all-reduce of the flat bucket per step (diffmvs_amd.trainer.Trainer), then the identical clip + AdamW on every rank --
    parameters stay bit-identical across ranks; BatchNorm running statistics stay per rank, rank 0's are saved (DataParallel keeps
    replica 0's, train.py:139 saves model.module);
  * rank 0 writes `<logdir>/model_{epoch:06d}.ckpt` = {'epoch', 'model', 'optimizer'} every --save_freq epochs; --resume loads the
    highest-numbered one on every rank (train.py:330-339) and continues the LR schedule where it stopped.
Samples follow the dict contract of the reference's training datasets (datasets/dtu.py:196-210: imgs, proj_matrices, depth_values, depth
and mask per stage); `TreeTrainSet` reads them from an MVS tree with ground-truth depth maps, `SyntheticTrainSet` renders them."""
from __future__ import annotations

import argparse
import json
import os
import random
import time
from typing import Dict, List, Sequence

import numpy as np
import torch

from . import formats as IO
from . import shard, synth

_STAGES = (("stage1", 8), ("stage2", 4), ("stage3", 2), ("stage4", 1))


# ------------------------------------------------------------------------------------------ datasets
class SyntheticTrainSet:
    """`n` rendered scenes (diffmvs_amd.synth: one slanted textured plane seen from `pool` + 1 cameras, ground-truth depth of the reference
    view per stage).  A sample = the reference view + nviews - 1 source views DRAWN from the pool by the caller's generator."""

    def __init__(self, n: int, H: int, W: int, nviews: int, pool: int | None = None, seed: int = 0, numdepth: int = 384):
        self.n, self.H, self.W, self.nviews, self.numdepth, self.seed = n, H, W, nviews, numdepth, seed
        self.pool = max(pool or (nviews + 1), nviews - 1)
        self._cache: Dict[int, tuple] = {}

    def __len__(self):
        return self.n

    def get(self, idx: int, rng: random.Random) -> dict:
        if idx not in self._cache:
            if len(self._cache) > 64:
                self._cache.clear()
            self._cache[idx] = synth.synth_inputs(self.H, self.W, self.pool, B=1, seed=self.seed + idx, numdepth=self.numdepth, with_gt=True)
        imgs, proj, dv, gt, mask = self._cache[idx]
        ids = [0] + sorted(rng.sample(range(1, self.pool + 1), self.nviews - 1))          # datasets/dtu.py:125
        return {"imgs": [imgs[v][0] for v in ids], "proj_matrices": {k: p[0, ids] for k, p in proj.items()}, "depth_values": dv[0],
                "depth": {k: g[0] for k, g in gt.items()}, "mask": {k: m[0] for k, m in mask.items()}, "index": idx, "view_ids": ids}


class TreeTrainSet:
    """An MVS tree in the evaluation layout (diffmvs_amd.formats.MVSDataset: <root>/<scan>/{images, cams_1 | cams, pair.txt}) that also holds
    `depth_gt/%08d.pfm` (and optionally `mask/%08d.png`) for its reference views.  Multi-scale ground truth like datasets/dtu.py:100-112:
    nearest-neighbour subsampling by 8 / 4 / 2 / 1; the mask defaults to `depth_min < depth < depth_max`."""

    def __init__(self, root: str, scans: Sequence[str], nviews: int, numdepth: int = 384, dataset: str = "general"):
        self.ds = IO.MVSDataset(root, nviews, numdepth, dataset=dataset, scan=list(scans))
        self.root, self.nviews = root, nviews
        self.items = []
        for i, (scan, ref, srcs) in enumerate(self.ds.metas):
            base = os.path.join(root, scan) if dataset != "general" else root
            if os.path.exists(os.path.join(base, "depth_gt", f"{ref:08d}.pfm")) and len(srcs) >= nviews - 1:
                self.items.append((i, base))

    def __len__(self):
        return len(self.items)

    def get(self, idx: int, rng: random.Random) -> dict:
        i, base = self.items[idx]
        scan, ref, srcs = self.ds.metas[i]
        ids = [ref] + rng.sample(list(srcs), self.nviews - 1)                             # datasets/dtu.py:125
        loaded = [self.ds.load_view(scan, v) for v in ids]
        s = IO.make_sample([x[0] for x in loaded], [x[1] for x in loaded], [x[2] for x in loaded], loaded[0][3], loaded[0][4], self.ds.numdepth)
        depth = np.ascontiguousarray(IO.read_pfm(os.path.join(base, "depth_gt", f"{ref:08d}.pfm"))[0]).astype(np.float32)
        H, W = loaded[0][0].shape[:2]
        if depth.shape != (H, W):
            raise ValueError(f"{base}/depth_gt/{ref:08d}.pfm: {depth.shape} does not match the image {(H, W)}")
        mfile = os.path.join(base, "mask", f"{ref:08d}.png")
        if os.path.exists(mfile):
            from PIL import Image
            m = (np.array(Image.open(mfile).convert("L")) > 10).astype(np.float32)
        else:
            m = ((depth > loaded[0][3]) & (depth < loaded[0][4])).astype(np.float32)
        out = {"imgs": [torch.from_numpy(a) for a in s["imgs"]], "proj_matrices": {k: torch.from_numpy(p) for k, p in s["proj_matrices"].items()},
               "depth_values": torch.from_numpy(s["depth_values"]), "depth": {}, "mask": {}, "index": idx, "view_ids": ids}
        for name, f in _STAGES:
            out["depth"][name] = torch.from_numpy(np.ascontiguousarray(depth[::f, ::f]))
            out["mask"][name] = torch.from_numpy(np.ascontiguousarray(m[::f, ::f]))
        return out


def collate_train(samples: Sequence[dict], device) -> dict:
    """what default_collate makes of a list of training samples (train.py:359), moved to the device (tocuda, train.py:184)"""
    V = len(samples[0]["imgs"])
    st = lambda xs: torch.stack(list(xs)).to(device)  # noqa: E731
    return {"imgs": [st(s["imgs"][v] for s in samples) for v in range(V)],
            "proj_matrices": {k: st(s["proj_matrices"][k] for s in samples) for k in samples[0]["proj_matrices"]},
            "depth_values": st(s["depth_values"] for s in samples),
            "depth": {k: st(s["depth"][k] for s in samples) for k in samples[0]["depth"]},
            "mask": {k: st(s["mask"][k] for s in samples) for k in samples[0]["mask"]}}


# ------------------------------------------------------------------------------------------ sampler and per-rank streams
class RankStridedSampler:
    """DistributedSampler semantics for the reference's `shuffle=True, drop_last=True` train loader (train.py:359-360): the permutation of
    an epoch is a function of (seed, epoch) only -- identical on every rank -- and rank r owns positions r, r + world, ... of its first
    `world * (n // world)` entries, so the ranks' samples are disjoint and equally many."""

    def __init__(self, n: int, rank: int, world: int, seed: int = 0, shuffle: bool = True):
        if not (0 <= rank < world):
            raise ValueError(f"rank {rank} outside world {world}")
        self.n, self.rank, self.world, self.seed, self.shuffle = n, rank, world, seed, shuffle

    def indices(self, epoch: int) -> List[int]:
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + epoch)
            perm = torch.randperm(self.n, generator=g).tolist()
        else:
            perm = list(range(self.n))
        return perm[self.rank:self.world * (self.n // self.world):self.world]

    def batches(self, epoch: int, batch_size: int) -> List[List[int]]:
        idx = self.indices(epoch)
        return [idx[i:i + batch_size] for i in range(0, len(idx) - batch_size + 1, batch_size)]      # drop_last


class RankStreams:
    """The random draws of one (rank, epoch): source views (python), diffusion step t (host generator), stage noise (device generator).
    Distinct per rank -- under the reference's single process they all came from one global stream -- and re-creatable on resume."""

    def __init__(self, seed: int, rank: int, epoch: int, device):
        base = (seed * 1000003 + rank * 7919 + epoch * 104729) & 0x7FFFFFFF
        self.views = random.Random(base + 1)
        self.t_gen = torch.Generator().manual_seed(base + 2)
        self.noise_gen = torch.Generator(device=device).manual_seed(base + 3)
        self.device = torch.device(device)
        self.t_log: List[List[int]] = []

    def t_source(self, B, T, device):
        t = torch.randint(0, T, (B,), generator=self.t_gen)
        self.t_log.append(t.tolist())
        return t.to(device).long()

    def noise_source(self, shape, device):
        return torch.randn(shape, generator=self.noise_gen, device=self.device).to(device)

    def install(self, model):
        model.t_source, model.noise_source = self.t_source, self.noise_source


# ------------------------------------------------------------------------------------------ the loop
def latest_checkpoint(logdir: str):
    """train.py:330-335: the highest-numbered *.ckpt of the log directory"""
    if not os.path.isdir(logdir):
        return None
    saved = sorted((fn for fn in os.listdir(logdir) if fn.endswith(".ckpt")), key=lambda x: int(x.split("_")[-1].split(".")[0]))
    return os.path.join(logdir, saved[-1]) if saved else None


def build_dataset(a):
    if a.trainpath:
        scans = [""]
        if a.trainlist:
            with open(a.trainlist) as f:
                scans = [ln.strip() for ln in f if ln.strip()]
        return TreeTrainSet(a.trainpath, scans, a.trainviews, a.numdepth, dataset=a.dataset)
    return SyntheticTrainSet(a.synthetic, a.height, a.width, a.trainviews, pool=a.view_pool, seed=a.seed, numdepth=a.numdepth)


def run(a, device=None, ops=None) -> dict:
    from models import CasDiffMVS
    from .trainer import Trainer, parse_lrepochs
    rank, world, local = shard.env_rank_world()
    if device is None:
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    if world > 1:
        shard.init_distributed(a.backend, device if str(device).startswith("cuda") else None)
    margs = synth.make_args("casdiffmvs" if a.method == "casdiffmvs" else "diffmvs", numdepth_initial=a.numdepth_initial, numdepth=a.numdepth)
    model = CasDiffMVS(margs, test=False)
    if a.loadckpt:
        model.load_state_dict(torch.load(a.loadckpt, map_location="cpu")["model"])             # train.py:340-344 (strict)
    else:
        model.load_state_dict(synth.synth_state_dict(model.state_dict(), 123 + (0 if a.same_init else rank)), strict=True)
    model.to(device)
    ds = build_dataset(a)
    sampler = RankStridedSampler(len(ds), rank, world, seed=a.seed)
    steps_per_epoch = len(sampler.batches(0, a.batch_size))
    if steps_per_epoch == 0:
        raise SystemExit(f"train_driver: {len(ds)} samples are fewer than one batch of {a.batch_size} on each of {world} ranks")
    kw = {}
    if a.lr_sche == "onecycle":
        kw["total_steps"] = steps_per_epoch * a.epochs + 100                                   # train.py:374
    elif a.lr_sche == "mslr":
        ms, gamma = parse_lrepochs(a.lrepochs)
        kw.update(milestones=ms, lr_gamma=gamma, steps_per_epoch=steps_per_epoch)
    tr = Trainer(model, margs, ops=ops, lr=a.lr, wd=a.wd, **kw)                                # broadcasts rank 0's weights and buffers
    start_epoch = 0
    if a.resume:
        ck = latest_checkpoint(a.logdir)
        if ck is None:
            raise SystemExit(f"--resume: no *.ckpt in {a.logdir}")
        start_epoch = tr.load_checkpoint(torch.load(ck, map_location=device))
        tr.step_count = steps_per_epoch * start_epoch                                          # the schedule continues where the epoch count says
    log = {"rank": rank, "world": world, "steps_per_epoch": steps_per_epoch, "start_epoch": start_epoch, "seen": [], "view_draws": [],
           "t_draws": [], "loss": [], "lr": []}
    for epoch in range(start_epoch, a.epochs):
        streams = RankStreams(a.seed, rank, epoch, device)
        streams.install(model)
        seen, draws = [], []
        for bi, idxs in enumerate(sampler.batches(epoch, a.batch_size)):
            t0 = time.time()
            samples = [ds.get(i, streams.views) for i in idxs]
            seen += idxs
            draws += [s["view_ids"] for s in samples]
            lr = tr.current_lr()
            loss, parts, gnorm, _ = tr.train_sample(collate_train(samples, device))
            log["loss"].append(float(loss))
            log["lr"].append(lr)
            if rank == 0 and not a.quiet:
                print("Epoch {}/{}, Iter {}/{}, lr {:.6f}, train loss = {:.3f}, grad norm = {:.3f}, time = {:.3f}".format(
                    epoch, a.epochs, bi, steps_per_epoch, lr, float(loss), float(gnorm), time.time() - t0), flush=True)
        log["seen"].append(seen)
        log["view_draws"].append(draws)
        log["t_draws"].append(streams.t_log)
        if (epoch + 1) % a.save_freq == 0 and rank == 0:                                       # train.py:136-141
            os.makedirs(a.logdir, exist_ok=True)
            torch.save(tr.checkpoint(epoch), os.path.join(a.logdir, "model_{:0>6}.ckpt".format(epoch)))
    flat = tr.flat.data.detach().double().cpu()
    log["weights_sum"], log["weights_abs_sum"], log["steps_done"] = float(flat.sum()), float(flat.abs().sum()), tr.step_count
    if world > 1:
        import torch.distributed as dist
        gathered = [torch.zeros_like(tr.flat.data) for _ in range(world)]
        dist.all_gather(gathered, tr.flat.data)
        log["weights_identical_across_ranks"] = bool(all(torch.equal(gathered[0], g) for g in gathered))
        dist.barrier()
    return log


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--logdir", default="./checkpoints/debug", help="checkpoint directory (train.py --logdir)")
    ap.add_argument("--resume", action="store_true", help="continue from the highest-numbered checkpoint of --logdir (train.py:330-339)")
    ap.add_argument("--loadckpt", default=None, help="initial weights: a reference checkpoint ({'model': state_dict})")
    ap.add_argument("--method", default="casdiffmvs", choices=["casdiffmvs", "diffmvs"])
    ap.add_argument("--trainpath", default=None, help="MVS tree with depth_gt/ (TreeTrainSet); default: --synthetic scenes")
    ap.add_argument("--trainlist", default=None)
    ap.add_argument("--dataset", default="general", choices=["dtu", "tank", "eth3d", "general"])
    ap.add_argument("--synthetic", type=int, default=64, help="number of rendered scenes when no --trainpath is given")
    ap.add_argument("--height", type=int, default=576)
    ap.add_argument("--width", type=int, default=768)
    ap.add_argument("--view_pool", type=int, default=None, help="source views rendered per synthetic scene (the draw picks trainviews - 1 of them)")
    ap.add_argument("--trainviews", type=int, default=9, help="images per sample: 1 reference + trainviews - 1 sources (scripts/train/*.sh)")
    ap.add_argument("--numdepth", type=int, default=384)
    ap.add_argument("--numdepth_initial", type=int, default=48)
    ap.add_argument("--epochs", type=int, default=16)
    ap.add_argument("--batch_size", type=int, default=4, help="per GPU (the reference's DataParallel split its batch ACROSS the GPUs)")
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--wd", type=float, default=1e-3)
    ap.add_argument("--lr_sche", default="onecycle", choices=["onecycle", "mslr", "const"])
    ap.add_argument("--lrepochs", default="10,12,14:2")
    ap.add_argument("--save_freq", type=int, default=1)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--same_init", type=int, default=1, help=argparse.SUPPRESS)       # 0: per-rank initial weights (test of the broadcast)
    ap.add_argument("--backend", default="nccl", help=argparse.SUPPRESS)
    ap.add_argument("--quiet", action="store_true")
    return ap.parse_args(argv)


def main(argv=None):
    log = run(parse_args(argv))
    print("TRAIN_DRIVER " + json.dumps({k: v for k, v in log.items() if k not in ("view_draws", "t_draws")}), flush=True)
    return log


if __name__ == "__main__":
    main()
