"""Time the CasDiffMVS/DiffMVS training step of BASELINE.json configs[3] (forward + loss + backward + all-reduce +
clip + AdamW; 768x576, 9 views, batch 4 per GPU) -- one process per GPU, launched by torchrun for N > 1.
Rank 0 prints one JSON line (samples/s over all ranks)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffmvs_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="casdiffmvs")
    ap.add_argument("--H", type=int, default=576)
    ap.add_argument("--W", type=int, default=768)
    ap.add_argument("--src", type=int, default=8)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--profile", action="store_true")
    a = ap.parse_args()
    from models import CasDiffMVS, compute_inverse_loss
    args = synth.make_args(a.variant, numdepth_initial=48)
    model = CasDiffMVS(args, test=False)
    model.load_state_dict(synth.synth_state_dict(model.state_dict(), 123), strict=True)
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        from diffmvs_amd.shard import init_distributed
        init_distributed("nccl", dev)
    model.to(dev).train()
    from diffmvs_amd.trainer import Trainer
    tr = Trainer(model, args, total_steps=100000)
    imgs, proj, dv, gt, mask = synth.synth_inputs(a.H, a.W, a.src, B=a.batch, seed=3 + rank, with_gt=True)
    imgs = [i.to(dev) for i in imgs]
    proj = {k: v.to(dev) for k, v in proj.items()}
    gt = {k: v.to(dev) for k, v in gt.items()}
    mask = {k: v.to(dev) for k, v in mask.items()}
    dv = dv.to(dev)

    sample = {"imgs": imgs, "proj_matrices": proj, "depth_values": dv, "depth": gt, "mask": mask}

    def step():
        return tr.train_sample(sample)[0]

    step()
    torch.cuda.synchronize()
    if a.profile:
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
            step()
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=70))
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    if world > 1:
        from diffmvs_amd.shard import barrier_and_max
        dt = barrier_and_max(dt, dev)
    if rank == 0:
        print(json.dumps({"metric": "training samples/sec", "variant": a.variant, "H": a.H, "W": a.W, "V": a.src + 1,
                          "batch_per_gpu": a.batch, "n_gpus": world, "ms_per_step": dt * 1e3,
                          "value": a.batch * world / dt, "loss": float(loss.detach()),
                          "max_mem_GB": torch.cuda.max_memory_allocated() / 2 ** 30}))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
