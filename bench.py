#!/usr/bin/env python3
"""Headline benchmark: depth-maps/sec of the DiffMVS depth-estimation forward
(BASELINE.json configs[1]: DTU eval 640x512, 5 source views, numdepth_initial=48, 1 DDIM
step, fp32) on N MI355X of one node.

A step = one model() call on one batch of `--batch` reference views per GPU (inputs already
resident in HBM), timed like reference test.py:122-127 (device sync on both sides).  Reference
views shard across GPUs with no data-path collective (inference is embarrassingly parallel,
SURVEY section 8e) => weak scaling; value = all ranks' depth maps / max-over-ranks time.

Also reported on the same JSON line:
  roofline      the homography-warp kernel (getcost_quad_kernel, launched stage_iters[1]=4 times per
                step): algorithmic bytes per launch (SURVEY section 8d formula) / mean launch
                duration from HIP events recorded on the launch stream inside the timed region
  cpu_baseline  oracle/diffmvs_oracle.py (CPU restatement pinned to the reference) timed on this
                box's host cores on a bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from diffmvs_amd import synth  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 achievable)
FP32_MFMA_PEAK_TFS = 157.3   # v_mfma_f32_16x16x4_f32, exact fp32 (MI355X_MICROARCH.md)


def getcost_algorithmic_bytes(B, C, S, n, G, H, W):
    """4 * [C*HW (ref) + S*C*HW (src) + n*HW (hypotheses) + S*HW (view weights) + G*n*HW (out)]"""
    return 4 * B * H * W * (C + S * C + n + S + G * n)


def scene_geometry_getcost(ops, B, H, W, S, n, iters=20):
    """The GetCost kernel on the geometry a TRAINED network produces from its second GRU iteration on: hypotheses centred on the
    synthetic scene's true depth (sigma 0.01 of the normalised inverse-depth range), confidence 0.5.  The timed model above runs
    random-init weights, whose depth maps are noise (as is the first iteration of every diffusion stage of any network:
    scale * randn, update.py:472).  Untimed side measurement, same stage-2 shapes."""
    from diffmvs_amd.ops import Ops, g4_channels
    o = Ops(ops.lib, ops.device)
    dev = ops.device
    gi = synth.getcost_scene_inputs(H, W, S, B, stage=2, C=32, noise=0.01, conf=0.5)
    t = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in gi.items()}
    rt = o.compose_proj(t["proj"].float().contiguous())
    perm = g4_channels(32).to(dev)
    ref4, src4 = t["ref"][..., perm].contiguous(), t["src"][..., perm].contiguous()
    tail = (rt, t["inv"], t["conf"], t["view_w"], t["disp_min"], t["disp_max"], n, t["interval"], 0.25, 4.0, t["vw_shift"])
    out = {}
    for name, fn in (("quad", lambda: o.getcost_quad(ref4, src4, *tail)),):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(iters):
            fn()
        en.record()
        torch.cuda.synchronize()
        out[name] = st.elapsed_time(en) * 1e-3 / iters
    h2, w2 = H // 4, W // 4
    alg = getcost_algorithmic_bytes(B, 32, S, n, 4, h2, w2)
    return {"kernel": "getcost_quad_kernel<32,6> (hypotheses around the scene's true depth, sigma 0.01, confidence 0.5)",
            "bound": "hbm", "achieved": round(alg / out["quad"] / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(alg / out["quad"] / 1e9 / HBM_PEAK_GBS, 4), "traffic": None, "algorithmic_bytes_per_launch": alg,
            "avg_launch_us": round(out["quad"] * 1e6, 2)}


class ProbeLib:
    """ctypes binding of diffmvs_amd/libdmvs_probe.so (include/dmvs_probe.h): bench-only measurement kernels, loaded here and nowhere in
    the depth-estimation path.  Absent library => the probe legs are skipped and the line says so."""
    PATH = os.path.join(ROOT, "diffmvs_amd", "libdmvs_probe.so")

    def __init__(self):
        import ctypes as C
        from diffmvs_amd import _lib
        self.dll = C.CDLL(self.PATH)
        self.dll.dmvs_probe_abi_version.restype = C.c_int
        if self.dll.dmvs_probe_abi_version() != 2:
            raise RuntimeError("libdmvs_probe.so: unexpected ABI version")
        self.dll.dmvs_probe_getcost_loads_f32.argtypes = [C.POINTER(_lib.GetCostDesc), C.c_void_p]
        self.dll.dmvs_probe_getcost_loads_f32.restype = C.c_int
        self.dll.dmvs_probe_getcost_pair_loads_f32.argtypes = [C.POINTER(_lib.GetCostDesc), C.c_void_p]
        self.dll.dmvs_probe_getcost_pair_loads_f32.restype = C.c_int
        self.dll.dmvs_probe_random_line_gather.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_uint32, C.c_void_p]
        self.dll.dmvs_probe_random_line_gather.restype = C.c_int
        self._C = C

    def getcost_loads(self, desc, stream):
        rc = self.dll.dmvs_probe_getcost_loads_f32(self._C.byref(desc), stream)
        if rc:
            raise RuntimeError(f"dmvs_probe_getcost_loads_f32 -> {rc}")

    def getcost_pair_loads(self, desc, stream):
        rc = self.dll.dmvs_probe_getcost_pair_loads_f32(self._C.byref(desc), stream)
        if rc:
            raise RuntimeError(f"dmvs_probe_getcost_pair_loads_f32 -> {rc}")

    def random_line_gather(self, table, n_lines, n_quads, lines_per_quad, mode, window, seed, stream):
        rc = self.dll.dmvs_probe_random_line_gather(self._C.c_void_p(table.data_ptr()), n_lines, n_quads, lines_per_quad, mode, window, seed, stream)
        if rc:
            raise RuntimeError(f"dmvs_probe_random_line_gather -> {rc}")


def _event_us(fn, iters):
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) * 1e3 / iters


def getcost_ceiling_probe(model, eng, imgs, proj, dv, alg_bytes):
    """The memory-system ceiling of GetCost's address stream, measured BY THIS PROCESS (untimed legs, after the timed steps):
      in_step   one extra forward in which dmvs_probe_getcost_loads_f32 -- the product kernel's own hypotheses, projection, texel masks,
                addresses and loads with nothing computed from the loaded registers (csrc/probe/getcost_probe.hip) -- is launched right
                BEFORE each product GetCost launch, on the same stream, with the same descriptor: it sees the caches exactly as the product
                launch of a timed step does (the feature maps evicted by the convolutions since the last GRU iteration);
      isolated  the four launches of that step replayed from their recorded inputs, 12 warm-up launches, then product and probe in
                alternating blocks of back-to-back launches (warm caches, warm clocks: the figure round 5 quoted from a builder session);
      random_line_gather   the calibration kernel: 128-byte lines of a table of GetCost's source footprint requested as two 64-byte
                quad-coalesced pieces from 8 waves per SIMD with no arithmetic -- every line once in scrambled order (what the memory system
                retires for independent missing lines), uniformly random with repeats, and from overlapping 48-line bands (each line asked
                for by ~8 neighbouring quads close in time, the locality of texels scattered along epipolar segments)."""
    import ctypes as C
    from diffmvs_amd import _lib
    probe = ProbeLib()
    ops = eng.ops
    stream = lambda: C.c_void_p(torch.cuda.current_stream(ops.device).cuda_stream)  # noqa: E731
    captured, in_step = [], []

    def hook(d, t):
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        probe.getcost_loads(d, stream())
        en.record()
        in_step.append((st, en))
        keep = {k: (v.clone() if (v is not None and k in ("rt", "inv_depth", "confidence", "view_w", "disp_min", "disp_max")) else v) for k, v in t.items()}
        keep["out_cost"], keep["out_samples"] = torch.empty_like(t["out_cost"]), torch.empty_like(t["out_samples"])
        d2 = _lib.GetCostDesc()
        C.memmove(C.byref(d2), C.byref(d), C.sizeof(d))
        for k in ("rt", "inv_depth", "confidence", "view_w", "disp_min", "disp_max", "out_cost", "out_samples"):
            setattr(d2, k, None if keep[k] is None else keep[k].data_ptr())
        captured.append((d2, keep))

    with torch.no_grad():
        ops.getcost_hook = hook
        try:
            model(imgs, proj, dv)
            torch.cuda.synchronize()
        finally:
            ops.getcost_hook = None
    in_step_us = [round(s_.elapsed_time(e_) * 1e3, 2) for s_, e_ in in_step]
    iso = []
    for d2, keep in captured:
        prod = lambda: ops.lib.call("dmvs_getcost_quad_f32", C.byref(d2), stream())  # noqa: E731
        prb = lambda: probe.getcost_loads(d2, stream())  # noqa: E731
        for _ in range(12):
            prod()
        torch.cuda.synchronize()
        pu, qu = [], []
        for _ in range(2):
            pu.append(_event_us(prod, 8))
            qu.append(_event_us(prb, 8))
        iso.append({"product_us": round(sum(pu) / len(pu), 2), "probe_us": round(sum(qu) / len(qu), 2)})
    del captured
    # ---- calibration: plain random 128-byte-line gather over GetCost's source footprint (B x S views of h2 x w2 texels of 128 bytes)
    gather = None
    try:
        n_lines = int(alg_bytes["src_texels"])
        table = torch.zeros(n_lines * 128, dtype=torch.uint8, device=ops.device)
        rows = {}
        for name, mode, quads, lpq, win in (("every_line_once", 0, n_lines // 8, 8, 0), ("uniform_8_per_quad", 1, n_lines, 8, 0),
                                            ("band48_8_per_quad", 2, n_lines, 8, 48)):
            fn = lambda: probe.random_line_gather(table, n_lines, quads, lpq, mode, win, 12345, stream())  # noqa: E731
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            us = _event_us(fn, 6)
            req_bytes = quads * lpq * 128
            rows[name] = {"us": round(us, 2), "lines_requested": quads * lpq, "requested_GBs": round(req_bytes / us / 1e3, 1)}
        rows["every_line_once"]["hbm_frac_of_peak"] = round(rows["every_line_once"]["requested_GBs"] / HBM_PEAK_GBS, 4)
        gather = {"table_bytes": n_lines * 128, "request_shape": "two 64-byte quad-coalesced pieces per 128-byte line, 2 lines in flight per lane, 8 waves per SIMD, no arithmetic", **rows}
        del table
    except Exception as e:      # the calibration must never cost the headline line
        gather = {"error": repr(e)[:200]}
    gate_us = alg_bytes["getcost"] / (0.60 * HBM_PEAK_GBS * 1e9) * 1e6
    avg = lambda xs: round(sum(xs) / max(1, len(xs)), 2)  # noqa: E731
    return {"measured_by": "this process", "library": "diffmvs_amd/libdmvs_probe.so (csrc/probe/getcost_probe.hip: getcost_quad_body<QuadLoadsOnly>)",
            "gate_0p60_us": round(gate_us, 1),
            "in_step_probe_us": in_step_us, "in_step_probe_avg_us": avg(in_step_us),
            "isolated_warm": iso, "isolated_probe_avg_us": avg([r["probe_us"] for r in iso]), "isolated_product_avg_us": avg([r["product_us"] for r in iso]),
            "random_line_gather": gather}


def batch_sweep(model, make_batch, batches=(1, 2, 4, 8, 16, 32, 64), iters=6):
    """Untimed side measurement: ms per depth map against the batch size (eager launch sequence; up to batch 8 also through the
    captured HIP graph of the same forward -- the reference's harness runs batch 1, test.py:101-127)."""
    out = {}
    was = model.hip_graphs
    for B in batches:
        imgs, proj, dv = make_batch(B)
        row = {}
        for graphs in ((False, True) if B <= 8 else (False,)):
            model.hip_graphs = graphs
            with torch.no_grad():
                for _ in range(2):
                    model(imgs, proj, dv)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(iters):
                    model(imgs, proj, dv)
                torch.cuda.synchronize()
            row["graph_ms_per_map" if graphs else "eager_ms_per_map"] = round((time.perf_counter() - t0) / iters / B * 1e3, 4)
        out[str(B)] = row
    model.hip_graphs = was
    return out


def cpu_baseline(a):
    """oracle/diffmvs_oracle.py (the CPU restatement pinned to the reference) on the host cores."""
    from oracle import diffmvs_oracle as O
    from models import CasDiffMVS
    O.use_grid_sample_warp(True)       # the warp through F.grid_sample like the reference (the spelled-out checker is 1.3x slower)
    cores = max(1, min(len(os.sched_getaffinity(0)), a.cpu_threads))
    torch.set_num_threads(cores)
    args = synth.make_args("diffmvs", numdepth_initial=48)
    sd = synth.synth_state_dict(CasDiffMVS(args, test=True).state_dict(), 123)
    ci, cp, cd = synth.synth_inputs(a.height, a.width, a.src_views, B=1, seed=100)
    src = synth.NoiseSource(0)
    with torch.no_grad():
        O.forward(sd, args, ci, cp, cd, noise_fn=lambda shape: src(shape, "cpu"))      # warm-up
        t0 = time.perf_counter()
        n = 0
        while n < a.cpu_forwards and time.perf_counter() - t0 < 25.0:
            O.forward(sd, args, ci, cp, cd, noise_fn=lambda shape: src(shape, "cpu"))
            n += 1
        ct = time.perf_counter() - t0
    print(json.dumps({"value": round(n / ct, 4), "unit": "depth-maps/s", "cores": cores, "kind": "port",
                      "sample": f"{n} forwards of the same workload at batch 1 after 1 warm-up, torch CPU backend (mkldnn convs, "
                                f"F.grid_sample warp as in the reference) with {cores} of the box's "
                                f"{len(os.sched_getaffinity(0))} hardware threads; measured in the build container on 8 threads: "
                                f"the imported reference 0.86 s/map, this port 0.94 s/map (9 % slower)"}), flush=True)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: one worker process per GPU (LOCAL_RANK = GPU index), the same
    environment contract torch.distributed.run provides, rendezvous on 127.0.0.1.  Rank 0's JSON line is this process'
    output; any failing rank fails the run."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=(None if r == 0 else subprocess.DEVNULL)))
    rcs = [p.wait() for p in procs]
    if any(rcs):
        raise SystemExit(f"bench.py: worker exit codes {rcs}")


def timed_steps(step, steps, warmup, barrier):
    """W untimed + exactly K timed steps bracketed by barrier + device sync on both sides -> elapsed seconds"""
    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    return time.perf_counter() - t0


def stub_main(a):
    """The N-rank plumbing of EVERY line this file can print (env contract, process group, barrier-bracketed timing, max over ranks, one
    JSON line from rank 0) with a stand-in for the model: used by the CPU tests over gloo, never a measurement.  Per --config the stub goes
    through what that line's real main does between processes: cfg2 / cfg3 reference views per rank and no collective; cfg5 / --scene-mode
    scenes dealt to the ranks by shard_scenes; cfg4 ONE all-reduce (SUM) of a flat fp32 bucket of CasDiffMVS' gradient size per step, timed."""
    from diffmvs_amd import shard
    rank, world, local = shard.env_rank_world()
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    td = shard.init_distributed(a.backend) if world > 1 else None
    x = torch.randn(64, 64)
    kind = "scene" if a.scene_mode else a.config
    extra, units_per_step = {}, a.batch
    bucket, ar_s = None, []
    if kind == "cfg4":
        bucket = torch.full((925435,), float(rank + 1))          # CasDiffMVS: 925 435 trainable parameters
        extra = {"allreduce_bytes": bucket.numel() * 4}
    elif kind in ("cfg5", "scene"):
        scenes = ["scene%02d" % i for i in range(2 * world)]
        mine = shard.shard_scenes(scenes, rank, world)
        units_per_step = len(mine) * a.batch
        extra = {"scene_sharding": {"scenes_total": len(scenes), "scenes_of_rank0": mine if rank == 0 else None,
                                    "sharding": "shard_scenes: scene i -> rank i % world, no collective"}}

    def step():
        time.sleep(0.002 * (rank + 1))          # rank-dependent duration: the slowest rank must set the time
        y = x @ x
        if bucket is not None and td is not None:
            t0 = time.perf_counter()
            td.all_reduce(bucket, op=td.ReduceOp.SUM)        # the training path's one collective
            ar_s.append(time.perf_counter() - t0)
            bucket.mul_(1.0 / world)
        return y

    elapsed = timed_steps(step, a.steps, a.warmup, (lambda: td.barrier()) if td else (lambda: None))
    whole = shard.barrier_and_max(elapsed)
    total_units = shard.total_items(units_per_step) if kind in ("cfg5", "scene") else units_per_step * world
    if bucket is not None:
        extra["allreduce_ms_per_step"] = round(1e3 * sum(ar_s[-a.steps:]) / max(1, len(ar_s[-a.steps:])), 4) if ar_s else None
        # every rank holds the same averaged bucket after each step (the mean of 1..world stays fixed under repeated averaging)
        extra["bucket_value"] = float(bucket[0])
    if rank == 0:
        print(json.dumps({"metric": "stub", "stub_of": kind, "value": round(total_units * a.steps / whole, 3), "n_gpus": world,
                          "world_size_seen": td.get_world_size() if td else 1, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": round(whole / a.steps * 1e3, 4), "rank0_ms_per_step": round(elapsed / a.steps * 1e3, 4),
                          "multi_gpu": "unmeasured on hardware (this is the CPU stand-in of the N-rank plumbing)", **extra}),
              flush=True)
    if td:
        td.destroy_process_group()


def kernel_source_hash():
    """sha256 (first 16 hex digits) of the warp kernels' source: PMC traffic files record it, a stale file is refused"""
    import hashlib
    h = hashlib.sha256()
    for f in ("warp_quad.hip", "warp_quad_core.h", "dmvs_common.h"):
        with open(os.path.join(ROOT, "diffmvs_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def train_main(a):
    """BASELINE.json configs[3]: CasDiffMVS training step (forward(train) -> compute_inverse_loss -> backward -> ONE all-reduce of
    the flat gradient bucket over RCCL -> clip(2.0) -> AdamW; reference train.py:179-209, :351), 768x576, 9 views, batch 4 per
    GPU, fp32, through Trainer.train_sample.  A second line, never the headline: samples/s over all ranks, ms inside the
    all-reduce, the RCCL world size seen."""
    from diffmvs_amd import shard
    from diffmvs_amd.trainer import Trainer
    from models import CasDiffMVS
    rank, world, local = shard.env_rank_world()
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    td = shard.init_distributed(a.backend, dev) if world > 1 else None
    H, W, S = 576, 768, 8
    B = a.batch if "--batch" in sys.argv else 4
    args = synth.make_args("casdiffmvs", numdepth_initial=48)
    model = CasDiffMVS(args, test=False)
    model.load_state_dict(synth.synth_state_dict(model.state_dict(), 123), strict=True)
    model.to(dev).train()
    tr = Trainer(model, args, total_steps=1000000)
    tr.time_allreduce = True
    imgs, proj, dv, gt, mask = synth.synth_inputs(H, W, S, B=B, seed=300 + rank, with_gt=True)
    sample = {"imgs": [i.to(dev) for i in imgs], "proj_matrices": {k: v.to(dev) for k, v in proj.items()}, "depth_values": dv.to(dev),
              "depth": {k: v.to(dev) for k, v in gt.items()}, "mask": {k: v.to(dev) for k, v in mask.items()}}

    def barrier():
        if td:
            td.barrier()
        torch.cuda.synchronize()

    last = {}

    def step():
        last["loss"] = tr.train_sample(sample)[0]

    for _ in range(a.warmup):
        step()
    tr.allreduce_events.clear()
    elapsed = timed_steps(step, a.steps, 0, barrier)
    elapsed = shard.barrier_and_max(elapsed, dev)
    ar_ms = [s_.elapsed_time(e_) for s_, e_ in tr.allreduce_events]
    # one extra, untimed step with an event pair around the backward's two dominant kernel families (kept out of the timed region)
    from diffmvs_amd.ops import Ops
    ops = Ops.for_device(dev)
    ops.timers = {"dmvs_getcost_bwd_f32": [], "dmvs_conv2d_wgrad_f32": []}
    step()
    torch.cuda.synchronize()
    tm, ops.timers = ops.timers, None
    gb_ms = [s_.elapsed_time(e_) for s_, e_ in tm["dmvs_getcost_bwd_f32"]]
    wg_ms = [s_.elapsed_time(e_) for s_, e_ in tm["dmvs_conv2d_wgrad_f32"]]
    gb_bytes, wg_flops = sum(tm.get("_getcost_bwd_bytes", [])), sum(tm.get("_wgrad_flops", []))
    roof = {"roofline_getcost_bwd": {"kernel": "getcost_bwd_kernel<*> (gradient scatter into the source features, hardware fp32 atomics)", "bound": "hbm",
                                     "achieved": round(gb_bytes / (sum(gb_ms) * 1e-3) / 1e9, 2) if gb_ms else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": round(gb_bytes / (sum(gb_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if gb_ms else None,
                                     "launches": len(gb_ms), "ms_per_step": round(sum(gb_ms), 3),
                                     "note": "bound in practice by the L2's fp32 atomic rate (one atomic per tap and channel), not by HBM: DESIGN.md 3.2"},
            "roofline_conv2d_wgrad": {"kernel": "conv2d_wgrad_kernel<*> (all weight-gradient launches of the step)", "bound": "mfma",
                                      "achieved": round(wg_flops / (sum(wg_ms) * 1e-3) / 1e12, 2) if wg_ms else None, "peak": FP32_MFMA_PEAK_TFS, "unit": "TFLOP/s",
                                      "frac": round(wg_flops / (sum(wg_ms) * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFS, 4) if wg_ms else None,
                                      "launches": len(wg_ms), "ms_per_step": round(sum(wg_ms), 3)}}
    if rank == 0:
        print(json.dumps({
            "metric": "training samples/sec (CasDiffMVS 768x576, 9 views)", "value": round(B * world * a.steps / elapsed, 3), "unit": "samples/s",
            "n_gpus": world, "rccl_world_size": (td.get_world_size() if td else 1), "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (rendered slanted-plane scenes with ground-truth depth)",
            "config": {"workload": "CasDiffMVS training step 768x576, 8 src views, batch 4 per GPU, fp32 (BASELINE.json configs[3]; not the headline)",
                       "batch_per_gpu": B, "parallelism": f"data parallel x{world}: one all-reduce (SUM) of the {tr.flat.numel * 4 / 1e6:.2f} MB flat fp32 gradient bucket per step",
                       "weights": "seeded random init"},
            "allreduce_ms_per_step": round(sum(ar_ms) / max(1, len(ar_ms)), 4) if ar_ms else None,
            "allreduce_bytes": tr.flat.numel * 4, "loss": float(last["loss"]), **roof,
            "max_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}), flush=True)
    if td:
        td.destroy_process_group()


def scene_main(a):
    """Second line, never the headline: the SAME network and geometry as the headline (BASELINE.json configs[1]) evaluated the way a
    dataset is -- whole scenes (49 views each, DTU-shaped: every view is the reference once and a source of ~5 neighbours; the pair table
    plays datasets' pair.txt) -- with every image pushed through FeatureNet ONCE per scene (SceneFeatureStore, features resident in HBM)
    instead of once per sample it occurs in (reference test.py:92-127; the headline line does that too: 576 images per 96 maps).
    A step = `--scenes-per-step` scenes: feature store of their images + the forward of all their reference views.  The same scenes
    through the per-sample forward are timed next to it (`per_sample`)."""
    from diffmvs_amd import shard
    from models import CasDiffMVS
    rank, world, local = shard.env_rank_world()
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    td = shard.init_distributed(a.backend, dev) if world > 1 else None
    variant, nd_initial, prec, arith = "diffmvs", 48, a.precision or "fp32", a.conv_arith or "split"
    if a.config == "cfg3":      # the same comparison on BASELINE.json configs[2]: CasDiffMVS 1152x864, 7 source views, bf16 storage + arithmetic
        variant, prec, arith = "casdiffmvs", a.precision or "bf16", a.conv_arith or "bf16"
        a.height, a.width, a.src_views = 864, 1152, 7
        if "--scenes-per-step" not in sys.argv:
            a.scenes_per_step = 1
    H, W, S, NV = a.height, a.width, a.src_views, a.scene_views
    args = synth.make_args(variant, numdepth_initial=nd_initial, precision=prec, conv_arith=arith)
    model = CasDiffMVS(args, test=True).eval()
    model.load_state_dict(synth.synth_state_dict(model.state_dict(), 123))
    model = model.to(dev)
    scene = synth.synth_scene(H, W, n_views=NV, n_src=S, seed=200 + rank)
    imgs, proj, dv, view_ids = synth.scene_batch(scene, list(range(NV)))
    k = a.scenes_per_step
    # further scenes of a step: the rendered one with every image shifted horizontally (other content, same cameras), like the headline's batch
    images = torch.cat([torch.roll(scene["images"], shifts=41 * i, dims=-1) for i in range(k)], 0).to(dev)
    ids = torch.cat([view_ids + NV * i for i in range(k)], 0)
    proj_d = {kk: torch.cat([p] * k, 0).to(dev) for kk, p in proj.items()}
    dv_d = torch.cat([dv] * k, 0).to(dev)
    ref_imgs = images[ids[:, 0].to(dev)].contiguous()
    per_sample_imgs = [images[ids[:, v].to(dev)].contiguous() for v in range(S + 1)]

    def barrier():
        if td:
            td.barrier()
        torch.cuda.synchronize()

    def step():
        store = model.scene_features(images)
        return model([ref_imgs], proj_d, dv_d, feats=store.gather(ids))

    def step_per_sample():
        return model(per_sample_imgs, proj_d, dv_d)

    with torch.no_grad():
        elapsed = timed_steps(step, a.steps, a.warmup, barrier)
        elapsed = shard.barrier_and_max(elapsed, dev)
        ps = timed_steps(step_per_sample, max(2, a.steps // 4), 1, barrier) / max(2, a.steps // 4)
    maps = NV * k
    if rank == 0:
        print(json.dumps({
            "metric": f"depth-maps/sec ({W}x{H}, {S} src views), scene mode", "value": round(maps * a.steps * world / elapsed, 3), "unit": "depth-maps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if arith == "bf16" else "f32",
            "data": f"synthetic scenes of {NV} views (one rendered per rank, further scenes = horizontally shifted copies), pair table = {S} nearest grid neighbours",
            "config": {"workload": f"{'DiffMVS' if variant == 'diffmvs' else 'CasDiffMVS'} eval {W}x{H}, {S} src views, numdepth_initial={nd_initial}, {prec} feature storage, "
                                   f"{arith} conv arithmetic -- NOT THE HEADLINE: whole scenes, each image through FeatureNet once per scene (feature store resident in HBM)",
                       "scenes_per_step": k, "views_per_scene": NV, "ref_views_per_gpu_per_step": maps, "images_through_featurenet_per_step": maps,
                       "images_through_featurenet_per_step_per_sample_mode": maps * (S + 1),
                       "parallelism": f"scene sharding x{world}, no collective", "weights": "seeded random init"},
            "per_sample": {"ms_per_step": round(ps * 1e3, 4), "depth_maps_per_s": round(maps / ps, 3),
                           "note": "the same scenes, FeatureNet on all V images of every reference view (reference test.py:92-127; the headline's order of work)"},
            "speedup_vs_per_sample": round(ps / (elapsed / a.steps), 4),
            "max_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}), flush=True)
    if td:
        td.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("DMVS_BENCH_BATCH", "96")),
                    help="reference views per GPU per step (576 images at the default 96: ~30 GB of the 288 GB; measured on the "
                         "MI355X: 16 -> 981 depth-maps/s, 32 -> 1054, 64 -> 1109, 96 -> 1120: larger tile counts fill the 256 CUs' MFMA pipes better)")
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--src-views", type=int, default=5)
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg3", "cfg4", "cfg5"],
                    help="cfg2 = BASELINE.json configs[1] (the metric's configuration, default); cfg3 = configs[2]: CasDiffMVS 1152x864, "
                         "7 source views, bf16 feature storage; cfg4 = configs[3]: the CasDiffMVS training step, data parallel; cfg5 = "
                         "configs[4]: CasDiffMVS 1920x1056, 11 source views, numdepth_initial 96, fp16 feature storage, scenes sharded "
                         "over the ranks (second lines, not the headline)")
    ap.add_argument("--precision", default=None, choices=["fp32", "bf16", "fp16"], help="feature storage precision (default: the config's)")
    ap.add_argument("--conv-arith", default=None, choices=["fp32", "bf16", "split"],
                    help="matrix arithmetic of the 2-D convolutions (default: fp32 for cfg2 -- the headline is an fp32 number -- and "
                         "bf16 for cfg3, whose BASELINE.json entry is a bf16 configuration)")
    ap.add_argument("--scene-mode", action="store_true",
                    help="second line (cfg2 geometry): whole 49-view scenes, every image through FeatureNet once per scene (see scene_main)")
    ap.add_argument("--scenes-per-step", type=int, default=2)
    ap.add_argument("--scene-views", type=int, default=49)
    ap.add_argument("--graphs", action="store_true", help="run the timed steps through the captured HIP graph of the forward")
    ap.add_argument("--no-batch-sweep", action="store_true")
    ap.add_argument("--conv-table", action="store_true", help="print the per-shape table of the step's conv2d launches to stderr")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probe", action="store_true", help="skip the GetCost ceiling-probe / random-line-gather legs (untimed; ~10 s)")
    ap.add_argument("--cpu-forwards", type=int, default=8)
    ap.add_argument("--cpu-threads", type=int, default=32, help="threads of the CPU baseline leg (more than ~32 is slower at batch 1)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--backend", default="nccl", help=argparse.SUPPRESS)        # "gloo" + --stub-model: the CPU test of the N>1 plumbing
    ap.add_argument("--stub-model", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.cpu_baseline_only:
        return cpu_baseline(a)
    if a.gpus > 1 and "RANK" not in os.environ:
        return spawn_ranks(a.gpus)              # plain `python bench.py --gpus N`: start the N ranks ourselves
    if a.stub_model:
        return stub_main(a)
    if a.config == "cfg4":
        return train_main(a)
    if a.scene_mode:
        return scene_main(a)

    from diffmvs_amd import shard
    rank, world, local = shard.env_rank_world()
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    dist = world > 1
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if dist:
        td = shard.init_distributed(a.backend, dev)       # RCCL: rendezvous, barrier and one 8-byte max-reduce only

    from models import CasDiffMVS
    variant = "diffmvs"
    if a.config == "cfg3":
        variant = "casdiffmvs"
        a.height, a.width, a.src_views = 864, 1152, 7
        if "DMVS_BENCH_BATCH" not in os.environ and "--batch" not in sys.argv:
            a.batch = 4
        a.precision = a.precision or "bf16"
        a.conv_arith = a.conv_arith or "bf16"
    nd_initial = 48
    scene_note = None
    if a.config == "cfg5":
        # Tanks&Temples-sized inference (reference test.py:92-127: a per-scene loop): scenes, not reference views, are the unit
        # that shards -- shard_scenes() keeps a scene's depth maps on one rank for the fusion step that follows
        variant, nd_initial = "casdiffmvs", 96
        a.height, a.width, a.src_views = 1056, 1920, 11
        if "DMVS_BENCH_BATCH" not in os.environ and "--batch" not in sys.argv:
            a.batch = 2
        a.precision = a.precision or "fp16"
        scenes = ["scene%02d" % i for i in range(2 * world)]
        mine = shard.shard_scenes(scenes, rank, world)
        scene_note = {"scenes_total": len(scenes), "scenes_of_rank0": mine, "sharding": "shard_scenes: scene i -> rank i % world, no collective"}
    prec = a.precision or "fp32"
    arith = a.conv_arith or "split"
    args = synth.make_args(variant, numdepth_initial=nd_initial, precision=prec, conv_arith=arith)
    model = CasDiffMVS(args, test=True).eval()
    sd = synth.synth_state_dict(model.state_dict(), 123)
    model.load_state_dict(sd)
    model = model.to(dev)
    H, W, S, B = a.height, a.width, a.src_views, a.batch
    # the synthetic scenes are rendered on the host (~1 s per reference view with its source views): at most 16 are rendered
    # per rank; further batch items re-use those images shifted horizontally by a different amount each (so that no two items
    # share content and hence depth maps / gather patterns) with the cameras of the scene they were derived from.  (The
    # generator's cameras rotate further with the batch index -- meant for small test batches: beyond ~16 the source views look
    # away from the scene, most samples fall outside the images and the warp kernels look faster than they are.)
    uniq = min(B, 16)
    imgs, proj, dv = synth.synth_inputs(H, W, S, B=uniq, seed=100 + rank)
    base = ([i.to(dev) for i in imgs], {k: v.to(dev) for k, v in proj.items()}, dv.to(dev))

    def make_batch(nb):
        idx = torch.arange(nb) % uniq
        pj, dvs = {k: p[idx] for k, p in proj.items()}, dv[idx]
        ims = []
        for v in range(S + 1):
            t = torch.cat([base[0][v]] * ((nb + uniq - 1) // uniq), 0)[:nb].contiguous()
            for i in range(uniq, nb):
                t[i] = torch.roll(t[i], shifts=37 * (i // uniq) + 3 * (i % uniq), dims=-1)
            ims.append(t)
        return ims, {k: p.to(dev) for k, p in pj.items()}, dvs.to(dev)

    imgs, proj, dv = make_batch(B)
    eng = model.engine()
    model.hip_graphs = bool(a.graphs)

    def barrier():
        if dist:
            td.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(a.warmup):
            model(imgs, proj, dv)
        eng.ops.timers = {"dmvs_getcost_quad_f32": [], "dmvs_warp_corr_init_quad_f32": []}
        elapsed = timed_steps(lambda: model(imgs, proj, dv), a.steps, 0, barrier)
    timers, eng.ops.timers = eng.ops.timers, None
    elapsed = shard.barrier_and_max(elapsed, dev)      # whole-job time = slowest rank
    model.hip_graphs = False                           # the per-kernel event legs below need the eager launch sequence
    if a.graphs:      # no per-launch events inside a graph replay: one extra eager step for the roofline legs
        with torch.no_grad():
            eng.ops.timers = {k: [] for k in timers}
            model(imgs, proj, dv)
            torch.cuda.synchronize()
        timers, eng.ops.timers = eng.ops.timers, None
    # one extra, untimed step with an event pair around every conv2d launch (kept out of the timed region: ~200 launches)
    with torch.no_grad():
        eng.ops.timers = {"dmvs_conv2d_f32": []}
        t1 = time.perf_counter()
        model(imgs, proj, dv)
        torch.cuda.synchronize()
        conv_step_s = time.perf_counter() - t1
    timers.update(eng.ops.timers)
    eng.ops.timers = None
    # per-shape table of the step's conv2d launches: TFLOP/s against the fp32-MFMA peak AND algorithmic GB/s against the HBM peak; a
    # row's binding resource is the larger of the two fractions (a 1x1 layer moves 4 bytes per 2 x cout flops: HBM; a 3x3 layer at 32+
    # channels: the matrix pipe).  `sum` = (t_mfma + t_hbm) / t: near 1.0 the launch pays its matrix time PLUS its memory time (no overlap).
    conv_rows = []
    if timers.get("_conv2d_shape"):
        by = {}
        nb = timers.get("_conv2d_bytes") or [0.0] * len(timers["_conv2d_shape"])
        for (st_, en_), fl, shp, bts in zip(timers["dmvs_conv2d_f32"], timers["_conv2d_flops"], timers["_conv2d_shape"], nb):
            e = by.setdefault(shp, [0, 0.0, 0.0, 0.0])
            e[0] += 1; e[1] += st_.elapsed_time(en_); e[2] += fl; e[3] += bts
        for shp, (n, ms, fl, bts) in sorted(by.items(), key=lambda kv: -kv[1][1]):
            f_m, f_h = fl / ms / 1e9 / FP32_MFMA_PEAK_TFS, bts / ms / 1e6 / HBM_PEAK_GBS
            conv_rows.append({"shape": list(shp), "launches": n, "ms": round(ms, 3), "tflops": round(fl / ms / 1e9, 1), "frac_mfma": round(f_m, 3),
                              "gbs": round(bts / ms / 1e6, 0), "frac_hbm": round(f_h, 3), "bound": "mfma" if f_m >= f_h else "hbm",
                              "frac": round(max(f_m, f_h), 3), "sum": round(f_m + f_h, 3)})
    if a.conv_table and rank == 0:
        print("B cin cout kh kw s Hout Wout mode gated | launches  ms/step  TFLOP/s  frac_mfma  GB/s  frac_hbm  bound  sum", file=sys.stderr)
        for r in conv_rows:
            print(*r["shape"], "|", r["launches"], r["ms"], r["tflops"], r["frac_mfma"], int(r["gbs"]), r["frac_hbm"], r["bound"], r["sum"], file=sys.stderr)
    scene = scene_geometry_getcost(eng.ops, B, H, W, S, args.CostNum[1]) if (rank == 0 and a.config == "cfg2") else None

    sweep = batch_sweep(model, make_batch) if (rank == 0 and world == 1 and not a.no_batch_sweep and a.config == "cfg2") else None
    maps = B * a.steps * world
    value = maps / elapsed
    gc_ms = [s.elapsed_time(e) for s, e in timers["dmvs_getcost_quad_f32"]]
    wi_ms = [s.elapsed_time(e) for s, e in timers["dmvs_warp_corr_init_quad_f32"]]
    gc_avg_s = sum(gc_ms) / max(1, len(gc_ms)) * 1e-3
    # the launches of a step in order = the GRU iterations of the diffusion stage(s): iteration 1 samples around pure noise
    # (scale * randn, update.py:472), iterations 2.. around the network's own estimate
    per_step = len(gc_ms) // max(1, a.steps) if not a.graphs else len(gc_ms)
    gc_bytes = timers.get("_getcost_bytes") or []
    by_iter = []
    for it in range(per_step if per_step and len(gc_ms) % per_step == 0 else 0):
        ms_it = gc_ms[it::per_step]
        by_it = gc_bytes[it::per_step] if len(gc_bytes) == len(gc_ms) else []
        avg_us = sum(ms_it) / len(ms_it) * 1e3
        bts = sum(by_it) / len(by_it) if by_it else None
        by_iter.append({"launch_in_step": it + 1, "avg_launch_us": round(avg_us, 2), "launches": len(ms_it),
                        "frac": round(bts / (avg_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if bts else None})
    h2, w2 = H // 4, W // 4
    alg = getcost_algorithmic_bytes(B, 32, S, args.CostNum[1], 4, h2, w2)
    h1, w1 = H // 8, W // 8
    alg_init = 4 * B * h1 * w1 * (48 + S * 48 + S * 4 * 48)      # ref + src + per-view volumes out
    if timers.get("_getcost_bytes"):          # quad kernels: bytes recorded per launch (stage-2 and stage-3 shapes, feature element size)
        alg = sum(timers["_getcost_bytes"]) / len(timers["_getcost_bytes"])
    if timers.get("_warp_init_bytes"):
        alg_init = sum(timers["_warp_init_bytes"]) / len(timers["_warp_init_bytes"])
    achieved = alg / gc_avg_s / 1e9 if gc_avg_s > 0 else 0.0
    wi_avg_s = sum(wi_ms) / max(1, len(wi_ms)) * 1e-3
    cv_s = sum(s.elapsed_time(e) for s, e in timers["dmvs_conv2d_f32"]) * 1e-3
    cv_flops = sum(timers.get("_conv2d_flops", []))

    # HBM traffic per getcost launch from the PMC passes (rocprofv3 cannot run inside the timed process); only
    # quoted when the committed measurement was taken at this batch size
    traffic, traffic_note = None, None
    tj = os.path.join(ROOT, "profiles", "r6_getcost_traffic.json")
    if os.path.exists(tj):
        with open(tj) as f:
            tinfo = json.load(f)
        if tinfo.get("kernel_source_sha") != kernel_source_hash():
            traffic_note = f"{os.path.relpath(tj, ROOT)} was measured on another version of warp_quad.hip: not quoted"
        elif tinfo.get("batch") == B and (H, W, S) == (512, 640, 5):
            traffic = tinfo["traffic_bytes_per_launch"]
            traffic_note = (os.path.relpath(tj, ROOT) + " (rocprofv3 PMC passes of this command on this kernel source, counters scaled by the "
                            "factors of profiles/r3_traffic_calibration.json; not measured by this process)")

    # the memory-system ceiling of GetCost's address stream, measured by this process in untimed legs (see getcost_ceiling_probe)
    ceiling = None
    if rank == 0 and world == 1 and a.config == "cfg2" and not a.no_probe:
        try:
            ceiling = getcost_ceiling_probe(model, eng, imgs, proj, dv, {"getcost": alg, "src_texels": B * S * h2 * w2})
            ceiling["product_in_step_avg_us"] = round(gc_avg_s * 1e6, 2)
            ceiling["product_over_probe_in_step"] = round(ceiling["in_step_probe_avg_us"] / (gc_avg_s * 1e6), 3) if gc_avg_s > 0 else None
            ceiling["product_over_probe_isolated"] = round(ceiling["isolated_probe_avg_us"] / max(ceiling["isolated_product_avg_us"], 1e-9), 3)
            ceiling["probe_frac_of_hbm_peak_in_step"] = round(alg / (ceiling["in_step_probe_avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
            ceiling["reading"] = ("in_step_probe_avg_us above gate_0p60_us = the product kernel's own line-request stream with NO arithmetic already takes longer "
                                  "than the 0.60 mark allows; product_over_probe_* = how close the product runs to that ceiling")
        except Exception as e:      # an absent / failing probe library must never cost the headline line
            ceiling = {"measured_by": "this process", "error": repr(e)[:300]}

    result = {
        "metric": "depth-maps/sec (640x512, 5 src views)", "value": round(value, 3), "unit": "depth-maps/s",
        "n_gpus": world, "rccl_world_size": (td.get_world_size() if dist else 1), "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic (rendered slanted-plane scenes; 16 rendered per rank, further batch items = horizontally shifted copies)",
        "config": {"workload": f"DiffMVS DTU eval {W}x{H}, {S} src views, numdepth_initial=48, 1 DDIM step, fp32",
                   "ref_views_per_gpu_per_step": B, "parallelism": f"ref-view sharding x{world}, no collective",
                   "weights": "seeded random init (no checkpoint offline)",
                   "launch": "captured HIP graph of the forward" if a.graphs else "eager launch sequence (~350 kernels per step)"},
        # the second half of BASELINE.json's metric ("DTU abs-rel vs ref") cannot be measured here: no DTU scans and no trained checkpoint in
        # the image.  What IS measured against the reference: depth maps on the reference's own inputs (goldens recorded by
        # importing it) -- tests/test_model_gpu.py, final-depth relative L1 1e-7 .. 4e-7 against the 1e-3 bar
        "dtu_abs_rel_vs_ref": None,
        "accuracy_note": "DTU abs-rel unmeasurable offline (no dataset / checkpoint); depth parity vs the reference's recorded outputs is what tests/test_model_gpu.py asserts (1e-3 bar) -- this line makes no parity claim of its own",
        "batch_sweep_ms_per_map": sweep,
        "roofline": {"kernel": "GetCost: getcost_quad_kernel<32,6> (quad per pixel, one launch for any geometry)",
                     "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "traffic_source": traffic_note,
                     "algorithmic_bytes_per_launch": alg, "avg_launch_us": round(gc_avg_s * 1e6, 2),
                     "launches_timed": len(gc_ms), "per_gru_iteration": by_iter,
                     "ceiling_probe_us": (ceiling or {}).get("in_step_probe_avg_us"), "ceiling_probe": ceiling},
        "roofline_warp_init": {"kernel": ("warp_init_quad_kernel<48> (stage-1 plane sweep, quad per pixel, texels from global memory: tune DMVS_TUNE_SWEEP_GLOBAL)"
                                          if eng.ops.tune["sweep"] else
                                          "warp_init_band_kernel<48> (stage-1 plane sweep, quad per pixel, source band of a 16x4 pixel tile staged in LDS)"), "bound": "hbm",
                               "achieved": round(alg_init / wi_avg_s / 1e9, 2) if wi_avg_s > 0 else 0.0,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(alg_init / wi_avg_s / 1e9 / HBM_PEAK_GBS, 4) if wi_avg_s > 0 else 0.0,
                               "algorithmic_bytes_per_launch": alg_init, "avg_launch_us": round(wi_avg_s * 1e6, 2)},
        "roofline_scene_geometry": scene,
        # where the step's time actually goes: all 2-D convolution launches together, priced against the fp32 arithmetic the reference does
        "roofline_conv2d": {"kernel": "conv2d_mfma_kernel<*> / conv1x1_px4_kernel<*> (all 2-D convolution launches of the step)", "bound": "mfma",
                            "achieved": round(cv_flops / cv_s / 1e12, 2) if cv_s > 0 else 0.0, "peak": FP32_MFMA_PEAK_TFS,
                            "unit": "TFLOP/s", "frac": round(cv_flops / cv_s / 1e12 / FP32_MFMA_PEAK_TFS, 4) if cv_s > 0 else 0.0,
                            "launches_timed": len(timers["dmvs_conv2d_f32"]),
                            "share_of_step_time": round(cv_s / conv_step_s, 4),
                            # the same launches against BOTH roofs: algorithmic bytes / time next to flops / time, and per layer shape the roof
                            # that binds it (the family as a whole is matrix-bound by time share; the 1x1 and <= 8-channel rows are not)
                            "hbm_achieved_GBs": round(sum(timers.get("_conv2d_bytes", [])) / cv_s / 1e9, 1) if cv_s > 0 else None,
                            "ms_by_binding_roof": {k: round(sum(r["ms"] for r in conv_rows if r["bound"] == k), 3) for k in ("mfma", "hbm")},
                            "rows": [[*r["shape"][:8], r["launches"], r["ms"], r["frac_mfma"], r["frac_hbm"], r["bound"]] for r in conv_rows[:24]],
                            "rows_columns": "B cin cout kh kw stride Hout Wout launches ms_per_step frac_of_fp32_mfma_peak frac_of_hbm_peak binding_roof (top 24 shapes by time)"},
    }

    if a.config != "cfg2":
        which = "configs[2]" if a.config == "cfg3" else "configs[4]: Tanks&Temples full resolution, scene-sharded inference"
        result["metric"] = f"depth-maps/sec ({W}x{H}, {S} src views)"
        result["config"]["workload"] = (f"CasDiffMVS eval {W}x{H}, {S} src views, numdepth_initial={nd_initial}, sampling_timesteps 0/1/1, "
                                        f"{prec} feature storage, {arith} matrix arithmetic in the 2-D convolutions (fp32 accumulation), everything else "
                                        f"fp32 (BASELINE.json {which}; not the headline configuration)")
        result["dtype"] = "bf16" if arith == "bf16" else "f32"
        result["roofline"]["kernel"] = "GetCost: getcost_quad_kernel<32,4> + <16,4> (stage 2 and stage 3 launches, bytes averaged per launch)"
        result["roofline_conv2d"] = None
        result["roofline_warp_init"]["kernel"] = result["roofline_warp_init"]["kernel"].replace("<48>", "<48> D=%d" % nd_initial)
        if scene_note:
            result["config"]["scene_sharding"] = scene_note
            result["config"]["multi_gpu"] = ("measured on %d GPU(s) in this run" % world) + ("" if world > 1 else
                                             "; --gpus N shards the scenes over N ranks (unmeasured on hardware: 1-GPU leases)")
    result["config"]["feature_storage"] = prec
    result["config"]["conv_arith"] = arith
    if arith == "bf16" and a.config == "cfg2":      # an experiment line, never the headline: say so where the driver reads it
        result["dtype"] = "bf16"
        result["config"]["workload"] += " -- NON-HEADLINE EXPERIMENT: bf16 matrix arithmetic in the 2-D convolutions"
    if arith == "split":
        # fp32 ACCURACY, not a reduced precision: every fp32 operand of the honoured multi-tap convolutions as three bf16 values, six partial products
        # per product on the bf16 matrix cores, fp32 accumulation -- as close to fp64 as the exact-fp32 kernels on this GPU (DESIGN.md 4.5;
        # tests/test_ops.py::test_conv2d_split_bf16_arithmetic, tests/test_model_gpu.py::test_cfg2_full_size_split_bf16_arithmetic)
        result["config"]["conv_arith_note"] = ("split = fp32-accurate products from bf16 triples (hi + mid + lo, six partial products down to 2^-18, fp32 "
                                               "accumulation) in the multi-tap 2-D convolutions where faster; measured against fp64 as close as the exact-fp32 "
                                               "kernels (profiles/r6_split_accuracy.txt); --conv-arith fp32 runs exact fp32 MFMAs everywhere")
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.config == "cfg2":
        # the CPU leg runs in a child process with a hard time limit so that an oversubscribed or slow
        # host can never stall the GPU measurement
        import subprocess
        try:
            cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only",
                                 "--height", str(H), "--width", str(W), "--src-views", str(S),
                                 "--cpu-forwards", str(a.cpu_forwards), "--cpu-threads", str(a.cpu_threads)],
                                capture_output=True, text=True, timeout=150)
            line = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
            result["cpu_baseline"] = json.loads(line[-1]) if line else {
                "value": None, "unit": "depth-maps/s", "cores": 0, "kind": "port",
                "sample": "cpu leg failed: " + cp.stderr[-200:]}
        except subprocess.TimeoutExpired:
            result["cpu_baseline"] = {"value": None, "unit": "depth-maps/s", "cores": 0, "kind": "port",
                                      "sample": "cpu leg exceeded its 150 s limit"}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
