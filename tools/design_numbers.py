"""Regenerate the end-of-round numbers block of DESIGN.md section 7 from the committed bench lines (profiles/r5_*.json), so that the
text quotes exactly what the files hold (tests/test_docs.py checks it):   python tools/design_numbers.py"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda n: json.load(open(os.path.join(ROOT, "profiles", n)))  # noqa: E731


def main():
    line, full = P("r5_bench_line.json"), P("r5_bench_full_line.json")
    cfg3, cfg4, cfg5, cfg5b8, cfg5b16, scene = (P("r5_bench_cfg3_line.json"), P("r5_bench_cfg4_line.json"), P("r5_bench_cfg5_line.json"),
                                                P("r5_bench_cfg5_b8_line.json"), P("r5_bench_cfg5_b16_line.json"), P("r5_bench_scene_line.json"))
    sc3 = P("r5_bench_scene_cfg3_line.json")
    rf, it = line["roofline"], line["roofline"]["per_gru_iteration"]
    sw = full["batch_sweep_ms_per_map"]
    cb = full["cpu_baseline"]
    tr = rf["traffic"]
    text = f"""End-of-round numbers (1 x MI355X, `profiles/r5_bench_line.json`: the default `python bench.py` on the HEAD kernels -- the round's last GPU sessions are
`tools/sessions/r5_final2.sh` (the whole GPU suite with `-x` green, `smoke()`, this command at 1310 depth-maps/s, the rocprofv3 / PMC / probe
passes) and `r5_final_b.sh` (`smoke()` + this line again, once the traffic and probe files measured on the HEAD kernel source were committed
so that the line quotes them; no `csrc/` or `include/` change in between or after): **{round(line['value'])} depth-maps/s** at the default 96 reference views per
step ({line['ms_per_step']:.1f} ms per step; the same command on the seven boxes of this round's sessions: 73.3 .. 75.0 ms, i.e. 1280 .. 1310 depth-maps/s --
box-to-box variation is larger than the round's kernel gains; round 4: 1294 / 1299 on the driver's run, round 3: 1200, round 2: 1118, round 1: 848 at B=16).
`roofline.frac` {rf['frac']:.2f} ({rf['avg_launch_us']:.0f} us per launch; first GRU iteration {it[0]['frac']:.2f}, iterations 2-4 {min(x['frac'] for x in it[1:]):.3f}-{max(x['frac'] for x in it[1:]):.3f};
`traffic` {('%.2f GB' % (tr / 1e9)) if tr else 'n/a'} per launch against {rf['algorithmic_bytes_per_launch'] / 1e9:.2f} GB algorithmic: no wasted re-reads; `ceiling_probe_us` {rf['ceiling_probe_us']}: the
kernel's own address stream without arithmetic, section 3.1 -- the 0.60 mark would be 372 us), `roofline_scene_geometry.frac` {line['roofline_scene_geometry']['frac']:.2f},
`roofline_warp_init.frac` {line['roofline_warp_init']['frac']:.2f}, `roofline_conv2d.frac` {line['roofline_conv2d']['frac']:.2f} over {100 * line['roofline_conv2d']['share_of_step_time']:.0f} % of the step.
The FULL default run (`profiles/r5_bench_full_line.json`: 20 steps, batch sweep, `cpu_baseline`): {round(full['value'])} depth-maps/s ({full['ms_per_step']:.1f} ms);
by batch {sw['16']['eager_ms_per_map']:.2f} ms per map at B=16, {sw['32']['eager_ms_per_map']:.2f} at 32, {sw['64']['eager_ms_per_map']:.2f} at 64; CPU baseline {cb['value']:.2f} maps/s (oracle port, {cb['cores']}
threads) => ~{round(full['value'] / cb['value'] / 10) * 10}x -- a reported baseline, not a kernel-quality figure.  The profiled run (`profiles/r5_bench_b96_profiled_line.json` +
`r5_bench_b96_kernel_stats.csv`) agrees with its event timing on the plane sweep and on GetCost (`tests/test_docs.py`).  Kernel-time split per
B=96 forward (`tools/kernel_families.py profiles/r5_bench_b96_kernel_stats.csv 8`): conv2d 53.2 ms, conv3d 9.10 (round 4: 9.77 -- the paired
kernels of 4.1), fused stem 4.56, GetCost ~2.3 ms in the timed steps, plane sweep 1.06, GroupNorm apply 1.82, everything else 2.4.

The gates of the previous review, as measured.  GREEN at HEAD: the GPU suite in its new order, the reproducibility assertion on all outputs
of cfg2 / cfg3 / cfg5 three more runs each (5.1).  Opt-ins: timed, two made default, the rest deleted (4.1, `profiles/r5_optins.jsonl`).
GetCost: the ceiling probe is ABOVE the 0.60 mark, stated on the bench line -- the gate is closed as "at the measured ceiling", not
met.  Scene mode: built, bit-identical, 1.48x (below).  NOT met: conv2d >= 0.70 / step <= 70 ms ({line['roofline_conv2d']['frac']:.2f} / {line['ms_per_step']:.1f} ms: the paired 3-D kernels
were the round's only step-time gain, -0.6 ms; the 1x1-into-producer fusion was again not built), batch 1 <= 2.4 ms graphed
({sw['1']['graph_ms_per_map']:.2f}), plane sweep >= 0.36 ({line['roofline_warp_init']['frac']:.2f}: the two-phase form was re-costed at ~20 % fewer instructions for 16 KB of LDS and
not built, section 11), cfg4 >= 36 samples/s ({cfg4['value']:.1f}).

"""
    text += f"""**Batch 1** (the reference's harness, `test.py:101-104`): {sw['1']['eager_ms_per_map']:.2f} ms per map eager, **{sw['1']['graph_ms_per_map']:.2f} ms** through the captured HIP graph
(`GraphedForward`; `python -m diffmvs_amd.eval` now replays the graph by default for `--batch_size` <= 8, also with the scene cache);
{sw['2']['eager_ms_per_map']:.2f} at B=2, {sw['4']['eager_ms_per_map']:.2f} at B=4, {sw['8']['eager_ms_per_map']:.2f} at B=8.  The profile
of the batch-1 forward (`profiles/r5_bench_b1_kernel_stats.csv`): 3.0 ms of kernel time over 279 launches, the 2-D convolutions at
~14 us average.  The launch count is not what bounds it: a batch-1 convolution has 20-320 workgroups for 256 CUs and each workgroup
runs its cin/8 chunks as a dependent DMA -> wait -> MFMA chain of ~1.5 us per chunk with nothing else resident to overlap it.  With the
scene cache a batch-1 forward no longer contains FeatureNet on its 6 images (~1/5 of its launches).

Second lines (never the headline):
* `bench.py --scene-mode` (`profiles/r5_bench_scene_line.json`; the headline's network and geometry evaluated as whole 49-view scenes,
  every image through FeatureNet once per scene -- section 8): **{round(scene['value'])} depth-maps/s** ({scene['ms_per_step']:.1f} ms per {scene['config']['ref_views_per_gpu_per_step']}-view step) against
  {round(scene['per_sample']['depth_maps_per_s'])} for the same scenes through the per-sample forward, same process: **{scene['speedup_vs_per_sample']:.2f}x**; `--scene-mode --config cfg3`
  (`r5_bench_scene_cfg3_line.json`: one 49-view scene of 1152x864 images per step, CasDiffMVS, bf16): {sc3['value']:.0f} against {sc3['per_sample']['depth_maps_per_s']:.0f} depth-maps/s, {sc3['speedup_vs_per_sample']:.2f}x;
* `bench.py --config cfg3` (CasDiffMVS 1152x864, 7 src views, 4 reference views per step, bf16 feature storage + bf16 matrix
  arithmetic, `dtype` "bf16"): **{cfg3['value']:.1f} depth-maps/s** ({cfg3['ms_per_step']:.1f} ms per step; round 4: 184.8);
* `bench.py --config cfg4` (CasDiffMVS training step 768x576, 8 src views, batch 4 per GPU, fp32, `Trainer.train_sample`):
  **{cfg4['value']:.1f} samples/s** on one GPU ({cfg4['ms_per_step']:.0f} ms per step); `DMVS_GETCOST_BWD=gather` (per-pixel gather backward only, no tile
  pre-pass) {P('r5_bench_cfg4_gather_bwd_line.json')['value']:.1f}: the hybrid default costs nothing on the noise geometry of a random-weight model; with `--gpus N`
  the line carries `allreduce_ms_per_step` and the RCCL world size (one 3.70 MB all-reduce per step) -- not exercised on hardware (1-GPU boxes);
* `bench.py --config cfg5` (BASELINE configs[4]: CasDiffMVS 1920x1056, 11 src views, nd_initial 96, fp16 feature storage, scenes
  sharded over ranks): **{cfg5['value']:.1f} depth-maps/s** at the default 2 reference views per step ({cfg5['ms_per_step']:.1f} ms), {cfg5b8['value']:.1f} at `--batch 8`,
  {cfg5b16['value']:.1f} at `--batch 16` (GetCost `frac` {cfg5['roofline']['frac']:.2f} / {cfg5b8['roofline']['frac']:.2f} / {cfg5b16['roofline']['frac']:.2f} -- half-size texels, the same number of lines -- plane sweep
  {cfg5['roofline_warp_init']['frac']:.2f} / {cfg5b8['roofline_warp_init']['frac']:.2f} / {cfg5b16['roofline_warp_init']['frac']:.2f}): the largest configuration is within 14 % of its large-batch rate already at batch 2; the
  8-GPU figure the config names is the driver's to measure.
"""
    p = os.path.join(ROOT, "DESIGN.md")
    s = open(p).read()
    a = s.index("End-of-round numbers (1 x MI355X")
    b = s.index("If the boundary hands over host tensors instead")
    s = s[:a] + text + s[b:]
    open(p, "w").write(s)
    print("DESIGN.md section 7 regenerated:", round(line["value"]), "depth-maps/s,", line["ms_per_step"], "ms")


if __name__ == "__main__":
    main()
