"""Copy the artefacts of the round's last GPU session (gpurun_out/r5_final, tools/sessions/r5_final.sh) into profiles/ under their committed
names, fold the ceiling probe, and regenerate DESIGN.md section 7:   python tools/ingest_final.py [session dir] [--bench-only]"""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last(f):
    return json.loads([ln for ln in open(f).read().strip().splitlines() if ln.startswith("{")][-1])


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    S = os.path.join(ROOT, args[0] if args else "gpurun_out/r5_final")
    P = os.path.join(ROOT, "profiles")
    pairs = [("bench_full", "r5_bench_line"), ("bench_full", "r5_bench_full_line")]
    if "--bench-only" not in sys.argv:
        pairs += [("prof_b96_line", "r5_bench_b96_profiled_line"), ("bench_cfg3", "r5_bench_cfg3_line"), ("bench_cfg4", "r5_bench_cfg4_line"),
                  ("bench_cfg5_b2", "r5_bench_cfg5_line"), ("bench_cfg5_b8", "r5_bench_cfg5_b8_line"), ("bench_cfg5_b16", "r5_bench_cfg5_b16_line"),
                  ("bench_scene", "r5_bench_scene_line")]
    S1 = os.path.join(ROOT, "gpurun_out", "r5_final")        # cfg4 and cfg5 at batch 8 / 16 were not repeated by r5_final2.sh (identical kernels)
    for src, dst in pairs:
        f = os.path.join(S, src + ".json")
        json.dump(last(f if os.path.exists(f) else os.path.join(S1, src + ".json")), open(os.path.join(P, dst + ".json"), "w"))
    with open(os.path.join(P, "r5_conv2d_layers_b96.txt"), "w") as f:
        f.write("".join(ln for ln in open(os.path.join(S, "bench_full.err")) if "amdgpu.ids" not in ln))
    if "--bench-only" not in sys.argv:
        shutil.copy(os.path.join(S, "b96_kernel_stats.csv"), os.path.join(P, "r5_bench_b96_kernel_stats.csv"))
        for f in ("r5_getcost_traffic.json", "r5_pmc_hbm_traffic_per_kernel.csv"):
            shutil.copy(os.path.join(S, f), os.path.join(P, f))
        shutil.copy(os.path.join(S, "getcost_probe.jsonl"), os.path.join(P, "r5_getcost_ceiling_probe_final_b96.jsonl"))
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ceiling_probe.py"), os.path.join(P, "r5_getcost_ceiling_probe_final_b96.jsonl")], check=True)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_numbers.py")], check=True)


if __name__ == "__main__":
    main()
