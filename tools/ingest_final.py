"""Copy the artefacts of the round's last GPU session (gpurun_out/r6_final, tools/sessions/r6_final.sh) into profiles/ under their committed
names and regenerate DESIGN.md's numbers block:   python tools/ingest_final.py [session dir] [--bench-only]"""
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
    S = os.path.join(ROOT, args[0] if args else "gpurun_out/r6_final")
    P = os.path.join(ROOT, "profiles")
    pairs = [("bench_full", "r6_bench_line")]
    if "--bench-only" not in sys.argv:
        pairs += [("prof_b96_line", "r6_bench_b96_profiled_line"), ("bench_cfg3", "r6_bench_cfg3_line"), ("bench_cfg4", "r6_bench_cfg4_line"),
                  ("bench_cfg5_b2", "r6_bench_cfg5_line"), ("bench_scene", "r6_bench_scene_line")]
    for src, dst in pairs:
        f = os.path.join(S, src + ".json")
        if os.path.exists(f):
            json.dump(last(f), open(os.path.join(P, dst + ".json"), "w"))
    with open(os.path.join(P, "r6_conv2d_layers_b96.txt"), "w") as f:
        f.write("".join(ln for ln in open(os.path.join(S, "bench_full.err")) if "amdgpu.ids" not in ln))
    if "--bench-only" not in sys.argv:
        for src, dst in (("b96_kernel_stats.csv", "r6_bench_b96_kernel_stats.csv"), ("cfg4_kernel_stats.csv", "r6_cfg4_kernel_stats.csv"),
                         ("b1_kernel_stats.csv", "r6_bench_b1_kernel_stats.csv")):
            if os.path.exists(os.path.join(S, src)):
                shutil.copy(os.path.join(S, src), os.path.join(P, dst))
        for f in ("r6_getcost_traffic.json", "r6_pmc_hbm_traffic_per_kernel.csv"):
            if os.path.exists(os.path.join(S, f)):
                shutil.copy(os.path.join(S, f), os.path.join(P, f))
        for f in ("pytest_gpu.log", "smoke.log"):
            if os.path.exists(os.path.join(S, f)):
                with open(os.path.join(P, "r6_final_" + f.replace(".log", "_tail.txt")), "w") as o:
                    o.write("".join(open(os.path.join(S, f)).readlines()[-4:]))
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_numbers.py")], check=True)


if __name__ == "__main__":
    main()
