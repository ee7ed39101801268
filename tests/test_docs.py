"""The numbers DESIGN.md quotes for the end of the round are the ones in the committed bench lines and rocprof summaries (profiles/): a stale
headline is the easiest thing to leave behind after a late kernel change.  DESIGN.md is the CURRENT design (<= 400 lines); the per-round
history lives in profiles/HISTORY.md."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _design():
    return open(os.path.join(ROOT, "DESIGN.md")).read()


def _P(name):
    return json.load(open(os.path.join(ROOT, "profiles", name)))


def test_design_is_the_current_design_not_the_history():
    text = _design()
    assert len(text.splitlines()) <= 400
    assert os.path.exists(os.path.join(ROOT, "profiles", "HISTORY.md"))
    for n in (1, 2, 3, 4, 5):
        assert f"Round {n}:" not in text and f"**Round {n}" not in text, f"per-round archaeology (round {n}) belongs in profiles/HISTORY.md"


def test_design_numbers_block_is_generated_from_the_committed_lines():
    import design_numbers as DN
    text = _design()
    a, b = text.index(DN.BEGIN) + len(DN.BEGIN), text.index(DN.END)
    assert text[a:b].strip() == DN.block().strip(), "run python tools/design_numbers.py"
    line = _P("r6_bench_line.json")
    assert line["metric"].startswith("depth-maps/sec") and line["n_gpus"] == 1 and line["dtype"] == "f32"
    assert line["config"]["workload"].startswith("DiffMVS DTU eval 640x512, 5 src views, numdepth_initial=48")
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and abs(rf["achieved"] / rf["peak"] - rf["frac"]) < 1e-3
    assert abs(rf["algorithmic_bytes_per_launch"] / (rf["avg_launch_us"] * 1e-6) / 1e9 - rf["achieved"]) < 1.0
    cp = rf["ceiling_probe"]
    assert cp["measured_by"] == "this process" and len(cp["in_step_probe_us"]) == 4 and cp["in_step_probe_avg_us"] > cp["gate_0p60_us"]
    assert cp["random_line_gather"]["every_line_once"]["us"] > 0
    cb = line["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0
    cv = line["roofline_conv2d"]
    assert cv["rows"] and all(r[-1] in ("mfma", "hbm") for r in cv["rows"]) and cv["hbm_achieved_GBs"] > 0


def test_kernel_stats_agree_with_the_bench_line():
    """the rocprof average of the plane-sweep kernel and the event-timed figure on the bench line of the same run agree"""
    line = _P("r6_bench_b96_profiled_line.json")
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "r6_bench_b96_kernel_stats.csv"))))
    wi = [r for r in rows if "warp_init_band_kernel" in r["Name"]]
    assert len(wi) == 1
    prof_us = float(wi[0]["AverageNs"]) / 1e3
    assert abs(prof_us - line["roofline_warp_init"]["avg_launch_us"]) / prof_us < 0.03
    # GetCost: the csv mixes the timed steps' launches with the untimed legs (scene geometry: 25 launches; the probe legs replay the product too)
    gq = [r for r in rows if "getcost_quad_kernel" in r["Name"]][0]
    assert abs(float(gq["MinNs"]) / 1e3 - line["roofline_scene_geometry"]["avg_launch_us"]) / line["roofline_scene_geometry"]["avg_launch_us"] < 0.15
    assert float(gq["MaxNs"]) / 1e3 > line["roofline"]["avg_launch_us"] * 0.95


def test_design_test_counts_match_the_suite():
    """DESIGN.md section 5 quotes how many tests each marker selects; count them"""
    import subprocess
    m = re.search(r"`pytest -m gpu`: (\d+) tests.*?`-m \"not gpu\"`: (\d+) tests", _design(), re.S)
    assert m, "DESIGN.md section 5 no longer states the test counts"
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "-q", "--collect-only", "-m", "gpu"],
                         capture_output=True, text=True, cwd=ROOT).stdout
    n_gpu, n_all = map(int, re.search(r"(\d+)/(\d+) tests collected", out).groups())
    assert (n_gpu, n_all - n_gpu) == (int(m.group(1)), int(m.group(2))), (n_gpu, n_all - n_gpu, m.groups())


def test_traffic_file_matches_the_kernel_source():
    """profiles/r6_getcost_traffic.json was measured on the committed warp kernels (bench.py refuses it otherwise)"""
    sys.path.insert(0, ROOT)
    import bench
    t = _P("r6_getcost_traffic.json")
    assert t["kernel_source_sha"] == bench.kernel_source_hash()
    assert 0.8 < t["traffic_bytes_per_launch"] / 1785200640 < 1.2
