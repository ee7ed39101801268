"""The numbers DESIGN.md quotes for the end of the round are the ones in the committed bench line and rocprof summary
(profiles/): a stale headline is the easiest thing to leave behind after a late kernel change."""
import csv
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _design():
    return open(os.path.join(ROOT, "DESIGN.md")).read()


def test_design_quotes_the_committed_bench_line():
    line = json.load(open(os.path.join(ROOT, "profiles", "r5_bench_line.json")))
    text = _design()
    text = text[text.index("End-of-round numbers"):]
    text = text[:text.index("**Batch 1**")]
    assert line["metric"].startswith("depth-maps/sec") and line["n_gpus"] == 1 and line["dtype"] == "f32"
    assert f"**{round(line['value'])} depth-maps/s**" in text
    assert f"{line['ms_per_step']:.1f} ms per step" in text
    for key in ("roofline", "roofline_scene_geometry", "roofline_warp_init", "roofline_conv2d"):
        name = "roofline.frac" if key == "roofline" else key + ".frac"
        m = re.search(r"`%s`\s+(0\.\d+)" % re.escape(name), text)
        assert m, name
        assert abs(float(m.group(1)) - line[key]["frac"]) < 0.006, (name, m.group(1), line[key]["frac"])
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and abs(rf["achieved"] / rf["peak"] - rf["frac"]) < 1e-3
    assert abs(rf["algorithmic_bytes_per_launch"] / (rf["avg_launch_us"] * 1e-6) / 1e9 - rf["achieved"]) < 1.0
    # the full default run (batch sweep, CPU leg) is its own committed line
    full = json.load(open(os.path.join(ROOT, "profiles", "r5_bench_full_line.json")))
    assert f"{round(full['value'])} depth-maps/s" in text and f"{full['ms_per_step']:.1f} ms" in text
    cb = full["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0
    assert f"CPU baseline {cb['value']:.2f} maps/s" in text


def test_kernel_stats_agree_with_the_bench_line():
    """the rocprof average of the plane-sweep kernel and the event-timed figure on the bench line of the same run agree"""
    line = json.load(open(os.path.join(ROOT, "profiles", "r5_bench_b96_profiled_line.json")))
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "r5_bench_b96_kernel_stats.csv"))))
    wi = [r for r in rows if "warp_init_band_kernel" in r["Name"]]
    assert len(wi) == 1
    prof_us = float(wi[0]["AverageNs"]) / 1e3
    assert abs(prof_us - line["roofline_warp_init"]["avg_launch_us"]) / prof_us < 0.03
    # GetCost: the csv mixes the steps' launches with bench.py's scene-geometry side measurement (25 launches)
    gq = [r for r in rows if "getcost_quad_kernel" in r["Name"]][0]
    n, avg = int(gq["Calls"]), float(gq["AverageNs"]) / 1e3
    side = line["roofline_scene_geometry"]["avg_launch_us"]
    steps_avg = (n * avg - 25 * side) / (n - 25)
    assert abs(steps_avg - line["roofline"]["avg_launch_us"]) / steps_avg < 0.05


def test_design_test_counts_match_the_suite():
    """DESIGN.md section 5 quotes how many tests each marker selects; count them"""
    import subprocess
    import sys
    m = re.search(r"`pytest -m gpu`: (\d+) tests.*?`-m \"not gpu\"`: (\d+) tests", _design(), re.S)
    assert m, "DESIGN.md section 5 no longer states the test counts"
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "-q", "--collect-only", "-m", "gpu"],
                         capture_output=True, text=True, cwd=ROOT).stdout
    n_gpu, n_all = map(int, re.search(r"(\d+)/(\d+) tests collected", out).groups())
    assert (n_gpu, n_all - n_gpu) == (int(m.group(1)), int(m.group(2))), (n_gpu, n_all - n_gpu, m.groups())


def test_traffic_file_matches_the_kernel_source():
    """profiles/r5_getcost_traffic.json was measured on the committed warp kernels (bench.py refuses it otherwise)"""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    t = json.load(open(os.path.join(ROOT, "profiles", "r5_getcost_traffic.json")))
    assert t["kernel_source_sha"] == bench.kernel_source_hash()
    assert 0.8 < t["traffic_bytes_per_launch"] / 1785200640 < 1.2
