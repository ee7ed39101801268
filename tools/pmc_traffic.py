"""Fold the two rocprofv3 PMC passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; each with --kernel-trace, CSV output) into
profiles/<round>_pmc_hbm_traffic_per_kernel.csv and profiles/<round>_getcost_traffic.json.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <round-tag> <batch> [<kernel name prefix>]

Counter unit: KiB per dispatch (hbm_bytes = counter * 1024).  Corrections: the factors tools/traffic_calib.py measured on this
GPU in the quad kernels' own access patterns (profiles/r3_traffic_calibration.json: known bytes / counter): FETCH_SIZE x 1.91 for
the quad gather of whole 128-byte texels (x 2.0 for a wide coalesced stream, as MI355X_MICROARCH.md says; x 1.0 when only 64 bytes of
each line are requested), WRITE_SIZE x 0.947 for 4-byte-per-lane 64-byte runs.  The json records the sha of the warp kernels'
source it was measured on; bench.py refuses it when the source has changed since."""
import collections
import csv
import glob
import json
import os
import sys


def per_kernel(d, counter):
    f = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        grid = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
        acc[(r["Kernel_Name"], grid)].append(float(r["Counter_Value"]))
    return acc


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0]


def main():
    fetch_dir, write_dir, tag, batch = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    kname = sys.argv[5] if len(sys.argv) > 5 else "getcost_quad_kernel<32, 6,"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fe, wr = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    rows = []
    for (k, g), v in sorted(fe.items(), key=lambda kv: -sum(kv[1])):
        w = wr.get((k, g), [0.0])
        # (the GetCost launches of the timed steps come first in dispatch order; bench.py's untimed scene-geometry side
        # measurement -- same kernel, other inputs -- follows: keep the first 8 = 2 forwards x 4 GRU iterations)
        if kname in short(k):
            v, w = v[:8], w[:8]
        rows.append((short(k), g, len(v), sum(v) / len(v) / 1024, sum(w) / len(w) / 1024))
    out = os.path.join(root, "profiles", f"{tag}_pmc_hbm_traffic_per_kernel.csv")
    with open(out, "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, each with --kernel-trace) -- python bench.py "
                f"--steps 2 --warmup 1 --no-cpu-baseline (cfg2, B={batch}); mean MiB per dispatch, raw counters;\n"
                "# corrections per access pattern: profiles/r3_traffic_calibration.json\n")
        f.write("kernel,grid,dispatches,fetch_raw_MiB,write_MiB\n")
        for r in rows[:60]:
            f.write(f"\"{r[0]}\",{r[1]},{r[2]},{r[3]:.2f},{r[4]:.2f}\n")
    gc = [r for r in rows if r[0].startswith(kname)]
    if gc:
        r = max(gc, key=lambda r: r[1])
        fetch_raw, write = r[3] * 2 ** 20, r[4] * 2 ** 20
        cal = json.load(open(os.path.join(root, "profiles", "r3_traffic_calibration.json")))["patterns"]
        kf = [v["factor"] for k, v in cal.items() if "whole 128-byte texels" in k][0]
        kw = [v["factor"] for k, v in cal.items() if "64-byte runs" in k][0]
        sys.path.insert(0, root)
        import bench
        info = {"batch": batch, "kernel": kname.replace(", ", ","), "fetch_size_raw_bytes": int(fetch_raw), "fetch_correction": kf,
                "write_size_raw_bytes": int(write), "write_correction": kw,
                "traffic_bytes_per_launch": int(kf * fetch_raw + kw * write), "kernel_source_sha": bench.kernel_source_hash(),
                "calibration": "profiles/r3_traffic_calibration.json (tools/traffic_calib.py: 2 GiB known-byte kernels in the quad access patterns)",
                "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, profiles/{tag}_pmc_hbm_traffic_per_kernel.csv"}
        with open(os.path.join(root, "profiles", f"{tag}_getcost_traffic.json"), "w") as f:
            json.dump(info, f, indent=1)
        print(json.dumps(info))
    print(out)


if __name__ == "__main__":
    main()
