"""Fold the two rocprofv3 PMC passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; each with --kernel-trace, CSV output) into
profiles/<round>_pmc_hbm_traffic_per_kernel.csv and profiles/<round>_getcost_traffic.json.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <round-tag> <batch> [<kernel name prefix>]

Counter unit: KiB per dispatch (hbm_bytes = counter * 1024).  gfx950 correction (MI355X_MICROARCH.md, HBM section):
FETCH_SIZE reports half of a wide coalesced read, so corrected_fetch = 2 x raw."""
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
    kname = sys.argv[5] if len(sys.argv) > 5 else "getcost_quad_kernel<32, 6>"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fe, wr = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    rows = []
    for (k, g), v in sorted(fe.items(), key=lambda kv: -sum(kv[1])):
        w = wr.get((k, g), [0.0])
        rows.append((short(k), g, len(v), sum(v) / len(v) / 1024, sum(w) / len(w) / 1024))
    out = os.path.join(root, "profiles", f"{tag}_pmc_hbm_traffic_per_kernel.csv")
    with open(out, "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, each with --kernel-trace) -- python bench.py "
                f"--steps 2 --warmup 1 --no-cpu-baseline (cfg2, B={batch}); mean MiB per dispatch, raw counters;\n"
                "# gfx950: corrected_fetch = 2 x raw FETCH_SIZE (MI355X_MICROARCH.md HBM section)\n")
        f.write("kernel,grid,dispatches,fetch_raw_MiB,write_MiB\n")
        for r in rows[:60]:
            f.write(f"\"{r[0]}\",{r[1]},{r[2]},{r[3]:.2f},{r[4]:.2f}\n")
    gc = [r for r in rows if r[0].startswith(kname)]
    if gc:
        r = max(gc, key=lambda r: r[1])
        fetch_raw, write = r[3] * 2 ** 20, r[4] * 2 ** 20
        info = {"batch": batch, "kernel": kname.replace(", ", ","), "fetch_size_raw_bytes": int(fetch_raw), "fetch_correction": 2.0,
                "write_size_bytes": int(write), "traffic_bytes_per_launch": int(2 * fetch_raw + write),
                "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, profiles/{tag}_pmc_hbm_traffic_per_kernel.csv"}
        with open(os.path.join(root, "profiles", f"{tag}_getcost_traffic.json"), "w") as f:
            json.dump(info, f, indent=1)
        print(json.dumps(info))
    print(out)


if __name__ == "__main__":
    main()
