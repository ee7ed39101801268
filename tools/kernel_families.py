"""Per-family split of a rocprofv3 kernel_stats.csv of a bench.py run: python tools/kernel_families.py <csv> <forwards>

<forwards> = model forwards in the profiled process (bench.py: warm-up + timed steps + 1 per-conv-event step), so that the
totals print as ms per forward; without it the totals are those of the whole profiled run and are labelled so.  Families: conv2d (incl. the direct 1x1 kernel), conv3d, stem, getcost (NB: includes bench.py's
untimed scene-geometry side measurement), warp_init, GroupNorm, other."""
import csv
import sys


def family(name):
    if "conv2d_mfma" in name or "conv1x1_px4" in name or "conv1x1_direct" in name:
        return "conv2d"
    if "stem" in name:
        return "stem"
    if "conv3d" in name:
        return "conv3d"
    if "getcost" in name:
        return "getcost"
    if "warp_init" in name:
        return "warp_init"
    if "gn_" in name or "groupnorm" in name:
        return "groupnorm"
    return "other"


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    given = len(sys.argv) > 2
    fwd = float(sys.argv[2]) if given else 1.0
    unit = "ms per forward" if given else "ms in the whole profiled run"
    per = "/fwd" if given else " calls"
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    fam = {}
    for r in rows:
        fam[family(r["Name"])] = fam.get(family(r["Name"]), 0.0) + float(r["TotalDurationNs"])
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1]):
        print(f"{k:10s} {100 * v / tot:5.1f} %  {v / fwd / 1e6:8.2f} {unit}")
    print(f"{'all':10s} 100.0 %  {tot / fwd / 1e6:8.2f} {unit}")
    for r in rows[:25]:
        print(f"  {float(r['AverageNs']) / 1e3:9.1f} us x {int(r['Calls']) / fwd:6.1f}{per}  {r['Name'][:110]}")


if __name__ == "__main__":
    main()
