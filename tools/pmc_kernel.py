"""Average the PMC counters of one kernel over its dispatches: python tools/pmc_kernel.py <substring> <dir> [<dir> ...]
(each <dir> = one `rocprofv3 --kernel-trace --pmc ...` pass; counters are summed over the chip by rocprofv3)."""
import csv
import glob
import sys
from collections import defaultdict


def main():
    pat, dirs = sys.argv[1], sys.argv[2:]
    for d in dirs:
        acc, n = defaultdict(float), defaultdict(int)
        for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if pat in row["Kernel_Name"]:
                    acc[row["Counter_Name"]] += float(row["Counter_Value"])
                    n[row["Counter_Name"]] += 1
        for k in sorted(acc):
            print("%-32s %16.0f  (avg of %d dispatches)" % (k, acc[k] / n[k], n[k]))


if __name__ == "__main__":
    main()
