#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per kernel name from the counter_collection CSVs of several passes."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root, out):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            a = acc[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    names = sorted({c for k in acc for c in acc[k]})
    lines = ["# mean counter value per dispatch (rocprofv3 --pmc, one pass per counter group)",
             f"{'kernel':<62}" + "".join(f"{n:>22}" for n in names) + f"{'dispatches':>12}"]
    for k in sorted(acc):
        n = max(v[1] for v in acc[k].values())
        lines.append(f"{k:<62}" + "".join(f"{(acc[k][c][0] / acc[k][c][1]) if c in acc[k] else float('nan'):>22.1f}" for c in names) + f"{n:>12}")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
