#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats CSV directory into a short text summary for profiles/."""
import csv
import glob
import os
import sys


def main(d, out):
    files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    lines = []
    for f in sorted(files):
        lines.append(f"# {os.path.relpath(f, d)}")
        rows = list(csv.DictReader(open(f)))
        lines.append(f"{'kernel':<70} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'pct':>6}")
        for r in rows:
            name = r.get("Name", "")[:70]
            lines.append(f"{name:<70} {r.get('Calls',''):>7} {float(r.get('TotalDurationNs',0))/1e6:>10.3f} "
                         f"{float(r.get('AverageNs',0))/1e3:>10.2f} {float(r.get('MinNs',0))/1e3:>9.2f} "
                         f"{float(r.get('MaxNs',0))/1e3:>9.2f} {r.get('Percentage',''):>6}")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
