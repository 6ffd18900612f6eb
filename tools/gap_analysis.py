#!/usr/bin/env python3
"""Idle time between consecutive kernels in a rocprofv3 kernel trace: gap_analysis.py <dir with *_kernel_trace.csv>"""
import csv, glob, os, sys
from collections import defaultdict
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
busy_end, gaps, gap_after = t0, [], defaultdict(lambda: [0, 0.0])
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > busy_end and prev is not None:
        g = (s - busy_end) / 1e3
        if g < 1000:                      # ignore host-side pauses between steps
            gaps.append(g)
            k = prev.split("(")[0][-40:]
            gap_after[k][0] += 1; gap_after[k][1] += g
    busy_end = max(busy_end, e); prev = r["Kernel_Name"]
print(f"kernels {len(rows)}, span {(t1 - t0) / 1e6:.1f} ms, idle in gaps < 1 ms: {sum(gaps) / 1e3:.2f} ms over {len(gaps)} gaps (mean {sum(gaps) / max(1, len(gaps)):.2f} us)")
for k, (n, tot) in sorted(gap_after.items(), key=lambda kv: -kv[1][1])[:8]:
    print(f"  after {k:<42} {n:5d} gaps, {tot / 1e3:7.3f} ms, mean {tot / n:6.2f} us")
