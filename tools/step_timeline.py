#!/usr/bin/env python3
"""Where one ASD-POCS step's wall time goes: step_timeline.py <dir with *_kernel_trace.csv> [step index].  Splits the kernel trace of
bench.py into steps at the first forward projection of a sweep (k_sart_tile<false...>) and prints, for one step, the phases by wall
clock (sweep, TV descent, everything else) and the union-busy time of each kernel family."""
import csv, glob, os, sys
from collections import defaultdict
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def fam(n):
    n = n.split("(")[0].replace("void ", "").replace("tomo::", "")
    if n.startswith("k_sart_tile<false"): return "fp_angle"
    if n.startswith("k_sart_tile"): return "sart_tile"
    if n.startswith("k_tv_march4"): return "tv"
    if n.startswith(("k_fp_tile", "k_fp_strip", "k_fp_list")): return "fp_all"
    return n.split("<")[0]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), fam(r["Kernel_Name"])) for r in rows]
# a sweep starts with two fp_angle launches (two chains) close together
starts = [i for i, e in enumerate(ev) if e[2] == "fp_angle" and (i == 0 or ev[i - 1][2] != "fp_angle")]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) - 3
a, b = starts[k], starts[k + 1]
step = ev[a:b]
t0, t1 = step[0][0], ev[b][0]
print(f"step {k}: {(t1 - t0) / 1e6:.3f} ms, {len(step)} kernels")
def union(evs):
    tot, end = 0, 0
    for s, e, _ in sorted(evs):
        if e > end:
            tot += e - max(s, end); end = e
    return tot
byf = defaultdict(list)
for e in step: byf[e[2]].append(e)
for n, evs in sorted(byf.items(), key=lambda kv: -union(kv[1])):
    print(f"  {n:<22} {len(evs):4d} launches, busy (union) {union(evs) / 1e6:7.3f} ms, first at {(min(e[0] for e in evs) - t0) / 1e6:7.3f}, last end {(max(e[1] for e in evs) - t0) / 1e6:7.3f}")
print(f"  all kernels: busy (union) {union(step) / 1e6:.3f} ms -> idle {(t1 - t0 - union(step)) / 1e6:.3f} ms")
