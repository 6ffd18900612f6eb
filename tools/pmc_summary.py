#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per kernel name from the counter_collection CSVs of several passes."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root, out, command=None, shape=None):
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
    # HBM bytes per launch as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes:
    # FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read.
    import json
    traffic = {}
    for k in acc:
        if "FETCH_SIZE" in acc[k] and "WRITE_SIZE" in acc[k]:
            f = acc[k]["FETCH_SIZE"][0] / acc[k]["FETCH_SIZE"][1]
            w = acc[k]["WRITE_SIZE"][0] / acc[k]["WRITE_SIZE"][1]
            traffic[k.replace("void ", "").strip()] = {"fetch_kib_raw": f, "write_kib": w, "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
    doc = {"method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                     "(gfx950: FETCH_SIZE counts 128-B requests as 64 B)", "kernels": traffic}
    if command:
        doc["command"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- " + command
    if shape:
        doc["shape"] = [int(v) for v in shape.split("x")]           # slices (= image side) x image side x tilts
    json.dump(doc, open(os.path.splitext(out)[0] + ".json", "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main(*sys.argv[1:5])
