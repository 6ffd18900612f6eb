#!/usr/bin/env python3
"""Print avg duration of kernels whose name contains a substring, from a rocprofv3 kernel_stats.csv dir."""
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in sys.argv[2:]):
            print(f'{r["Name"][:60]:<60} calls={r["Calls"]:>5} avg_us={float(r["AverageNs"])/1e3:9.1f}')
