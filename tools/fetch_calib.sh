#!/bin/bash
# FETCH_SIZE / WRITE_SIZE against known byte counts for the access widths of k_sart_resident (tools/micro/fetch_calib.hip).
#   gpurun -- 'bash tools/fetch_calib.sh > gpurun_out/r06_fetch_calibration.txt 2>&1'
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
O=$R/gpurun_out/fetch_calib; mkdir -p $O
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 $R/tools/micro/fetch_calib.hip -o $O/fetch_calib
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --output-format csv -d $O/p_$c -- $O/fetch_calib > $O/p_$c.log 2>&1; done
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
acc = collections.defaultdict(dict)
for f in glob.glob(O + "/p_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]] = float(r["Counter_Value"])
GiB = 1 << 30
print("# counter (KiB x 1024) / bytes actually touched (1 GiB read or written once, nothing cached)")
print(f"{'kernel':<18}{'touched':>14}{'FETCH_SIZE':>16}{'factor':>9}{'WRITE_SIZE':>16}{'factor':>9}")
for k in ("k_read_vec16", "k_read_poll8", "k_read_vec4", "k_read_scalar", "k_touch_lines", "k_write_gran8", "k_write_vec4", "k_write_vec16"):
    c = acc.get(k, {})
    f, w = c.get("FETCH_SIZE", float("nan")) * 1024, c.get("WRITE_SIZE", float("nan")) * 1024
    print(f"{k:<18}{GiB:>14}{f:>16.0f}{f / GiB:>9.3f}{w:>16.0f}{w / GiB:>9.3f}")
PY
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
