#!/bin/bash
# PMC counters of the kernels of one command, one rocprofv3 pass per counter group (no tracing domains besides the kernel
# dispatch records the counters ride on).  tools/pmc_kernel.sh OUTDIR "COUNTERS A" "COUNTERS B" ... -- python3 script args
O=$1; shift
CGRP=()
while [ "$1" != "--" ]; do CGRP+=("$1"); shift; done
shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for g in "${CGRP[@]}"; do
  rocprofv3 --pmc $g --output-format csv -d $O/p$i -- "$@" > $O/p$i.log 2>&1
  i=$((i+1))
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-44:]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    if any(len(v) for v in cs.values()):
        print(k)
        for c, v in sorted(cs.items()):
            print(f"    {c:28s} mean {sum(v) / len(v):16.1f}  over {len(v)} dispatches")
PY
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
