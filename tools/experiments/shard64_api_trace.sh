#!/bin/bash
# Which call sites issue the ~110 tiny operations per ASD-POCS step of a 64-slice shard?  HIP API trace + kernel
# trace of bench.py's world-1 sharded step; the summary lists, per step, the sequence of API calls between two k_sart_resident launches
# with the kernels / copies / fills they produced.   gpurun -- 'bash tools/experiments/shard64_api_trace.sh [steps]'
R="$(cd "$(dirname "$0")/../.." && pwd)"; O=$R/gpurun_out/shard64_api; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 rocprofv3 --hip-runtime-trace --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python3 $R/bench.py --force-dist --quick --nslice 64 --nray 512 --nproj 90 --steps ${1:-6} --warmup 2 --no-kernel-log > $O/log.txt 2>&1
tail -1 $O/log.txt | cut -c1-200
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
api = []
for f in glob.glob(O + "/trace/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        api.append((int(r["Start_Timestamp"]), r["Function"], r.get("Thread_Id", "")))
api.sort()
names = [a[1] for a in api]
# one step = from one launch of the resident sweep to the next: find them through the kernel trace's correlation with hipLaunchKernel order
kern = []
for f in glob.glob(O + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kern.append((int(r["Start_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]))
kern.sort()
idx = [i for i, k in enumerate(kern) if "k_sart_resident" in k[1]]
print("kernels + copies per step (between the last two resident sweeps):")
if len(idx) >= 2:
    seg = kern[idx[-2]:idx[-1]]
    c = collections.Counter(k[1] for k in seg)
    for k, v in c.most_common():
        print(f"  {v:4d}  {k}")
    print("  total", len(seg))
    print("sequence:")
    print("  " + " | ".join(k[1].replace("tomo::", "").replace("void ", "")[:28] for k in seg))
cnt = collections.Counter(names)
print("API calls of the whole run:", dict(cnt.most_common(14)))
PY
find $O -name "*.db" -delete
