#!/bin/bash
# SQ counters of the kernels whose name contains $1 while running "python3 tools/bench_fp.py $2..." (two passes; sums over launches)
cd /tmp && export TMPDIR=/tmp
K=$1; shift
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d /tmp/c1 -- python3 /root/repo/tools/bench_fp.py "$@" > /tmp/o1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/c2 -- python3 /root/repo/tools/bench_fp.py "$@" > /tmp/o2.log 2>&1
python3 - "$K" <<PY
import csv,glob,collections,sys
for d in ("/tmp/c1","/tmp/c2"):
    for f in glob.glob(d+"/**/*counter_collection.csv",recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:40]][r["Counter_Name"]]+=float(r["Counter_Value"])
        for k,v in acc.items():
            if sys.argv[1] in k: print(k, {a:int(b) for a,b in sorted(v.items())})
PY
