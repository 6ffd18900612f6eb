#!/bin/bash
# A/B of one engine option on the headline step: tools/ab_bench_opt.sh NAME V1 V2 [bench args]   (prints ms_per_step per value)
NAME=$1; shift; A=$1; shift; B=$1; shift
for v in $A $B $A $B; do
  python3 bench.py --quick --steps 20 --warmup 5 --opt $NAME=$v "$@" 2>/dev/null | python3 -c '
import sys, json
d = json.loads(sys.stdin.readline())
print(sys.argv[1], "ms_per_step", d["ms_per_step"], "every voxel stored", d.get("ms_per_step_every_voxel_stored"))' "$NAME=$v"
done
