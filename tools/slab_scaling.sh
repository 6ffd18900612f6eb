#!/bin/bash
# Per-rank cost of the strong-scaling headline on ONE GPU: the ASD-POCS step on the slab a rank of an N-GPU run owns
# (512/N slices of the 512^3 x 90 volume), through the slab-sharded engine with its real collective calls (world 1).
for ns in ${SLABS:-512 256 128 64}; do
python bench.py --quick --force-dist --nslice $ns --steps ${STEPS:-10} --warmup 2 "$@" 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); r = d['roofline']
print(d['config']['slices_per_gpu'], 'slices: ms/step', round(d['ms_per_step'], 3), '|', r['kernel'], round(1e3 * r['avg_ms'], 1), 'us | tv norm', round(1e3 * d['roofline_tv_norm']['avg_ms'], 1), 'us | tv update', round(1e3 * d['roofline_tv_update']['avg_ms'], 1), 'us')"
done
