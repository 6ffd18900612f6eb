#!/usr/bin/env python3
"""Print the figures of a bench.py JSON line that a kernel change moves (scratch helper)."""
import json
import sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ['value', 'ms_per_step', 'ms_per_step_every_voxel_stored']})
for k in ['roofline', 'roofline_tv_norm', 'roofline_tv_update', 'roofline_bp_angle', 'roofline_fp_angle']:
    r = d.get(k)
    if r:
        print(f"  {k:20s} {r['kernel']:36s} {r['avg_ms'] * 1e3:8.1f} us  frac {r['frac']:.3f}  launches {r['launches']}")
for k, v in (d.get('secondary') or {}).items():
    extra = ''
    for rk in ('roofline', 'roofline_fp_all', 'roofline_bp_all'):
        if isinstance(v.get(rk), dict):
            extra += f" | {v[rk]['kernel']} {v[rk]['avg_ms'] * 1e3:.0f} us frac {v[rk]['frac']:.3f}"
    print(f"  {k:52s} {v.get('ms_per_step', float('nan')):9.3f} ms{extra}" if 'ms_per_step' in v else f"  {k}: {v}")
