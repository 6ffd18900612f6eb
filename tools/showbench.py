import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    r=[f.split('/')[-1], round(d["ms_per_step"],2)]
    for k in ("roofline","roofline_bp_angle","roofline_fp_angle"):
        q=d.get(k)
        if q: r.append(f'{q["kernel"]}:{q["avg_ms"]*1e3:.1f}us/{q["achieved"]:.0f}GB/s x{q["launches"]}')
    print(*r)
