#!/usr/bin/env python3
"""Which voxels differ between the TV descent kernel forms? (diagnosis helper)"""
import sys
import numpy as np
sys.path.insert(0, ".")
from tomo_tv_amd._lib import VOL_RECON
from tomo_tv_amd.engine import tomoengine
from tomo_tv_amd.phantom import ellipsoids, tilt_angles
nx, n = int(sys.argv[1]), int(sys.argv[2])
ng = int(sys.argv[3]) if len(sys.argv) > 3 else 1
t = tomoengine(nx, n, np.deg2rad(tilt_angles(4)))
x = ellipsoids(nx, n) + 0.05 * np.random.default_rng(0).random((nx, n, n), dtype=np.float32)
res = {}
for name, opts in (("march4", {"tv_lds": 1, "tv_march4": 1}), ("reg", {"tv_lds": 1, "tv_march4": 0}), ("lds", {"tv_lds": 8, "tv_march4": 1})):
    for k, v in opts.items():
        t.set_option(k, v)
    t.set_volume(x, VOL_RECON)
    tv0 = t.tv_gd(ng, 1.0)
    res[name] = (tv0, t.get_volume())
for a, b in (("march4", "reg"), ("march4", "lds"), ("reg", "lds")):
    d = res[a][1] != res[b][1]
    idx = np.argwhere(d)
    print(a, b, "tv0", res[a][0] == res[b][0], "mismatches", int(d.sum()), "max abs", float(np.abs(res[a][1] - res[b][1]).max()))
    if len(idx):
        print("   slices", np.unique(idx[:, 0])[:20], "count", len(np.unique(idx[:, 0])), " y", np.unique(idx[:, 1])[:12], " z", np.unique(idx[:, 2])[:12])
