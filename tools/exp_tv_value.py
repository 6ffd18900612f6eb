#!/usr/bin/env python3
"""tv() alone: the value-only mode of the branch-free march against the register march (tv_march4 = 0)."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from tomo_tv_amd.engine import tomoengine
from tomo_tv_amd.phantom import ellipsoids, tilt_angles
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 512
t = tomoengine(nx, 512, np.deg2rad(tilt_angles(4)))
t.set_volume(ellipsoids(nx, 512) + 0.01 * np.random.default_rng(0).random((nx, 512, 512), dtype=np.float32))
vals = {}
for m4 in (1, 0, 1):
    t.set_option("tv_march4", m4)
    v = t.tv(); t.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): t.tv()
    t.synchronize()
    print(f"tv_march4={m4}: tv() {(time.perf_counter() - t0) / 20 * 1e6:.1f} us, value {v!r}")
    vals[m4] = v
assert abs(vals[1] - vals[0]) <= 1e-12 * vals[0]
