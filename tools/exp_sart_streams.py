#!/usr/bin/env python3
"""SART sweep time for sart_streams = 1 / 2 without any event profiling (see tools/exp_two_streams.py for two engines)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomo_tv_amd.engine import tomoengine
from tomo_tv_amd.phantom import ellipsoids, tilt_angles
from tomo_tv_amd._lib import VOL_ORIGINAL
n, P, nx = 512, 90, int(sys.argv[1]) if len(sys.argv) > 1 else 512
t = tomoengine(nx, n, np.deg2rad(tilt_angles(P)))
t.set_volume(ellipsoids(nx, n), VOL_ORIGINAL)
t.create_projections()
t.initialize_SART("sequential")
for ns in (1, 2, 1, 2):
    t.set_option("sart_streams", ns)
    t.SART(0.5, 1); t.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): t.SART(0.5, 1)
    t.synchronize()
    print("sart_streams=%d: %.2f ms per sweep" % (ns, (time.perf_counter() - t0) / 5 * 1e3))
