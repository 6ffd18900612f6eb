#!/usr/bin/env python3
"""Time the chained ART sweep of the ctvlib facade (cpu/sim_ASD.py's projection step) at 512^3 x 90."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomo_tv_amd import _lib as _tl
if os.environ.get("TOMO_LIB"):
    _tl.LIB_PATH = os.path.abspath(os.environ["TOMO_LIB"])
from tomo_tv_amd.engine import ctvlib, system_matrix
from tomo_tv_amd.phantom import ellipsoids, tilt_angles
ns, n, P = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 512, 90
ang = tilt_angles(P)
c = ctvlib(ns, n, P); c.load_A(system_matrix(n, ang))
c.set_original_volume_all(ellipsoids(ns, n)) if hasattr(c, "set_original_volume_all") else None
from tomo_tv_amd._lib import VOL_ORIGINAL
c.set_volume(ellipsoids(ns, n), VOL_ORIGINAL); c.create_projections(); c.row_inner_product()
c.ART(0.5); c.synchronize()
t0 = time.perf_counter()
for _ in range(3): c.ART(0.5)
c.synchronize()
print(f"ART sweep {ns}x{n}x{n} P={P}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms")
