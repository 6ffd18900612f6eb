#!/usr/bin/env python3
"""Time tv_fgp per inner iteration: two iterations per pass (k_fgp_fused2) against one (k_fgp_fused)."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomo_tv_amd import _lib as _tl
if os.environ.get("TOMO_LIB"):                      # A/B against another build of the library (tools only)
    _tl.LIB_PATH = os.path.abspath(os.environ["TOMO_LIB"])
from tomo_tv_amd.engine import tomoengine
from tomo_tv_amd.phantom import ellipsoids
from tomo_tv_amd._lib import VOL_RECON
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=512)
ap.add_argument("--nslice", type=int, default=512)
ap.add_argument("--iters", type=int, default=21)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
t = tomoengine(a.nslice, a.n, np.deg2rad(np.array([-20.0, 35.0])))
x = ellipsoids(a.nslice, a.n)
for pair in (1, 0, 1, 0):
    t.set_option("fgp_pair", pair)
    t.set_volume(x, VOL_RECON); t.tv_fgp(3, 0.1); t.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        t.tv_fgp(a.iters, 0.1)
    t.synchronize()
    ms = (time.perf_counter() - t0) / a.reps * 1e3
    print(f"fgp_pair={pair}: tv_fgp({a.iters}) {ms:.2f} ms = {ms / a.iters * 1e3:.0f} us per iteration (incl. the TV value and the final pass)")
