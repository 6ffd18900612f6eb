#!/usr/bin/env python3
"""Time the TV descent (tv_gd) and FGP-TV (tv_fgp) alone for engine option settings."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomo_tv_amd import _lib as _tl
if os.environ.get("TOMO_LIB"):                      # A/B against another build of the library (tools only)
    _tl.LIB_PATH = os.path.abspath(os.environ["TOMO_LIB"])
from tomo_tv_amd.engine import tomoengine
from tomo_tv_amd.phantom import ellipsoids, tilt_angles
from tomo_tv_amd._lib import VOL_RECON

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=512)
ap.add_argument("--nslice", type=int, default=512)
ap.add_argument("--ng", type=int, default=10)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--sweep", action="append", default=[], help="name=v1,v2,...")
a = ap.parse_args()
t = tomoengine(a.nslice, a.n, np.deg2rad(tilt_angles(8)))
x = ellipsoids(a.nslice, a.n) + 0.01 * np.random.default_rng(0).random((a.nslice, a.n, a.n), dtype=np.float32)
def run(tag):
    t.set_volume(x, VOL_RECON)
    t.tv_gd(a.ng, 0.5); t.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps): t.tv_gd(a.ng, 0.5)
    t.synchronize()
    gd = (time.perf_counter() - t0) / a.reps / a.ng
    t.tv_fgp(a.ng, 0.1); t.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps): t.tv_fgp(a.ng, 0.1)
    t.synchronize()
    fgp = (time.perf_counter() - t0) / a.reps / a.ng
    print(f"{tag}: tv_gd {gd * 1e6:.1f} us per inner iteration, tv_fgp {fgp * 1e6:.1f} us per iteration")
if not a.sweep: run("default")
for sw in a.sweep:
    k, vs = sw.split("=")
    for v in vs.split(","):
        t.set_option(k, int(v)); run(f"{k}={v}")
