#!/usr/bin/env python3
"""Time one SART sweep (per-angle fused steps) for engine option settings."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomo_tv_amd import _lib as _tl
if os.environ.get("TOMO_LIB"):                      # A/B against another build of the library (tools only)
    _tl.LIB_PATH = os.path.abspath(os.environ["TOMO_LIB"])
from tomo_tv_amd.engine import tomoengine
from tomo_tv_amd.phantom import ellipsoids, tilt_angles
from tomo_tv_amd._lib import VOL_ORIGINAL

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=512)
ap.add_argument("--nslice", type=int, default=512)
ap.add_argument("--nproj", type=int, default=90)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--sweep", action="append", default=[], help="name=v1,v2,...")
a = ap.parse_args()
t = tomoengine(a.nslice, a.n, np.deg2rad(tilt_angles(a.nproj)))
t.set_volume(ellipsoids(a.nslice, a.n, k=4), VOL_ORIGINAL)
t.create_projections()
t.restart_recon()
def run(tag):
    t.SART(0.5, 1); t.synchronize()
    t0 = time.perf_counter()
    t.SART(0.5, a.reps); t.synchronize()
    el = (time.perf_counter() - t0) / a.reps
    print(f"{tag}: sweep {el * 1e3:.2f} ms = {el / a.nproj * 1e6:.1f} us per angle step")
if not a.sweep: run("default")
for sw in a.sweep:
    k, vs = sw.split("=")
    for v in vs.split(","):
        t.set_option(k, int(v)); run(f"{k}={v}")
