#!/usr/bin/env python3
"""Time the all-angle forward / back projection alone (HIP events through torch on the engine stream)."""
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
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--opt", action="append", default=[])
ap.add_argument("--sweep", action="append", default=[], help="name=v1,v2,...: time the projections for each value, e.g. bp_tile=0,1")
a = ap.parse_args()
t0 = time.perf_counter()
t = tomoengine(a.nslice, a.n, np.deg2rad(tilt_angles(a.nproj)))
print(f"create {time.perf_counter() - t0:.2f} s")
for kv in a.opt:
    k, v = kv.split("="); t.set_option(k, int(v))
t.set_volume(ellipsoids(a.nslice, a.n), VOL_ORIGINAL)

def timeit(fn):
    fn(); t.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps): fn()
    t.synchronize()
    return (time.perf_counter() - t0) / a.reps * 1e3

def run(tag):
    fp = timeit(lambda: t.create_projections())
    t.restart_recon()
    sirt = timeit(lambda: t.SIRT(1))
    print(f"{tag}: FP {fp:.3f} ms   SIRT iteration {sirt:.3f} ms")

if not a.sweep:
    run("default")
for sw in a.sweep:
    k, vs = sw.split("=")
    for v in vs.split(","):
        t.set_option(k, int(v)); run(f"{k}={v}")
