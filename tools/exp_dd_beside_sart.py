#!/usr/bin/env python3
"""What would the residual norm of a snapshot cost if it ran on the second stream beside the NEXT SART sweep instead of beside
the TV descent?  Sweep alone, sweep with the evaluation in flight, TV descent alone, TV descent with the evaluation in flight."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomo_tv_amd.engine import tomoengine
from tomo_tv_amd.phantom import ellipsoids, tilt_angles
from tomo_tv_amd._lib import VOL_ORIGINAL

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n, P = 512, 90
t = tomoengine(ns, n, np.deg2rad(tilt_angles(P)))
t.set_volume(ellipsoids(ns, n, k=4), VOL_ORIGINAL)
t.create_projections()
t.initialize_SART("sequential")
t.restart_recon()
t.SART(0.25, 1); t.copy_recon(); t.tv_gd(10, 0.5); t.synchronize()

def timed(f, reps=4):
    f(); t.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    t.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

def sweep(): t.SART(0.25, 1)
def sweep_dd():
    t.data_distance_begin(); t.SART(0.25, 1); t.be.c("async_wait")
def tv(): t.tv_gd(10, 0.5)
def tv_dd():
    t.data_distance_begin(); t.tv_gd(10, 0.5); t.be.c("async_wait")
def dd(): t.data_distance()
a, b, c, d, e = timed(sweep), timed(sweep_dd), timed(tv), timed(tv_dd), timed(dd)
print(f"{ns} slices: sweep {a:.2f} ms, sweep beside the evaluation {b:.2f} (+{b - a:.2f}); TV descent {c:.2f}, beside the evaluation {d:.2f} (+{d - c:.2f}); evaluation alone {e:.2f}")
