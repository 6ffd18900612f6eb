#!/usr/bin/env python3
"""Time the pieces of bench.py's asd_pocs_art secondary config (cpu/sim_ASD.py:64-96 through the ctvlib facade) one by one."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomo_tv_amd import pytvlib
from tomo_tv_amd.engine import ctvlib
from tomo_tv_amd.phantom import ellipsoids, tilt_angles
c = ctvlib(512, 512, 90)
c.load_A(pytvlib.parallelRay(512, tilt_angles(90)))
c.set_volume(ellipsoids(512, 512), 2)
c.create_projections()
c.tv_eps = 1e-6
def T(name, fn, n=3):
    fn(); c.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    c.synchronize()
    print(f"{name:28s} {(time.perf_counter() - t0) / n * 1e3:8.2f} ms", flush=True)
T("copy_recon", c.copy_recon)
T("ART(0.5)", lambda: c.ART(0.5))
T("matrix_2norm", c.matrix_2norm)
T("data_distance", c.data_distance)
T("tv", c.tv)
T("tv_gd(10, 5.0)", lambda: c.tv_gd(10, 5.0))
def step():
    c.copy_recon(); c.ART(0.5); dp = c.matrix_2norm(); dd = c.data_distance(); c.copy_recon(); c.tv(); c.tv_gd(10, dp * 0.2); c.matrix_2norm()
T("whole step", step)
