#!/usr/bin/env python3
"""Which of the configs that precede it in bench.py's secondary block makes the ART-form step 28 ms instead of 23?"""
import os, sys, time, gc
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tomo_tv_amd import pytvlib
from tomo_tv_amd._lib import VOL_YK
from tomo_tv_amd.engine import ctvlib
from tomo_tv_amd.phantom import ellipsoids, tilt_angles

def art(label):
    c = ctvlib(512, 512, 90)
    c.load_A(pytvlib.parallelRay(512, tilt_angles(90)))
    c.set_volume(ellipsoids(512, 512), 2)
    c.create_projections()
    c.tv_eps = 1e-6
    sa = {"beta": 0.5, "dPOCS": None}
    def asd_art():
        c.copy_recon(); c.ART(sa["beta"]); sa["beta"] *= 0.985
        dp = c.matrix_2norm()
        if sa["dPOCS"] is None: sa["dPOCS"] = dp * 0.2
        dd = c.data_distance(); c.copy_recon(); c.tv(); c.tv_gd(10, sa["dPOCS"]); dg = c.matrix_2norm()
        if dg > dp * 0.95 and dd > 0.02: sa["dPOCS"] *= 0.95
    print(f"{label}: ART-form step {bench._time_steps(c, asd_art, 3):.2f} ms", flush=True)
    def sweep(): c.ART(0.5)
    print(f"{label}:   ART sweep alone {bench._time_steps(c, sweep, 3):.2f} ms", flush=True)

art("fresh process")
t = bench._engine(256, 256, 60); t.initialize_SART("sequential")
bench._time_steps(t, lambda: (t.SART(1.0, 1), t.data_distance()), 5); del t; gc.collect()
art("after config 2 (256^3 SART)")
t = bench._engine(512, 512, 90)
pytvlib.initialize_algorithm(t, "fista")
def fista_iter():
    pytvlib.run(t, "fista"); t.tv_fgp(10, 0.1, vol=VOL_YK); t.fista_momentum(0.3); t.data_distance(); t.tv(); t.fista_project_yk()
bench._time_steps(t, fista_iter, 5)
art("after config 3 FISTA (engine alive)")
t.remove_momentum(); t.restart_recon()
bench._time_steps(t, lambda: (t.SIRT(1), t.data_distance()), 5)
art("after config 3 SIRT (engine alive)")
del t; gc.collect()
art("after config 3 engine is gone")
