#!/usr/bin/env python3
"""bench.py's ART-form ASD-POCS config, alone and behind the configs that precede it there (what made it 28 ms instead of 23?)."""
import os, sys, time, gc
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tomo_tv_amd import pytvlib
from tomo_tv_amd.engine import ctvlib
from tomo_tv_amd.phantom import ellipsoids, tilt_angles

def art():
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
    for rep in range(3):
        print("  ART-form step: %.2f ms" % bench._time_steps(c, asd_art, 3), flush=True)
print("alone"); art()
which = sys.argv[1] if len(sys.argv) > 1 else "sart"
t = bench._engine(512, 512, 90)
if which == "sart":
    t.initialize_SART("sequential"); t.SART(0.5, 1); t.synchronize(); print("after a resident SART sweep on another engine (kept alive)")
    art()
del t; gc.collect()
print("after that engine is gone"); art()
