#!/usr/bin/env python3
"""Upper bound for a slab engine made of K independent sub-engines: K engines of 512/K slices each run the whole ASD-POCS
step loop (uncoupled: no TV halo exchange between them) on K Python threads, against one 512-slice engine."""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import asd_pocs_step
from tomo_tv_amd.engine import tomoengine
from tomo_tv_amd.phantom import ellipsoids, tilt_angles
from tomo_tv_amd._lib import VOL_ORIGINAL

n, P = 512, 90
ang = np.deg2rad(tilt_angles(P))

def make(nx, sub=1):
    t = tomoengine(nx, n, ang, sub_slabs=sub)
    t.set_volume(ellipsoids(nx, n), VOL_ORIGINAL)
    t.create_projections()
    t.initialize_SART("sequential")
    t.restart_recon()
    st = {"beta": 0.25, "i": 0, "dPOCS": 0.0, "norm": float(nx * n * P)}
    asd_pocs_step(t, st); t.synchronize()
    return t, st

def run(engs, reps=5):
    def body(t, st):
        for _ in range(reps):
            asd_pocs_step(t, st)
        t.synchronize()
    ths = [threading.Thread(target=body, args=e) for e in engs]
    t0 = time.perf_counter()
    for th in ths: th.start()
    for th in ths: th.join()
    return (time.perf_counter() - t0) / reps * 1e3

one = make(512)
print("one engine, 512 slices: %.2f ms per ASD-POCS step" % run([one]))
del one
for sub in (2, 2, 4):
    grp = make(512, sub)
    print("one engine, 512 slices as %d coupled sub-slabs (sub_slabs=%d): %.2f ms per step" % (sub, sub, run([grp])))
    del grp
for k in (2,):
    engs = [make(512 // k) for _ in range(k)]
    print("%d engines x %d slices on %d threads: %.2f ms per step (all 512 slices)" % (k, 512 // k, k, run(engs)))
    print("   the same engines one after the other: %.2f ms" % sum(run([e]) for e in engs))
