#!/usr/bin/env python3
"""Experiment: one 512-slice engine vs two 256-slice engines on their own streams (slices are independent): does running
two SART chains side by side hide the launch gaps and kernel tails of the dependent chain?"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomo_tv_amd.engine import tomoengine
from tomo_tv_amd.phantom import ellipsoids, tilt_angles
from tomo_tv_amd._lib import VOL_ORIGINAL

n, P = 512, 90
ang = np.deg2rad(tilt_angles(P))

def make(nx):
    t = tomoengine(nx, n, ang)
    t.set_volume(ellipsoids(nx, n), VOL_ORIGINAL)
    t.create_projections()
    t.initialize_SART("sequential")
    t.SART(0.5, 1); t.synchronize()
    return t

def run(engs, reps=5):
    def body(t):
        for _ in range(reps):
            t.SART(0.5, 1)
        t.synchronize()
    ths = [threading.Thread(target=body, args=(t,)) for t in engs]
    t0 = time.perf_counter()
    for th in ths: th.start()
    for th in ths: th.join()
    return (time.perf_counter() - t0) / reps * 1e3

one = make(512)
print("one engine, 512 slices: %.2f ms per sweep" % run([one]))
del one
for k in (2, 4):
    engs = [make(512 // k) for _ in range(k)]
    print("%d engines x %d slices, own streams: %.2f ms per sweep (all)" % (k, 512 // k, run(engs)))
    print("   the same engines one after the other: %.2f ms" % sum(run([e]) for e in engs))
    del engs
