#!/usr/bin/env python3
"""Does a SART sweep run faster when its working set fits the 256 MB Infinity Cache?  The 512 slices as K independent engines of
512/K slices (slices are independent in SART): all K sweeps at once on K threads, against G engines at a time (their slabs then
total 512/K*G slices: 134 MB per 128 slices), against one 512-slice engine (two chains of 256 slices)."""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomo_tv_amd.engine import tomoengine
from tomo_tv_amd.phantom import ellipsoids, tilt_angles
from tomo_tv_amd._lib import VOL_ORIGINAL

n, P = 512, 90
ang = np.deg2rad(tilt_angles(P))

def make(nx, first=0):
    t = tomoengine(nx, n, ang)
    t.set_volume(ellipsoids(512, n, first=first, count=nx), VOL_ORIGINAL)
    t.create_projections()
    t.initialize_SART("sequential")
    t.restart_recon()
    t.SART(0.25, 1); t.synchronize()
    return t

def sweep_group(engs, reps):
    def body(t):
        for _ in range(reps):
            t.SART(0.25, 1)
        t.synchronize()
    ths = [threading.Thread(target=body, args=(e,)) for e in engs]
    for th in ths: th.start()
    for th in ths: th.join()

def run(engs, at_a_time, reps=3):
    t0 = time.perf_counter()
    for i in range(0, len(engs), at_a_time):
        sweep_group(engs[i:i + at_a_time], reps)
    return (time.perf_counter() - t0) / reps * 1e3

one = make(512)
print("one engine, 512 slices (two chains of 256): %.2f ms per sweep" % run([one], 1))
del one
for k in (4, 8):
    nx = 512 // k
    engs = [make(nx, i * nx) for i in range(k)]
    for g in sorted({1, 2, 4, k}):
        if g <= k:
            print("%d engines x %d slices, %d at a time (%d MB in flight): %.2f ms per sweep of all 512 slices"
                  % (k, nx, g, g * nx * n * n * 4 // 2**20, run(engs, g)))
    del engs
