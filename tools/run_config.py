#!/usr/bin/env python3
"""Run one of the BASELINE.json configurations through the public API and print iterations/s (use under rocprofv3)."""
import argparse, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomo_tv_amd import _lib as _tl
if os.environ.get("TOMO_LIB"):                      # A/B against another build of the library (tools only)
    _tl.LIB_PATH = os.path.abspath(os.environ["TOMO_LIB"])
from tomo_tv_amd.engine import tomoengine
from tomo_tv_amd.phantom import ellipsoids, tilt_angles
from tomo_tv_amd._lib import VOL_ORIGINAL, VOL_RECON, VOL_YK
from tomo_tv_amd import pytvlib

ap = argparse.ArgumentParser()
ap.add_argument("--alg", default="fista", choices=["fista", "sirt", "sart", "kl", "art", "cgls"])
ap.add_argument("--n", type=int, default=512)
ap.add_argument("--nslice", type=int, default=512)
ap.add_argument("--nproj", type=int, default=90)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--opt", action="append", default=[], help="engine option name=int, repeatable")
a = ap.parse_args()
t = tomoengine(a.nslice, a.n, np.deg2rad(tilt_angles(a.nproj)))
for kv in a.opt:
    k, v = kv.split("=")
    t.set_option(k, int(v))
t.set_volume(ellipsoids(a.nslice, a.n), VOL_ORIGINAL)
t.create_projections()
t.restart_recon()
if a.alg == "fista":
    pytvlib.initialize_algorithm(t, "fista")
elif a.alg == "kl":
    pytvlib.initialize_algorithm(t, "kl-divergence")
def step(k, st):
    if a.alg == "fista":
        pytvlib.run(t, "fista")
        t.tv_fgp(10, 0.1, vol=VOL_YK)
        tk = 0.5 * (1 + np.sqrt(1 + 4 * st["t0"] ** 2))
        t.fista_momentum((st["t0"] - 1) / tk)
        st["t0"] = tk
        c = 0.5 * t.data_distance() ** 2 + 0.1 * t.tv()
        t.fista_project_yk()            # as TomoGPU.fista: the next step's A yk from this A r and the last
        return c
    if a.alg == "sirt":
        t.SIRT(1); return t.data_distance()
    if a.alg == "sart":
        t.SART(1.0, 1); return t.data_distance()
    if a.alg == "art":
        t.be.c("art", 0.5); return t.data_distance()
    if a.alg == "cgls":
        t.CGLS(1); return t.data_distance()
    return t.poisson_ML(0.1)
st = {"t0": 1.0}
step(0, st); t.synchronize()
t0 = time.perf_counter()
for k in range(a.iters):
    c = step(k, st)
t.synchronize()
el = time.perf_counter() - t0
print(f"{a.alg} {a.nslice}x{a.n}x{a.n} P={a.nproj}: {a.iters / el:.2f} it/s, {el / a.iters * 1e3:.1f} ms/iter, cost {c:.6g}")
