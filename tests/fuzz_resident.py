#!/usr/bin/env python3
"""Random-shape parity sweep of the volume-resident SART sweep on the GPU (not a pytest file): fuzz_resident.py SEED CASES [NMAX].
Image sides that are multiples of 8 up to 256 (1 ... 64 tiles: several chunk groups side by side down to the full chip), 1 ... 40
angles over random ranges, ragged slice counts, sequential and random angle order, one to three sweeps per call, tracked or not --
against the oracle (<= 2e-6) and against the streamed chain of the same engine (<= 1e-6).  Round 6: a third of the cases make the
resident launch FAIL on purpose (every wait giving up at its first look, or one random chunk refusing to commit): the call must
still return the sweep (the streamed chain redoes what did not commit) and count the fallback."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from tomo_tv_amd._lib import VOL_ORIGINAL, VOL_RECON
from tomo_tv_amd.engine import tomoengine
from tomo_tv_amd.phantom import ellipsoids


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30))


rng0 = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 2025)
NMAX = int(sys.argv[3]) if len(sys.argv) > 3 else 256          # largest image side (512: up to the full chip, one tile per CU)
bad = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    N = 8 * int(rng0.integers(1, NMAX // 8 + 1)); P = int(rng0.integers(1, 41)); Nx = int(rng0.integers(1, 300 if N <= 256 else 100)); seed = int(rng0.integers(0, 10**6))
    niter = int(rng0.integers(1, 4)); beta = float(rng0.uniform(0.1, 1.0)); order = "random" if rng0.random() < 0.4 else "sequential"
    amax = float(rng0.uniform(20, 89.9))
    rng = np.random.default_rng(seed)
    ang = np.sort(rng.uniform(-amax, amax, P))
    x = ellipsoids(Nx, N, seed=seed % 1000, k=4)
    ref = oracle.ctvlib(Nx, N, P); ref.load_A(oracle.parallel_ray(N, ang)); ref.original_volume = x.copy(); ref.create_projections()
    ref.b = (ref.b * (1.0 + 0.03 * rng.standard_normal(ref.b.shape)) + 0.2).astype(np.float32)
    out = {}
    for resident in (1, 0):
        dev = tomoengine(Nx, N, ang * np.pi / 180)
        dev.set_tilt_series(ref.b)
        if resident and dev.get_option("sart_resident_ready") != 1:
            print(f"case {case}: N={N} P={P} Nx={Nx}: no resident tables", flush=True)
            break
        dev.set_option("sart_resident", resident)
        sabotage = None
        if resident:
            u = rng0.random()
            if u < 0.17:
                sabotage = "spin0"; dev.set_option("sart_resident_spin", 0)
            elif u < 0.34:
                sabotage = "chunk"; dev.set_option("sart_resident_test_fail", 1 + int(rng0.integers(0, (Nx + 63) // 64)))
        dev.initialize_SART(order)
        dev._order_rng = np.random.default_rng(seed)
        dev.copy_recon()
        dp = dev.SART_tracked(beta, niter)
        out[resident] = (dev.get_volume(VOL_RECON), dp)
        if resident:
            fb = dev.get_option("sart_resident_fallbacks")
            out["sab"] = (sabotage, fb)
            if (sabotage is not None) != (fb > 0):
                print(f"case {case}: sabotage {sabotage} but {fb} fallbacks", flush=True); bad += 1
        del dev
    if 0 not in out or 1 not in out:
        continue
    # the oracle with the engine's angle order of the LAST sweep is only defined for the sequential order; random orders are held to the streamed chain
    e_chain = rel(out[1][0], out[0][0]); e_dp = abs(out[1][1] - out[0][1]) / max(abs(out[0][1]), 1e-30)
    e_or = None
    if order == "sequential":
        ref.restart_recon(); ref.copy_recon(); ref.SART(beta, niter)
        e_or = rel(out[1][0], ref.recon)
    ok = e_chain <= 1e-6 and e_dp <= 1e-5 and (e_or is None or e_or <= 2e-6) and np.isfinite(out[1][0]).all()
    bad += 0 if ok else 1
    print(f"case {case}: N={N} P={P} Nx={Nx} niter={niter} beta={beta:.2f} {order} amax={amax:.0f} sabotage={out['sab'][0]}: vs chain {e_chain:.2e}, dp {e_dp:.1e}, vs oracle {e_or if e_or is None else format(e_or, '.2e')} {'ok' if ok else 'FAIL'}", flush=True)
print("FAILED" if bad else "all ok", bad)
sys.exit(1 if bad else 0)
