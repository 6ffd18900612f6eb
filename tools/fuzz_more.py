#!/usr/bin/env python3
"""Open-ended random-shape parity sweep on the GPU (a superset of tests/test_gpu_fuzz.py): fuzz_more.py SEED CASES.
Every kernel family against the oracle at random (N, P, Nx, angles); prints the stage reached before each call so a
device fault names its case."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from tomo_tv_amd._lib import VOL_ORIGINAL
from tomo_tv_amd.engine import tomoengine
from tomo_tv_amd.phantom import ellipsoids
def rel(a,b): return float(np.linalg.norm(a.astype(np.float64)-b)/max(np.linalg.norm(b),1e-30))
rng0 = np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 777)
bad = 0
for case in range(int(sys.argv[2]) if len(sys.argv)>2 else 30):
    N = int(rng0.integers(3, 140)); P = int(rng0.integers(1, 30)); Nx = int(rng0.integers(1, 330)); seed = int(rng0.integers(0, 10**6))
    print(f'case {case}: N={N} P={P} Nx={Nx} seed={seed}', flush=True)
    rng = np.random.default_rng(seed)
    ang = np.sort(rng.uniform(-89.9, 89.9, P))
    x = ellipsoids(Nx, N, seed=seed % 1000, k=4)
    ref = oracle.ctvlib(Nx, N, P); ref.load_A(oracle.parallel_ray(N, ang)); ref.original_volume = x.copy(); ref.create_projections()
    dev = tomoengine(Nx, N, ang*np.pi/180); dev.set_volume(x, VOL_ORIGINAL); print(' fp', flush=True); dev.create_projections(); dev.synchronize()
    e = [rel(dev.get_projections(), ref.b)]
    dev.copy_recon(); ref.copy_recon()
    print(' sart', flush=True); dp = dev.SART_tracked(0.7, 2); dev.synchronize(); ref.SART(0.7, 2); dpr = ref.matrix_2norm(); ref.copy_recon()
    e.append(rel(dev.get_volume(), ref.recon)); e.append(abs(dp-dpr)/max(dpr,1e-30))
    print(' sirt', flush=True); dev.SIRT(2); dev.synchronize(); ref.SIRT_norm(2); e.append(rel(dev.get_volume(), ref.recon))
    ref.tv_eps = dev.tv_eps
    print(' tv', flush=True); tv, dg = dev.tv_gd_tracked(2, 0.01); dev.synchronize(); tvr = ref.tv_gd(2, 0.01); e.append(abs(tv-tvr)/tvr); e.append(rel(dev.get_volume(), ref.recon))
    print(' fgp', flush=True); a = dev.tv_fgp(3, 0.02); dev.synchronize(); b = ref.tv_fgp(3, 0.02); e.append(abs(a-b)/b); e.append(rel(dev.get_volume(), ref.recon))
    print(' cgls', flush=True); dev.CGLS(1); dev.synchronize()
    ok = max(e) < 1e-5 and np.isfinite(dev.get_volume()).all()
    bad += (not ok)
    print(f"N={N} P={P} Nx={Nx}: max err {max(e):.2e} {'ok' if ok else 'FAIL ' + str(e)}", flush=True)
print("FAILED" if bad else "ALL OK", bad)
