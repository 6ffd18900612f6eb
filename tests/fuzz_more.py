#!/usr/bin/env python3
"""Open-ended random-shape parity sweep on the GPU (a superset of tests/test_gpu_fuzz.py): fuzz_more.py SEED CASES.
Every kernel family against the oracle at random (N, P, Nx, angles); prints the stage reached before each call so a
device fault names its case."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle
import torch
torch.cuda.init()          # before the library touches the device (torch started second found "No HIP GPUs" on this image)
from local_ring import ThreadRing
from tomo_tv_amd._lib import VOL_ORIGINAL, VOL_RECON
from tomo_tv_amd.engine import ctvlib, system_matrix, tomoengine
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
    # the FISTA driver loop (gpu/reconstructor.py:121-155): gradient step on yk, FGP prox, momentum, cost, the next A yk by linearity
    print(' fista', flush=True)
    from tomo_tv_amd import pytvlib
    from tomo_tv_amd._lib import VOL_YK
    dev.restart_recon(); ref.restart_recon()
    pytvlib.initialize_algorithm(dev, "fista"); ref.initialize_fista()
    t0 = 1.0
    for k in range(4):
        pytvlib.run(dev, "fista"); dev.tv_fgp(3, 0.05, vol=VOL_YK)
        ref.SIRT_norm(1, target="yk"); ref.recon, ref.yk = ref.yk, ref.recon; ref.tv_fgp(3, 0.05); ref.recon, ref.yk = ref.yk, ref.recon
        tk = 0.5 * (1 + np.sqrt(1 + 4 * t0 ** 2)); dev.fista_momentum((t0 - 1) / tk); ref.fista_momentum((t0 - 1) / tk); t0 = tk
        cd = 0.5 * dev.data_distance() ** 2 + 0.05 * dev.tv(); cr = 0.5 * ref.data_distance(normalize=False) ** 2 + 0.05 * ref.tv()
        took = dev.fista_project_yk()
        e.append(abs(cd - cr) / max(cr, 1e-30)); e.append(0.0 if took else 1.0)
    e.append(rel(dev.get_volume(), ref.recon)); e.append(rel(dev.get_volume(VOL_YK), ref.yk))
    dev.remove_momentum()
    # two-stream SART == one chain (bitwise where the sub-slabs keep the slab's vector width, else an ulp per step)
    v0 = None
    dev.set_option("sart_resident", 0)
    for ns in (1, 2):
        dev.set_option("sart_streams", ns); dev.restart_recon(); dev.SART(0.7, 1); v = dev.get_volume()
        if v0 is None: v0 = v
    dev.set_option("sart_streams", 1); dev.set_option("sart_resident", -1)
    e.append(0.0 if np.array_equal(v0, v) else rel(v, v0))
    # ART (chained, segmented scan) and the Cimmino branch through the ctvlib facade
    print(' art', flush=True)
    c = ctvlib(Nx, N, P); c.load_A(system_matrix(N, ang)); c.set_tilt_series(ref.b)
    ref.row_inner_product(); ref.restart_recon(); ref.ART(0.5); ref.ART(0.45)
    c.row_inner_product(); c.ART(0.5); c.ART(0.45); e.append(rel(c.get_volume(), ref.recon))
    # slab-sharded TV / FGP on 2 or 3 slabs of this GPU (thread ring) against the whole slab
    if Nx >= 3:
        print(' sharded', flush=True)
        world = int(rng.integers(2, 4))
        xs = x + np.float32(0.03) * rng.random(x.shape, dtype=np.float32)
        def script(t):
            t.set_volume(xs, VOL_RECON); t.tv_eps = 1e-6
            a = t.tv_gd(2, 0.05); g1 = t.get_volume(); t.set_volume(xs, VOL_RECON); b = t.tv_fgp(3, 0.03); return a, g1, b, t.get_volume()
        whole = tomoengine(Nx, N, ang*np.pi/180); w = script(whole)
        got = ThreadRing(world).run(lambda comm: script(tomoengine(Nx, N, ang*np.pi/180, device=0, comm=comm)))[0]
        e += [abs(got[0]-w[0])/w[0], rel(got[1], w[1]), abs(got[2]-w[2])/w[2], 0.0 if np.array_equal(got[3], w[3]) else rel(got[3], w[3]) + 1e-5]
    ok = max(e) < 1e-5 and np.isfinite(dev.get_volume()).all()
    bad += (not ok)
    print(f"N={N} P={P} Nx={Nx}: max err {max(e):.2e} {'ok' if ok else 'FAIL ' + str(e)}", flush=True)
print("FAILED" if bad else "ALL OK", bad)
