#!/usr/bin/env python3
"""Time ChemicalTomo's data-fusion iteration (config 5 shape per GPU) through the public multimodal API."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomo_tv_amd.chemistry import multimodal, create_weighted_summation_weights
from tomo_tv_amd.phantom import ellipsoids, tilt_angles

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=512)
ap.add_argument("--nslice", type=int, default=512)
ap.add_argument("--nproj", type=int, default=70)
ap.add_argument("--nel", type=int, default=2)
ap.add_argument("--iters", type=int, default=3)
a = ap.parse_args()
ang = np.deg2rad(tilt_angles(a.nproj))
mm = multimodal(a.nslice, a.n, a.nel, ang, ang)
mm.set_gamma(1.6)
mm.set_weights(create_weighted_summation_weights([30, 8, 16][:a.nel], 1.6, 3))
gt = np.stack([ellipsoids(a.nslice, a.n, seed=5 + e) * (0.5 + 0.3 * e) for e in range(a.nel)])
# synthetic measurements through the engine's own operators
mm.set_volume(gt)
mm._mm_model()
mm.he.be.c("forward_projection", mm.MODEL, 0)
bh = mm.he.get_projections(); mm.set_haadf_tilt_series(bh / bh.max())
for e in range(a.nel):
    mm.ce.be.c("forward_projection", int(mm._x[e]), int(mm._b[e]))
mm.restart_recon()
mm.set_measureChem(True); mm.set_measureHaadf(True)
for _ in range(3):
    mm.poisson_ml(0.05)
mm.rescale_tomograms(10); mm.rescale_projections()
mm.sirt_data_fusion(10, 0.05, 5); mm.tv_fgp_4D(5, 1e-4)
mm.ce.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    h, c = mm.sirt_data_fusion(10, 0.05, 5)
    tv = mm.tv_fgp_4D(5, 1e-4)
mm.ce.synchronize()
el = (time.perf_counter() - t0) / a.iters
print(f"data_fusion {a.nel} elements {a.nslice}x{a.n}x{a.n} P={a.nproj}: {1 / el:.2f} it/s, {el * 1e3:.1f} ms/iter (iterSIRT=5, tvIter=5); costs {h:.4g} {c:.4g} {tv:.4g}")
