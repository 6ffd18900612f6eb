#!/usr/bin/env python3
"""Where does one ASD-POCS iteration at 512^2 x 90 leave the oracle?  (round 3 diagnosis; prints a table)

usage: python tools/diag_tv_fullsize.py [Nx] [lib.so]
Stages, each from IDENTICAL inputs on both sides: SART sweep from zero; tv_gd(ng) for ng = 1, 2, 5, 10 started from the oracle's
SART result; the oracle against itself from a start moved by one ulp per voxel (8 seeds); both against an fp64 evaluation of the
same descent (numpy) -- which of the two fp32 paths is nearer the exact arithmetic.
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 64
if len(sys.argv) > 2:
    from tomo_tv_amd import _lib
    _lib.LIB_PATH = sys.argv[2]
import oracle  # noqa: E402
from tomo_tv_amd._lib import VOL_ORIGINAL  # noqa: E402
from tomo_tv_amd.engine import tomoengine  # noqa: E402
from tomo_tv_amd.phantom import ellipsoids, tilt_angles  # noqa: E402

n, p = 512, 90
eps = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-6


def rel(a, b):
    return float(np.linalg.norm((a.astype(np.float64) - b).ravel()) / np.linalg.norm(b.astype(np.float64).ravel()))


def tvgd64(x, ng, dPOCS, eps):
    x = x.astype(np.float64)
    for _ in range(ng):
        ip, jp, kp = np.roll(x, -1, 0), np.roll(x, -1, 1), np.roll(x, -1, 2)
        D = np.sqrt(eps + (x - ip) ** 2 + (x - jp) ** 2 + (x - kp) ** 2)
        R = 1.0 / D
        g = (3 * x - ip - jp - kp) * R + (x - np.roll(x, 1, 0)) * np.roll(R, 1, 0) + (x - np.roll(x, 1, 1)) * np.roll(R, 1, 1) \
            + (x - np.roll(x, 1, 2)) * np.roll(R, 1, 2)
        x = x - dPOCS * g / np.sqrt((g * g).sum())
    return np.maximum(x, 0)


x = ellipsoids(512, n, first=256 - nx // 2, count=nx)
dev = tomoengine(nx, n, np.deg2rad(tilt_angles(p)))
dev.tv_eps = eps
dev.set_volume(x, VOL_ORIGINAL)
dev.create_projections()
b = dev.get_projections()
oracle.set_num_threads(oracle.usable_cpus())
ref = oracle.ctvlib(nx, n, p)
ref.load_A(oracle.parallel_ray(n, tilt_angles(p)))
ref.tv_eps = eps
ref.set_tilt_series(b)
ref.copy_recon()
ref.SART(0.25, 1)
dev.copy_recon()
dev.SART(0.25, 1)
dp = ref.matrix_2norm()
print(f"Nx={nx} eps={eps:g}: SART sweep rel-L2 {rel(dev.get_volume(), ref.recon):.2e}; dp {dp:.6e} / {dev.matrix_2norm():.6e}")
start = ref.recon.copy()
dPOCS = 0.2 * dp
for ng in (1, 2, 5, 10):
    ref.recon[:] = start
    dev.set_volume(start)
    tv_r, tv_d = ref.tv_gd(ng, dPOCS), dev.tv_gd(ng, dPOCS)
    got = dev.get_volume()
    t = time.time()
    x64 = tvgd64(start, ng, dPOCS, eps)
    line = f"tv_gd({ng:2d}): HIP vs oracle {rel(got, ref.recon):.2e}   HIP vs fp64 {rel(got, x64):.2e}   oracle vs fp64 {rel(ref.recon, x64):.2e}"
    if ng == 10:
        selfs = []
        base = ref.recon.copy()
        for seed in range(4):
            rng = np.random.default_rng(seed)
            ref.recon[:] = np.nextafter(start, np.where(rng.random(start.shape) < 0.5, -np.inf, np.inf).astype(np.float32))
            ref.tv_gd(ng, dPOCS)
            selfs.append(rel(ref.recon, base))
        line += "   oracle vs itself (+-1 ulp start): " + " ".join(f"{s:.2e}" for s in selfs)
    print(line, f"  tv0 {tv_d:.6e}/{tv_r:.6e}", flush=True)
