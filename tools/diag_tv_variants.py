#!/usr/bin/env python3
"""Which arithmetic of the TV descent is nearest the exact (binary64) trajectory at 512^3?  (round 4)

usage: python tools/diag_tv_variants.py [Nx] lib1.so [lib2.so ...]      (parent: oracle; one child process per library)
       python tools/diag_tv_variants.py --child lib.so start.npy dPOCS eps out_prefix

The parent makes the start volume the way tests/test_gpu_baseline_parity.py does (oracle SART sweep from zero on the noise-free
tilt series of the ellipsoid phantom, 512^2 x 90), evaluates tv_gd(5) and tv_gd(10) with the oracle in fp32 and in binary64, and
lets every library variant (built with different -D switches) run the same descents through the C ABI in a process of its own.
Prints, per variant: relative L2 against the oracle (5 and 10 steps), against binary64 (10 steps), next to the oracle's own
distance from binary64 and its response to a start moved by one ulp per voxel.
"""
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rel(a, b):
    return float(np.linalg.norm((a.astype(np.float64) - b).ravel()) / np.linalg.norm(b.astype(np.float64).ravel()))


def child(lib, start_path, dPOCS, eps, out_prefix):
    from tomo_tv_amd import _lib
    _lib.LIB_PATH = os.path.abspath(lib)
    from tomo_tv_amd.engine import tomoengine
    from tomo_tv_amd.phantom import tilt_angles
    start = np.load(start_path)
    nx, n, _ = start.shape
    dev = tomoengine(nx, n, np.deg2rad(tilt_angles(8)))     # the TV calls do not touch the geometry: a small one
    dev.tv_eps = eps
    for ng in (5, 10):
        dev.set_volume(start)
        tv = dev.tv_gd(ng, dPOCS)
        np.save(f"{out_prefix}_{ng}.npy", dev.get_volume())
        # time one more call of the same length (kernels warm)
        dev.set_volume(start)
        dev.tomo_sync() if hasattr(dev, "tomo_sync") else None
        t = time.time()
        dev.tv_gd(ng, dPOCS)
        dev.get_recon(0)
        print(f"child {os.path.basename(lib)}: tv_gd({ng}) tv0 {tv:.6e}  wall {1e3 * (time.time() - t):.1f} ms", flush=True)


def main():
    args = sys.argv[1:]
    if args and args[0] == "--child":
        child(args[1], args[2], float(args[3]), float(args[4]), args[5])
        return
    nx = 512
    if args and args[0].isdigit():
        nx = int(args.pop(0))
    firsts = [None]
    if args and args[0].startswith("--firsts="):          # several slabs of the phantom: a sample of the ratio, not one draw
        firsts = [int(v) for v in args.pop(0).split("=")[1].split(",")]
    libs = args
    ratios = {os.path.basename(l): [] for l in libs}
    for first in firsts:
        run(nx, first, libs, ratios)
    if len(firsts) > 1:
        for k, v in ratios.items():
            v = np.array(v)
            print(f"{k:32s} distance to binary64 / the oracle's: mean {v.mean():.3f}  std {v.std(ddof=1):.3f}  min {v.min():.3f}  max {v.max():.3f}  (n = {v.size})")


def run(nx, first, libs, ratios):
    import oracle
    from tomo_tv_amd.phantom import ellipsoids, tilt_angles
    n, p, eps = 512, 90, 1e-6
    oracle.set_num_threads(oracle.usable_cpus())
    t0 = time.time()
    x = ellipsoids(512, n, first=(256 - nx // 2) if first is None else first, count=nx)
    ref = oracle.ctvlib(nx, n, p)
    ref.load_A(oracle.parallel_ray(n, tilt_angles(p)))
    ref.tv_eps = eps
    ref.original_volume = x
    ref.create_projections()
    ref.set_tilt_series(ref.b.copy())
    ref.original_volume = None
    ref.recon[:] = 0
    ref.copy_recon()
    ref.SART(0.25, 1)
    dp = ref.matrix_2norm()
    dPOCS = 0.2 * dp
    start = ref.recon.copy()
    tmp = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    sp = os.path.join(tmp, "tvdiag_start.npy")
    np.save(sp, start)
    print(f"Nx={nx} first={first}: start made in {time.time() - t0:.0f} s (oracle, {oracle.num_threads()} threads); dp {dp:.6e}", flush=True)
    o = {}
    for ng in (5, 10):
        ref.recon[:] = start
        ref.tv_gd(ng, dPOCS)
        o[ng] = ref.recon.copy()
    exact = ref.tv_gd_f64(10, dPOCS, start=start)
    selfs = []
    for seed in range(2):
        rng = np.random.default_rng(seed)
        ref.recon[:] = np.nextafter(start, np.where(rng.random(start.shape) < 0.5, -np.inf, np.inf).astype(np.float32))
        ref.tv_gd(10, dPOCS)
        selfs.append(rel(ref.recon, o[10]))
    e_ref = rel(o[10], exact)
    print(f"oracle: tv_gd(10) vs binary64 {e_ref:.3e}; vs itself from a +-1 ulp start {' '.join(f'{s:.2e}' for s in selfs)}", flush=True)
    for lib in libs:
        pre = os.path.join(tmp, "tvdiag_out")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib, sp, repr(dPOCS), repr(eps), pre],
                           capture_output=True, text=True)
        sys.stdout.write(r.stdout)
        if r.returncode:
            print(f"{lib}: child failed\n{r.stderr[-2000:]}")
            continue
        g5, g10 = np.load(pre + "_5.npy"), np.load(pre + "_10.npy")
        e_dev = rel(g10, exact)
        print(f"{os.path.basename(lib):32s} tv_gd(5) vs oracle {rel(g5, o[5]):.3e}   tv_gd(10) vs oracle {rel(g10, o[10]):.3e}   "
              f"vs binary64 {e_dev:.3e}  = {e_dev / e_ref:.2f} x the oracle's distance", flush=True)
        ratios[os.path.basename(lib)].append(e_dev / e_ref)
        os.remove(pre + "_5.npy")
        os.remove(pre + "_10.npy")
    os.remove(sp)


if __name__ == "__main__":
    main()
