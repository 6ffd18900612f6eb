#!/usr/bin/env python3
"""Does the data distance on the second stream run under the TV descent?  Intervals of k_fp_tile vs the TV kernels, for the
whole-volume engine and for the slab-sharded engine (forced collectives at world 1).  python tools/exp_dd_overlap.py [nslice]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
from tomo_tv_amd import _lib
from tomo_tv_amd.engine import tomoengine, multigpuengine
from tomo_tv_amd.phantom import ellipsoids, tilt_angles
import bench

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n, P = 512, 90
ang = np.deg2rad(tilt_angles(P))

def intervals(t, kid_id):
    be = t.be; L = be.L; nn = ctypes.c_int(0)
    _lib.check(L.tomo_profile_intervals(be.h, kid_id, be.h, None, None, 0, ctypes.byref(nn)))
    t0, t1 = np.zeros(max(nn.value, 1)), np.zeros(max(nn.value, 1))
    _lib.check(L.tomo_profile_intervals(be.h, kid_id, be.h, t0.ctypes.data_as(ctypes.c_void_p), t1.ctypes.data_as(ctypes.c_void_p), nn.value, ctypes.byref(nn)))
    return list(zip(t0[:nn.value], t1[:nn.value]))

def run(t, tag):
    t.set_tilt_series(np.random.default_rng(0).random((ns, n * P), dtype=np.float32))
    t.initialize_SART("sequential")
    st = {"i": 0, "beta": 0.25, "dPOCS": 0.0, "norm": 1.0}
    for _ in range(2): bench.asd_pocs_step(t, st)
    ids = [_lib.K_FP_TILE, _lib.K_FP_REDUCE, 2, 3]
    for k in ids: _lib.check(t.be.L.tomo_profile_enable(t.be.h, k, 1))
    import time
    t.synchronize(); t0 = time.perf_counter()
    for _ in range(4): bench.asd_pocs_step(t, st)
    t.synchronize(); ms = (time.perf_counter() - t0) / 4 * 1e3
    iv = {k: intervals(t, k) for k in ids}
    dd = iv[ids[0]] + iv[ids[1]]; tv = iv[2] + iv[3]
    ov = sum(max(0.0, min(b, d) - max(a, c)) for a, b in dd for c, d in tv)
    print(f"{tag}: {ms:.2f} ms/step; dd kernels {sum(b - a for a, b in dd) / 4:.3f} ms/step, TV kernels {sum(b - a for a, b in tv) / 4:.3f} ms/step, overlapped {ov / 4:.3f} ms/step")

run(tomoengine(ns, n, ang), "whole-volume engine")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
run(multigpuengine(ns, n, ang, force_collectives=True), "slab engine, forced collectives")
dist.destroy_process_group()
