#!/usr/bin/env python3
"""Time the SLAB-SHARDED tv_fgp per inner iteration on one GPU: a world-1 RCCL group with every exchange issued (self-sends), two
iterations per pass and exchange (k_fgp_fused2 on slabs, round 6) against one (k_fgp_fused).  Run as
    MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python3 tools/bench_fgp_sharded.py [--nslice 128]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from tomo_tv_amd.engine import multigpuengine
from tomo_tv_amd.phantom import ellipsoids
from tomo_tv_amd._lib import VOL_RECON
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=512)
ap.add_argument("--nslice", type=int, default=128)
ap.add_argument("--iters", type=int, default=21)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--json", action="store_true", help="one JSON line with the per-iteration times instead of the text")
a = ap.parse_args()
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
t = multigpuengine(a.nslice, a.n, np.deg2rad(np.array([-20.0, 35.0])), force_collectives=True)
import json
if not a.json:
    print("native collectives:", bool(t._native()))
x = ellipsoids(a.nslice, a.n)
out, us = {}, {True: [], False: []}
for pair in (True, False, True, False):
    t.fgp_pair = pair
    t.set_volume(x, VOL_RECON); t.tv_fgp(3, 0.1); t.synchronize()
    t.set_volume(x, VOL_RECON)
    t0 = time.perf_counter()
    for _ in range(a.reps):
        t.tv_fgp(a.iters, 0.1)
    t.synchronize()
    ms = (time.perf_counter() - t0) / a.reps * 1e3
    out[pair] = t.get_volume()
    us[pair].append(ms / a.iters * 1e3)
    if not a.json:
        print(f"sharded, fgp_pair={int(pair)}: tv_fgp({a.iters}) {ms:.2f} ms = {ms / a.iters * 1e3:.0f} us per iteration (incl. the TV value, the exchanges and the final pass)")
same = bool(np.array_equal(out[True].view(np.uint32), out[False].view(np.uint32)))
if a.json:
    print(json.dumps({"slab": f"{a.nslice}x{a.n}x{a.n}", "iterations": a.iters, "us_per_iteration_two_per_pass": min(us[True]),
                      "us_per_iteration_one_per_pass": min(us[False]), "bit_identical": same, "native_collectives": bool(t._native())}))
else:
    print("pair == one-per-pass, bit for bit:", same)
dist.destroy_process_group()
