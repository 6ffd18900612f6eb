"""One rank of a world-N RCCL run (started by tests/test_gpu_rccl_worldN.py as a fresh process per GPU, before anything in it
touches a GPU).  Every rank builds the slab-sharded engine with the GLOBAL sizes; rank 0 also reconstructs the whole volume on
one engine and compares.  Checks: one device per rank (LOCAL_RANK), sharded == single-engine ASD-POCS (the product's driver,
tracked sweep + asynchronous residual + deferred scalars) to 2e-6, the one-round TV descent protocol == the two-round form bit for
bit, fused FGP == the Obj / Grad pair, gathers.  Reference semantics: mpi_ctvlib.cpp:400-422 (ring), :455,547 (all-reduce)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

torch.cuda.set_device(local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

from tomo_tv_amd._lib import VOL_ORIGINAL  # noqa: E402
from tomo_tv_amd.engine import multigpuengine, tomoengine  # noqa: E402
from tomo_tv_amd.phantom import ellipsoids  # noqa: E402
from tomo_tv_amd.reconstructor import TomoGPU  # noqa: E402


def rel(a, b):
    return float(np.linalg.norm((a.astype(np.float64) - b).ravel()) / np.linalg.norm(b.astype(np.float64).ravel()))


N, P, Nx = 64, 15, 150                       # uneven slabs at world 3 (50 each) and 2 (75): 64-slice chunks + a ragged one
ang_deg = np.linspace(-70, 70, P)
x = ellipsoids(Nx, N, seed=7)
x += np.float32(0.05) * np.random.default_rng(3).random(x.shape, dtype=np.float32)   # no exact zeros: every slice couples in TV

# world 1 (the single-GPU boxes: a smoke of THIS script's logic) forces the real collectives in the one-rank group
t = multigpuengine(Nx, N, np.deg2rad(ang_deg), force_collectives=(world == 1))
assert t.is_multi_gpu_enabled() == (world > 1) and t.comm.world == world
ids = t.get_gpu_ids()
assert ids == list(range(world)) and t.get_gpu_id() == local, ids          # one device per rank, the rank's own
t.set_volume(x, VOL_ORIGINAL)
t.create_projections()
b = t.get_projections()                        # all_gather of the slabs' sinograms
assert b.shape == (Nx, N * P)

# the product's driver under a process group of more than one rank picks the sharded engine by itself (determine_gpu_config)
if world > 1:
    rec = TomoGPU(ang_deg, b.reshape(Nx, P, N).transpose(0, 2, 1))
    assert isinstance(rec.tomo, multigpuengine)
else:
    rec = TomoGPU.__new__(TomoGPU)
    rec.tomo, rec.verbose = t, False
    t.set_tilt_series(b)
dd, tv = rec.asd_pocs(Niter=3)
vol = rec.tomo.get_volume()
vol0 = rec.tomo.get_volume(dst=0)
assert (vol0 is None) == (rank != 0)

# the library's own RCCL groups (default) == the torch.distributed collectives; one communication round per TV step == two
# rounds, bit for bit; fused FGP == pair
assert t.be.native
outs = {}
for one_round in (True, False, "torch"):
    t.use_native_comm = one_round != "torch"
    t.tv_one_round = bool(one_round)
    t.set_volume(vol)
    t.copy_recon()
    tv0, dg = t.tv_gd_tracked(4, 0.3)
    outs[one_round] = (tv0, dg, t.get_volume())
t.use_native_comm = True
assert outs[True][0] == outs[False][0] and np.array_equal(outs[True][2], outs[False][2])
assert outs[True][0] == outs["torch"][0] and outs[True][1] == outs["torch"][1] and np.array_equal(outs[True][2], outs["torch"][2])
fg = {}
for fused in (True, False):
    t.fgp_fused = fused
    t.set_volume(vol)
    fg[fused] = (t.tv_fgp(4, 0.05), t.get_volume())
assert abs(fg[True][0] - fg[False][0]) <= 1e-6 * fg[False][0] and rel(fg[True][1], fg[False][1]) < 2e-6

if rank == 0:
    one = tomoengine(Nx, N, np.deg2rad(ang_deg), device=local)
    one.set_volume(x, VOL_ORIGINAL)
    one.create_projections()
    assert rel(b, one.get_projections()) < 1e-6
    single = TomoGPU.__new__(TomoGPU)          # the same driver on ONE engine holding the whole volume
    single.tomo, single.verbose = one, False
    one.set_tilt_series(b)
    dd1, tv1 = single.asd_pocs(Niter=3)
    e = rel(vol, one.get_volume())
    assert np.allclose(dd, dd1, rtol=5e-6) and np.allclose(tv, tv1, rtol=5e-6), (dd, dd1, tv, tv1)
    assert e < 2e-6, e
    assert np.array_equal(vol0, vol)
    one.set_volume(vol)
    one.copy_recon()
    tv0, dg = one.tv_gd_tracked(4, 0.3)
    assert abs(tv0 - outs[True][0]) <= 2e-6 * tv0 and rel(outs[True][2], one.get_volume()) < 2e-6
    one.set_volume(vol)
    tvf = one.tv_fgp(4, 0.05)
    assert abs(tvf - fg[True][0]) <= 2e-6 * tvf and rel(fg[True][1], one.get_volume()) < 2e-6
    print(f"RCCL_WORLD{world}_OK volume {e:.2e}", flush=True)
dist.barrier()
dist.destroy_process_group()
