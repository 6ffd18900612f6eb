"""Run by test_gpu_distributed.py in a child process: world_size-1 RCCL group exercising the torch plumbing of the
slab engine (torch-owned scalar/halo buffers bound into the C ABI, engine on torch's stream, all-reduce on the
device scalar between the TV gradient and update kernels).  The second sharded engine FORCES the real collectives in the
one-rank group (all_reduce, broadcast, batch_isend_irecv to itself, all_gather_into_tensor, gather): the RCCL calls, tensor
devices and stream ordering of the N > 1 path, on the only GPU a test box has.  Round 3: engines 2 and 3 run their collectives
NATIVELY (tomo_comm_*: ncclGroups issued by libtomo_hip.so on the engine's stream, self-sends in the one-rank group), engine 4
through torch.distributed; all four must agree."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from tomo_tv_amd.engine import multigpuengine, tomoengine  # noqa: E402
from tomo_tv_amd.phantom import ellipsoids  # noqa: E402
from tomo_tv_amd._lib import VOL_ORIGINAL  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
N, P, Nx = 32, 7, 70
ang = np.deg2rad(np.linspace(-60, 60, P))
x = ellipsoids(Nx, N, seed=5)
res = []
def torch_collectives():
    """the same sharded engine with its collectives through torch.distributed instead of the library's own RCCL groups"""
    t = multigpuengine(Nx, N, ang, force_collectives=True)
    assert t.be.native                      # the native communicator exists ...
    t.use_native_comm = False               # ... and is left unused
    return t


def folded_halo_advance():
    """the native path with the halo planes advanced INSIDE the update pass ("tv_halo_fold" = 1; automatic: slabs of two chunks or more)"""
    t = multigpuengine(Nx, N, ang, force_collectives=True)
    t.set_option("tv_halo_fold", 1)
    return t


def unfolded_halo_advance():
    t = multigpuengine(Nx, N, ang, force_collectives=True)
    t.set_option("tv_halo_fold", 0)
    return t


for make in (lambda: tomoengine(Nx, N, ang), lambda: multigpuengine(Nx, N, ang),
             lambda: multigpuengine(Nx, N, ang, force_collectives=True), torch_collectives, folded_halo_advance, unfolded_halo_advance):
    t = make()
    t.set_volume(x, VOL_ORIGINAL)
    t.create_projections()
    t.copy_recon()
    t.SART(0.5, 1)
    out = [t.matrix_2norm(), t.data_distance(), t.tv()]
    t.copy_recon()
    out.append(t.tv_gd(5, 0.2))
    out.append(t.matrix_2norm())
    v1 = t.get_volume()
    out.append(t.tv_fgp(4, 0.05))
    v2 = t.get_volume()
    out.append(t.rmse())
    # the ASD-POCS step with its batched scalar read (one all-reduce of a gathered tensor)
    from tomo_tv_amd._lib import S_DD, S_DIFF2
    t.copy_recon()
    t.SART_tracked(0.5, defer=True)
    t.data_distance_begin()
    out += list(t.tv_gd_tracked(3, 0.1, extra=(S_DD, S_DIFF2)))
    out.append(t.tv_fgp(1, 0.05))                       # single iteration: the Obj / Grad pair with its two exchanges
    res.append((np.array(out), v1, v2, t.get_recon(Nx - 1), t.get_volume(), t.get_projections()))
for other in res[1:]:
    a, b = res[0], other
    assert np.allclose(a[0], b[0], rtol=1e-6), (a[0], b[0])
    assert all(np.array_equal(p, q) for p, q in zip(a[1:], b[1:]))
# the fused multi-modal engine, slab-sharded on the same one-rank group (two native communicators: HAADF and chemical engine;
# tv_gd / tv_fgp per element through the library's own ring exchange) against the single-process class
from tomo_tv_amd.chemistry import create_weighted_summation_weights, multigpufusion, multimodal  # noqa: E402
ha, ca = np.deg2rad(np.linspace(-60, 60, 6)), np.deg2rad(np.linspace(-50, 55, 5))
w = create_weighted_summation_weights([30, 8], 1.6, 3)
gt = np.stack([ellipsoids(12, N, seed=3 + e, k=6) * np.float32(0.6 + 0.3 * e) for e in range(2)])
mm_out = []
for cls in (multimodal, multigpufusion):
    mm = cls(12, N, 2, ha, ca)
    if cls is multigpufusion:
        assert mm.ce.be.native and mm.he.be.native
    mm.set_gamma(1.6)
    mm.set_weights(w)
    mm.set_volume(gt)
    mm._mm_model()
    mm.he.be.c("forward_projection", mm.MODEL, 0)
    bh = mm.he.get_projections()
    for e in range(2):
        mm.ce.be.c("forward_projection", int(mm._x[e]), int(mm._b[e]))
    mm.set_haadf_tilt_series(bh / bh.max())
    mm.restart_recon()
    mm.set_measureChem(True)
    mm.set_measureHaadf(True)
    mm.estimate_lipschitz()
    c1 = [mm.poisson_ml(0.05) for _ in range(2)]
    mm.rescale_tomograms(10)
    mm.rescale_projections()
    c2 = mm.sirt_data_fusion(10, 0.05, 2)
    c3 = [mm.tv_fgp_4D(3, 1e-4), mm.tv_gd(2, 0.05)]
    mm_out.append((np.array(c1 + list(c2) + c3), mm.get_volume()))
assert np.allclose(mm_out[0][0], mm_out[1][0], rtol=1e-6), (mm_out[0][0], mm_out[1][0])
assert np.array_equal(mm_out[0][1], mm_out[1][1])
assert t.get_volume(dst=0).shape == (Nx, N, N)
assert t.is_multi_gpu_enabled() is False and t.get_gpu_ids() == [0]
dist.destroy_process_group()
print("NCCL_WORLD1_OK")
