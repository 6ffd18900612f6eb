"""The in-process multi-GPU facade (tomo_tv_amd/inprocess.py) on CPU: K slab engines behind ONE method table, one host thread
per engine, composed like the ranks of a job ("one call -> enqueue on all GPUs -> one sync"; reference:
tomofusion/__init__.py:21-34, gpu/reconstructor.py:23-33, multigpuengine.cpp:140-193).

The per-slab kernels are replaced by tests/slab_double.py (numpy + oracle), so what is under test is the product's host logic:
every facade call is forwarded to all slab engines at once, scalars come back all-reduced, getters assemble the whole array, plain
attributes are written to every rank, an exception on one rank surfaces on the caller and leaves the world usable.  The sharded
result must equal the single-process oracle.
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import oracle  # noqa: E402
from slab_double import OracleSlabBackend  # noqa: E402
from tomo_tv_amd import engine  # noqa: E402
from tomo_tv_amd._lib import VOL_TEMP  # noqa: E402
from tomo_tv_amd.inprocess import InProcessMultiGPU  # noqa: E402
from tomo_tv_amd.phantom import ellipsoids  # noqa: E402


class ShardedEngine(engine.tomoengine):
    _backend_cls = OracleSlabBackend


def rel(a, b):
    return float(np.linalg.norm((a.astype(np.float64) - b).ravel()) / np.linalg.norm(b.astype(np.float64).ravel()))


@pytest.mark.parametrize("world,Nx", [(2, 6), (3, 7), (1, 4), (4, 9), (8, 16)])
def test_facade_over_threads_equals_single_process_oracle(world, Nx):
    N, P = 16, 5
    ang = np.linspace(-65, 65, P)
    x = ellipsoids(Nx, N, seed=21)
    full = oracle.ctvlib(Nx, N, P)
    full.load_A(oracle.parallel_ray(N, ang))
    full.original_volume = x.copy()
    full.create_projections()
    t = InProcessMultiGPU(lambda comm, dev: ShardedEngine(Nx, N, ang * np.pi / 180, device=dev, comm=comm), [0] * world)
    try:
        assert t.get_gpu_ids() == [0] * world and t.is_multi_gpu_enabled() == (world > 1)
        t.tv_eps = full.tv_eps = 1e-6                     # written to every rank
        assert all(e.tv_eps == 1e-6 for e in t._engines)
        t.set_tilt_series(full.b)                         # the GLOBAL array; every rank keeps its slab
        t.copy_recon(); full.copy_recon()
        t.SART(0.5, 1); full.SART(0.5, 1)
        assert abs(t.matrix_2norm() - full.matrix_2norm()) <= 1e-5 * full.matrix_2norm()
        assert abs(t.data_distance() - full.data_distance(normalize=False)) <= 1e-5 * full.data_distance(normalize=False)
        assert abs(t.tv() - full.tv()) <= 1e-5 * full.tv()
        t.copy_recon(); full.copy_recon()
        tv0, tv0_ref = t.tv_gd(4, 0.3), full.tv_gd(4, 0.3)
        assert abs(tv0 - tv0_ref) <= 1e-5 * tv0_ref
        got = t.get_volume()
        assert got.shape == (Nx, N, N) and rel(got, full.recon) < 1e-5
        assert rel(t.get_recon(Nx // 2), full.recon[Nx // 2]) < 1e-5     # collective: broadcast from the owner
        t.copy_recon(); full.copy_recon()
        dp = t.SART_tracked(0.5, 1)
        full.SART(0.5, 1)
        assert abs(dp - full.matrix_2norm()) <= 1e-5 * full.matrix_2norm()
        assert rel(t.get_volume(VOL_TEMP), full.recon) < 1e-5
        # an exception on the ranks surfaces on the caller, and the world stays usable afterwards
        with pytest.raises(ValueError):
            t.set_tilt_series(np.zeros((Nx + 1, N * P), np.float32))
        assert abs(t.tv() - full.tv()) <= 1e-5 * full.tv()
    finally:
        t.close()


def test_plain_process_config_follows_the_visible_devices(monkeypatch):
    """determine_gpu_config (tomofusion/__init__.py:21-34): 'multigpu' when no device is named and more than one is visible."""
    from tomo_tv_amd import reconstructor
    monkeypatch.setattr(reconstructor, "device_count", lambda: [0, 1, 2, 3])
    assert reconstructor.determine_gpu_config(-1) == "multigpu"
    assert reconstructor.determine_gpu_config(2) == "singleconfig"
    monkeypatch.setenv("TOMO_SINGLE_GPU", "1")
    assert reconstructor.determine_gpu_config(-1) == "singleconfig"
    monkeypatch.delenv("TOMO_SINGLE_GPU")
    monkeypatch.setattr(reconstructor, "device_count", lambda: [0])
    assert reconstructor.determine_gpu_config(-1) == "singleconfig"
    monkeypatch.setattr(reconstructor, "device_count", lambda: [])
    with pytest.raises(ValueError):
        reconstructor.determine_gpu_config(-1)


def test_plain_process_never_builds_more_slabs_than_slices(monkeypatch):
    """``multigpuengine`` / ``multigpufusion`` in a plain process took every visible device, so a volume with fewer slices
    than GPUs raised 'more ranks than slices' where the single-GPU path (and the reference's per-slice scheduler,
    multigpuengine.cpp:163-193) works.  The device list is cut to the slice count; one device left = the plain single-GPU class."""
    from tomo_tv_amd import chemistry, inprocess
    seen = {}

    class FakeFacade:
        def __init__(self, make, devs):
            seen["devs"] = list(devs)

    class FakeSingle:
        def __init__(self, *a, **k):
            seen["single"] = k.get("device")

    monkeypatch.setattr(inprocess, "visible_devices", lambda: list(range(8)))
    monkeypatch.setattr(inprocess, "process_group_world", lambda: 1)
    monkeypatch.setattr(inprocess, "InProcessMultiGPU", FakeFacade)
    ang = np.deg2rad(np.linspace(-60, 60, 5))
    assert isinstance(engine.multigpuengine(3, 16, ang), FakeFacade) and seen["devs"] == [0, 1, 2]
    assert isinstance(engine.multigpuengine(100, 16, ang), FakeFacade) and seen["devs"] == list(range(8))
    assert isinstance(engine.multigpuengine(5, 16, ang, devices=[4, 5, 6, 7, 1, 2]), FakeFacade) and seen["devs"] == [4, 5, 6, 7, 1]
    monkeypatch.setattr(engine, "tomoengine", FakeSingle)
    assert isinstance(engine.multigpuengine(1, 16, ang), FakeSingle) and seen["single"] == 0
    assert isinstance(chemistry.multigpufusion(2, 16, 2, ang, ang), FakeFacade) and seen["devs"] == [0, 1]
    monkeypatch.setattr(chemistry, "multimodal", FakeSingle)
    assert isinstance(chemistry.multigpufusion(1, 16, 2, ang, ang, devices=[3, 4]), FakeSingle) and seen["single"] == 3


def test_driver_loops_run_on_the_rank_threads(monkeypatch):
    """``TomoGPU.asd_pocs`` / ``.sart`` over the in-process facade: the loop runs ON the rank threads (one crossing of the facade per
    driver call instead of ~7 per iteration: reconstructor.py ``_on_rank_threads``) and gives what the same driver gives over one
    engine that holds everything (gpu/reconstructor.py:75-192 drives ``self.tomo`` the same way whatever class it is)."""
    from tomo_tv_amd import reconstructor
    from tomo_tv_amd.inprocess import InProcWorld
    N, P, Nx = 16, 5, 7
    ang = np.linspace(-65, 65, P)
    x = ellipsoids(Nx, N, seed=23)
    full = oracle.ctvlib(Nx, N, P)
    full.load_A(oracle.parallel_ray(N, ang))
    full.original_volume = x.copy()
    full.create_projections()
    series = full.b.reshape(Nx, P, N).transpose(0, 2, 1)            # (Nslice, Nray, Nangles), what TomoGPU takes

    def driver(tomo):
        rec = reconstructor.TomoGPU.__new__(reconstructor.TomoGPU)   # (the constructor insists on a HIP device)
        rec.tomo, rec.verbose, rec.recon, rec.cost = tomo, False, None, None
        rec.set_tilt_series(series)
        return rec

    t1 = InProcessMultiGPU(lambda comm, dev: ShardedEngine(Nx, N, ang * np.pi / 180, device=dev, comm=comm), [0])   # (one slab: == oracle, test above)
    try:
        one = driver(t1)
        dd1, tv1 = one.asd_pocs(Niter=3)
        v1 = t1.get_volume()
        c1 = one.sart(Niter=2).copy()
    finally:
        t1.close()
    crossings = []
    real_run = InProcWorld.run
    monkeypatch.setattr(InProcWorld, "run", lambda self, fn: (crossings.append(1), real_run(self, fn))[1])
    t = InProcessMultiGPU(lambda comm, dev: ShardedEngine(Nx, N, ang * np.pi / 180, device=dev, comm=comm), [0, 0, 0])
    try:
        many = driver(t)
        n0 = len(crossings)
        dd3, tv3 = many.asd_pocs(Niter=3)
        used = len(crossings) - n0
        assert used <= 4, used                                        # initialize_algorithm's two calls + ONE for the whole loop
        assert np.allclose(dd3, dd1, rtol=1e-5) and np.allclose(tv3, tv1, rtol=1e-5)
        assert rel(t.get_volume(), v1) < 1e-5
        c3 = many.sart(Niter=2)
        assert np.allclose(c3, c1, rtol=1e-5) and many.cost is not None
    finally:
        t.close()


def test_stale_results_of_an_interrupted_call_are_dropped():
    """a facade call interrupted on the calling thread leaves its workers running; their late (rank, value) entries must
    not be consumed by the NEXT call as its own results.  Every job carries a sequence number (InProcWorld.run)."""
    import time
    from tomo_tv_amd.inprocess import InProcWorld
    w = InProcWorld(3)
    try:
        # what an interrupted run() leaves behind: jobs handed out, nobody collecting
        w._seq = seq = getattr(w, "_seq", 0) + 1
        for q in w._jobs:
            q.put((seq, lambda r: (time.sleep(0.05), "stale")[1]))
        assert w.run(lambda r: ("fresh", r)) == [("fresh", 0), ("fresh", 1), ("fresh", 2)]
        assert w.run(lambda r: r * 2) == [0, 2, 4]
    finally:
        w.close()


def test_on_rank_threads_copies_back_whatever_the_driver_set():
    """only cost / dd_vec / tv_vec came back from rank 0's clone; now every attribute the loop set or replaced does."""
    from tomo_tv_amd import reconstructor

    class FakeFacade(InProcessMultiGPU):
        def __init__(self, n):
            from tomo_tv_amd.inprocess import InProcWorld
            object.__setattr__(self, "_engines", [object() for _ in range(n)])
            object.__setattr__(self, "_world", InProcWorld(n))

    class Driver:
        def __init__(self, tomo):
            self.tomo, self.kept, self.replaced = tomo, [1, 2], "old"

        @reconstructor._on_rank_threads
        def loop(self, k):
            assert self._on_rank and not isinstance(self.tomo, InProcessMultiGPU)
            self.replaced = "new"
            self.fresh = np.arange(k)
            return k + 1

    f = FakeFacade(2)
    try:
        d = Driver(f)
        kept = d.kept
        assert d.loop(3) == 4
        assert d.replaced == "new" and np.array_equal(d.fresh, np.arange(3)) and d.kept is kept
        assert d.tomo is f and not getattr(d, "_on_rank", False)
    finally:
        f._world.close()
