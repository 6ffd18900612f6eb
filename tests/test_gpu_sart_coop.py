"""The cooperative SART chain (k_sart_tile COOP: the residual rows are formed inside the next tile step by its first
workgroups and handed over through flags) against the plain chain (tile step + k_resid_finish per angle): the rows come
from the same partial sums by the same additions, so the swept volumes must be bit-identical -- also when every workgroup
takes the do-it-yourself path, on ragged sizes, with a random angle order, over several sweeps and on two streams."""
import numpy as np
import pytest

from tomo_tv_amd._lib import VOL_ORIGINAL, VOL_RECON
from tomo_tv_amd.engine import tomoengine
from tomo_tv_amd.phantom import ellipsoids, tilt_angles

pytestmark = pytest.mark.gpu


def _sweep(ns, n, nproj, opts, order="sequential", niter=2, tracked=False):
    ang = np.deg2rad(tilt_angles(nproj))
    t = tomoengine(ns, n, ang)
    t.set_option("sart_resident", 0)      # these tests hold the STREAMED sweep's forms to each other (the resident sweep: test_gpu_sart_resident.py)
    for k, v in opts.items():
        t.set_option(k, v)
    vol = ellipsoids(ns, n)
    t.set_volume(vol, VOL_ORIGINAL)
    t.create_projections()
    t.initialize_SART(order)
    t._order_rng = np.random.default_rng(5)
    out = []
    if tracked:
        t.copy_recon()
        out.append(t.SART_tracked(0.7, niter))
    else:
        t.SART(0.7, niter)
    x = t.get_volume(VOL_RECON)
    return x, out


@pytest.mark.parametrize("ns,n,nproj", [(64, 64, 9), (70, 40, 7), (130, 96, 12), (3, 33, 5), (256, 128, 6)])
def test_coop_chain_is_bit_identical(gpu, ns, n, nproj):
    ref, _ = _sweep(ns, n, nproj, {"sart_coop": 0})
    got, _ = _sweep(ns, n, nproj, {"sart_coop": 1})
    assert np.array_equal(ref, got)
    own, _ = _sweep(ns, n, nproj, {"sart_coop": 1, "sart_coop_spin": -1})      # every workgroup forms its own rows
    assert np.array_equal(ref, own)
    assert np.isfinite(ref).all() and ref.max() > 0


def test_coop_chain_random_order_tracked_and_two_streams(gpu):
    ns, n, nproj = 256, 64, 11
    ref, nr = _sweep(ns, n, nproj, {"sart_coop": 0}, order="random", niter=3, tracked=True)
    got, ng = _sweep(ns, n, nproj, {"sart_coop": 1}, order="random", niter=3, tracked=True)
    assert np.array_equal(ref, got) and abs(nr[0] - ng[0]) <= 1e-12 * abs(nr[0])
    two, n2 = _sweep(ns, n, nproj, {"sart_coop": 1, "sart_streams": 2}, order="random", niter=3, tracked=True)
    assert np.array_equal(ref, two)
    assert abs(n2[0] - nr[0]) <= 1e-6 * abs(nr[0])       # the step norm adds sub-slab sums in another order


@pytest.mark.parametrize("ns,n,nproj", [(64, 64, 9), (70, 40, 7), (200, 96, 12)])
def test_skipping_unchanged_stores_is_bit_identical(gpu, ns, n, nproj):
    """k_sart_tile leaves out the store of a 256-byte piece whose bits did not change (in place): same volume, same norms."""
    ref, nr = _sweep(ns, n, nproj, {"sart_skip_same": 0}, niter=3, tracked=True)
    got, ng = _sweep(ns, n, nproj, {"sart_skip_same": 1}, niter=3, tracked=True)
    assert ref.tobytes() == got.tobytes()
    assert abs(nr[0] - ng[0]) <= 1e-12 * abs(nr[0])      # the step norm is a sum of per-workgroup doubles in arrival order
    assert (ref == 0).mean() > 0.05           # the phantom has a background the clamp holds at zero: pieces are skipped


@pytest.mark.parametrize("ns,n,nproj", [(64, 64, 9), (200, 48, 7)])
def test_cache_policy_and_chain_count_do_not_change_a_bit(gpu, ns, n, nproj):
    """The streamed form of the tile accesses ("sart_nt": non-temporal loads, write-through stores) and the number of
    sub-slab chains ("sart_streams") only change how the same values travel."""
    ref, _ = _sweep(ns, n, nproj, {"sart_nt": 0, "sart_streams": 1}, niter=2)
    for opts in ({"sart_nt": 1, "sart_streams": 1}, {"sart_nt": 1, "sart_streams": 2}, {"sart_nt": 0, "sart_streams": 2}, {}):
        got, _ = _sweep(ns, n, nproj, opts, niter=2)
        assert ref.tobytes() == got.tobytes(), opts


def test_scalar_snapshot_needs_a_snapshot(gpu):
    """tomo_scalars_snapshot_read without a snapshot in flight is an error, not stale values."""
    from tomo_tv_amd._lib import TomoError
    t = tomoengine(8, 16, np.deg2rad(tilt_angles(3)))
    with pytest.raises(TomoError):
        t.be.scalars_snapshot_read()
    t.be.scalars_snapshot()
    assert t.be.scalars_snapshot_read().shape[0] >= 8


def test_coop_needs_two_angles(gpu):
    """One tilt: consecutive links would read and write the same residual rows; the engine keeps the plain chain."""
    ref, _ = _sweep(64, 32, 1, {"sart_coop": 0}, niter=3)
    got, _ = _sweep(64, 32, 1, {"sart_coop": 1}, niter=3)
    assert np.array_equal(ref, got)
