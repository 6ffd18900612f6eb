"""Size-independent properties at the BASELINE.json shapes (the oracle is far too slow there).

config 2: 256^3, 60 tilts (SART)      config 3: 512^3, 90 tilts (FISTA / the SART+TV headline)
config 4: one GPU's shard of the 8-way split, 128 x 1024^2, 120 tilts
Properties: adjointness <Ax, y> = <x, A^T y>; linearity of the projector; fused SART == FP+BP SART; a volume
reconstructed as two half-slabs by two engines == one engine (what tilt-axis sharding relies on); the volume the
engine holds is what was uploaded; non-negativity and monotone data distance of SART.
"""
import numpy as np
import pytest

from conftest import rel_l2
from tomo_tv_amd._lib import SINO_B, SINO_G, VOL_ORIGINAL, VOL_RECON, VOL_TEMP
from tomo_tv_amd.engine import tomoengine
from tomo_tv_amd.phantom import ellipsoids, tilt_angles

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[(256, 256, 60), (512, 512, 90), (128, 1024, 120)],
                ids=["config2_256x60", "config3_512x90", "config4_shard_128x1024x120"])
def big(request, gpu):
    nx, n, p = request.param
    ang = np.deg2rad(tilt_angles(p))
    x = ellipsoids(nx, n)
    t = tomoengine(nx, n, ang)
    t.set_volume(x, VOL_ORIGINAL)
    t.create_projections()
    return t, x, (nx, n, p), ang


def test_roundtrip_and_adjointness(big):
    t, x, (nx, n, p), ang = big
    assert np.array_equal(t.get_volume(VOL_ORIGINAL), x)
    b = t.get_projections()
    assert b.shape == (nx, n * p) and np.isfinite(b).all() and b.min() >= 0
    rng = np.random.default_rng(0)
    y = rng.random(b.shape, dtype=np.float32)
    t.be.c("set_sinogram", SINO_G, y.ctypes.data)
    t.be.c("back_projection", SINO_G, VOL_TEMP)
    aty = t.get_volume(VOL_TEMP)
    lhs = float(np.vdot(b.astype(np.float64), y.astype(np.float64)))
    rhs = float(np.vdot(x.astype(np.float64), aty.astype(np.float64)))
    assert abs(lhs - rhs) <= 2e-6 * abs(lhs), (lhs, rhs)


def test_projector_linearity(big):
    t, x, (nx, n, p), ang = big
    b = t.get_projections()
    z = np.roll(x, 7, axis=1) * np.float32(0.5)
    t.set_volume(z, VOL_RECON)
    t.forward_projection()
    bz = t.get_model_projections()
    t.set_volume((2 * x + z).astype(np.float32), VOL_RECON)
    t.forward_projection()
    assert rel_l2(t.get_model_projections(), 2 * b.astype(np.float64) + bz) < 1e-6


def test_sart_fused_equals_unfused_and_converges(big):
    t, x, (nx, n, p), ang = big
    vols = []
    for fused in (1, 0):
        t.set_option("sart_fused", fused)
        t.restart_recon()
        t.SART(0.5, 1)
        vols.append(t.get_volume())
    t.set_option("sart_fused", 1)
    assert rel_l2(vols[0], vols[1]) < 2e-6
    assert vols[0].min() >= 0
    dd0 = t.data_distance()
    t.SART(0.5, 1)
    dd1 = t.data_distance()
    assert dd1 < dd0 < np.linalg.norm(t.get_projections())


def test_two_half_slabs_equal_one_slab(big):
    """Every slice shares one system matrix, so slab sharding must not change a single voxel."""
    t, x, (nx, n, p), ang = big
    b = t.get_projections()
    t.restart_recon()
    t.SART(0.7, 1)
    whole = t.get_volume()
    half = nx // 2
    parts = []
    for s0 in (0, half):
        e = tomoengine(half, n, ang)
        e.set_tilt_series(b[s0:s0 + half])
        e.SART(0.7, 1)
        parts.append(e.get_volume())
        del e
    assert rel_l2(np.concatenate(parts), whole) < 2e-6


def test_tile_projectors_equal_ray_and_pixel_driven_forms(big):
    """All-angle FP/BP from LDS tiles (k_fp_tile, k_bp_tile) against the ray-driven / pixel-driven kernels at full size:
    the normalised SIRT iterate agrees to summation-order round-off; the back-projection alone is bit-identical."""
    t, x, (nx, n, p), ang = big
    vols = {}
    for tile in (1, 0):
        t.set_option("fp_tile", tile); t.set_option("bp_tile", tile)
        t.restart_recon()
        t.SIRT(2)
        vols[tile] = (t.get_volume(), t.data_distance())
    assert rel_l2(vols[1][0], vols[0][0]) < 2e-6 and abs(vols[1][1] - vols[0][1]) <= 1e-5 * vols[0][1]
    out = {}
    for tile in (1, 0):
        t.set_option("fp_tile", 0); t.set_option("bp_tile", tile)
        t.restart_recon()
        t.SIRT(1)
        out[tile] = t.get_volume()
    t.set_option("fp_tile", 1); t.set_option("bp_tile", 1)
    assert np.array_equal(out[0], out[1])


def test_strip_projector_equals_tile_form_at_full_size(big):
    """The sheared-strip forward projector (k_fp_strip: the form large slabs run by default, round 4) against the tile form:
    the same matrix entries summed in a different order."""
    t, x, (nx, n, p), ang = big
    if not t.get_option("fp_strip_ready"):
        pytest.skip("the engine built no strip tables for this slab (slab-size rule)")
    out = {}
    forms = (["list"] if t.get_option("fp_list_ready") else []) + ["strip", "tile"]   # (k_fp_list: the strips as wave-uniform entry lists)
    for form in forms:
        if form == "tile":
            t.set_option("fp_tile", 1)
        else:
            t.set_option("fp_strip", 1)
            t.set_option("fp_list", 1 if form == "list" else 0)
        t.restart_recon()
        t.create_projections()
        b = t.get_projections()
        t.SIRT(2)
        out[form] = (b, t.get_volume(), t.data_distance())
    t.set_option("fp_strip", 1)
    t.set_option("fp_list", 1)
    t.create_projections()                      # the fixture's tilt series as the default form makes it
    t.restart_recon()
    for form in forms[:-1]:
        assert rel_l2(out[form][0], out["tile"][0]) < 1e-6, form
        assert rel_l2(out[form][1], out["tile"][1]) < 2e-6 and abs(out[form][2] - out["tile"][2]) <= 1e-5 * out["tile"][2], form


@pytest.fixture(scope="module", params=[(512, 512), (128, 1024)], ids=["config3_512cube", "config4_shard_128x1024sq"])
def tvbig(request, gpu):
    """TV / FGP never touch the projector tables: two tilts keep the engine small at the BASELINE volume sizes."""
    nx, n = request.param
    rng = np.random.default_rng(17)
    x = ellipsoids(nx, n) + np.float32(0.05) * rng.random((nx, n, n), dtype=np.float32)
    t = tomoengine(nx, n, np.deg2rad(np.array([-30.0, 40.0])))
    return t, np.ascontiguousarray(x, dtype=np.float32)


def test_tv_gd_three_kernel_forms_agree_at_full_size(tvbig):
    """tv_gd(3): the register march (default) == the LDS march bit for bit (same rounding sequence by construction) and
    == the direct-global stencil (IEEE sqrt + division per term) to 1e-6; the TV value agrees to 1e-6."""
    t, x = tvbig
    t.tv_eps = 1e-6
    res = {}
    for form in (1, 8, 0):
        t.set_option("tv_lds", form)
        t.set_volume(x, VOL_RECON)
        tv0 = t.tv_gd(3, 1.0)
        res[form] = (tv0, t.get_volume())
    t.set_option("tv_lds", 1)
    assert np.array_equal(res[1][1], res[8][1]) and abs(res[1][0] - res[8][0]) <= 1e-12 * res[8][0]   # (fp64 block sums in arrival order)
    assert rel_l2(res[1][1], res[0][1]) < 1e-6 and abs(res[1][0] - res[0][0]) <= 1e-6 * res[0][0]
    assert res[1][1].min() >= 0 and 0 < rel_l2(res[1][1], x) < 1e-2       # a real step was taken; positivity held
    assert t.tv() < res[1][0]                                               # and it lowered the TV


def test_fgp_fused_equals_two_kernel_form_at_full_size(tvbig):
    """tv_fgp(4, 0.1): one fused kernel per iteration == the Obj / Grad pair of tv_fgp.cu:244-268 (same arithmetic per
    voxel, D never written in the fused form); the prox lowers the TV and stays non-negative."""
    t, x = tvbig
    res = {}
    for fused in (1, 0):
        t.set_option("fgp_fused", fused)
        t.set_volume(x, VOL_RECON)
        tv_in = t.tv_fgp(4, 0.1)
        res[fused] = (tv_in, t.get_volume())
    t.set_option("fgp_fused", 1)
    assert res[1][0] == res[0][0]
    assert np.array_equal(res[1][1], res[0][1]) or rel_l2(res[1][1], res[0][1]) < 1e-7
    # two iterations per pass (k_fgp_fused2, what res[1] ran with) == one per pass, bit for bit, at an even and an odd count
    for iters in (4, 7):
        pair = {}
        for p in (1, 0):
            t.set_option("fgp_pair", p)
            t.set_volume(x, VOL_RECON)
            t.tv_fgp(iters, 0.1)
            pair[p] = t.get_volume()
        t.set_option("fgp_pair", 1)
        assert np.array_equal(pair[1], pair[0]), iters
    t.set_volume(res[1][1], VOL_RECON)
    t.tv_eps = 1e-6
    assert res[1][1].min() >= 0 and t.tv() < res[1][0]


def test_sart_two_stream_sub_slabs_equal_one_chain(big):
    """"sart_streams" = 2 (two sub-slabs of the slab on two streams, second chain enqueued by a second host thread) changes
    only the launch structure: every voxel is bit-identical to the single chain, also where a sub-slab runs its per-row
    kernels at another vector width than the whole slab (128 slices: 64 + 64 at one float per lane instead of two) --
    k_bp_angle writes every rounding out for that."""
    t, x, (nx, n, p), ang = big
    res = {}
    t.set_option("sart_resident", 0)              # the launch structure of the STREAMED sweep is what this test is about
    for ns in (1, 2):
        t.set_option("sart_streams", ns)
        t.restart_recon()
        t.copy_recon()
        dp = t.SART_tracked(0.6, 1)
        t.SART(0.6, 1)
        res[ns] = (dp, t.get_volume())
    t.set_option("sart_streams", 1)
    t.set_option("sart_resident", -1)
    # fp64 partial sums: atomics in any order, and a sub-slab may run its per-row kernels at another vector width
    assert res[1][0] == res[2][0] or abs(res[1][0] - res[2][0]) <= 1e-10 * res[1][0]
    assert np.array_equal(res[1][1], res[2][1])


def test_config4_whole_volume_on_one_gpu(gpu, monkeypatch):
    """BASELINE config 4 unsharded: 1024^3, 120 tilts on ONE GPU (4 GiB per volume, every byte offset beyond 2^31).
    The phantom is the 128-slice shard phantom tiled 8 times along the tilt axis, so every 128-slice block of the whole-
    volume result must equal what a 128 x 1024^2 engine makes of one block (slices share the system matrix): SART sweep
    voxel for voxel, data distance, and the TV descent of the periodic 8-fold volume."""
    import psutil
    if psutil.virtual_memory().available < 40 * 2 ** 30:
        pytest.skip("needs ~25 GiB of host memory for the 1024^3 arrays")
    nx, n, p, blk = 1024, 1024, 120, 128
    ang = np.deg2rad(tilt_angles(p))
    xb = ellipsoids(blk, n)
    # the whole volume projects by sheared strips (slab-size rule, round 4): give the 128-slice engine the same form, so that the
    # two sum every ray in the same order and the comparison below can stay bit for bit
    monkeypatch.setenv("TOMO_FP_STRIP", "1")
    small = tomoengine(blk, n, ang)
    small.set_volume(xb, VOL_ORIGINAL)
    small.create_projections()
    bb = small.get_projections()
    small.SART(0.5, 1)
    want = small.get_volume()
    dd_small = small.data_distance()
    del small
    t = tomoengine(nx, n, ang)
    t.set_volume(np.tile(xb, (nx // blk, 1, 1)), VOL_ORIGINAL)
    t.create_projections()
    b = t.get_projections()
    assert all(np.array_equal(b[k * blk:(k + 1) * blk], bb) for k in (0, 3, 7))
    del b
    t.SART(0.5, 1)
    got = t.get_volume()
    for k in (0, 4, 7):
        assert rel_l2(got[k * blk:(k + 1) * blk], want) < 2e-6, k
    dd = t.data_distance()
    assert abs(dd - dd_small * np.sqrt(nx / blk)) <= 1e-5 * dd
    # TV descent: the 8-fold periodic volume behaves like one period with periodic wrap (what a single slab does)
    t.tv_eps = 1e-6
    tv0 = t.tv_gd(2, 1.0)
    small = tomoengine(blk, n, ang[:2])
    small.set_volume(want, VOL_RECON)
    small.tv_eps = 1e-6
    tv0_small = small.tv_gd(2, 1.0 / np.sqrt(nx / blk))     # the step length is normalised by the global gradient norm
    assert abs(tv0 - tv0_small * (nx / blk)) <= 2e-6 * tv0
    got = t.get_volume()
    assert rel_l2(got[5 * blk:6 * blk], small.get_volume()) < 5e-6
