"""Parity of the HIP path (through the C ABI / ctypes facade) against the oracle and the golden fixtures.

Tolerance: relative L2 <= 1e-5 on volumes and sinograms for single operator applications and short runs (the
north-star figure); scalars to 1e-5 relative.  Longer runs state their own bound where fp32 summation-order
differences accumulate.
"""
import numpy as np
import pytest

import oracle
from conftest import rel_l2
from tomo_tv_amd import pytvlib
from tomo_tv_amd._lib import TomoError, VOL_ORIGINAL, VOL_RECON, VOL_TEMP, VOL_YK
from tomo_tv_amd.engine import ctvlib, tomoengine
from tomo_tv_amd.phantom import ellipsoids

pytestmark = pytest.mark.gpu
TOL = 1e-5

SHAPES = [(16, 5, 2), (32, 9, 4), (64, 16, 8)]


def make_pair(golden, N, P, Nx):
    """(gpu ctvlib facade, oracle ctvlib, golden trace) on the imported-reference matrix."""
    A = golden(f"A_N{N}_P{P}.npz")["A"]
    g = golden(f"trace_N{N}_P{P}_Nx{Nx}.npz")
    dev = ctvlib(Nx, N, P)
    dev.load_A(A)
    ref = oracle.ctvlib(Nx, N, P)
    ref.load_A(A)
    return dev, ref, g


def seed_volume(dev, ref, x):
    dev.set_volume(x, VOL_RECON)
    ref.recon[:] = x


@pytest.mark.parametrize("N,P,Nx", SHAPES)
def test_forward_projection_and_lipschitz(gpu, golden, N, P, Nx):
    dev, ref, g = make_pair(golden, N, P, Nx)
    pytvlib.create_projections(dev, g["x0"])
    b = dev.get_projections()
    assert b.shape == (Nx, N * P)
    assert rel_l2(b, g["b"]) < TOL
    assert abs(dev.lipschits() - float(g["lipschitz"])) <= 1e-6 * float(g["lipschitz"])
    # A^T b
    dev.back_projection_of_tilt_series()
    assert rel_l2(dev.get_volume(), g["ATb"]) < TOL


@pytest.mark.parametrize("N,P,Nx", SHAPES)
def test_sirt_landweber_trace(gpu, golden, N, P, Nx):
    """cpu/sim_tomo.py:35-61 harness with alg='SIRT': iterates at k=1,5,50 and the dd / rmse traces."""
    dev, ref, g = make_pair(golden, N, P, Nx)
    pytvlib.create_projections(dev, g["x0"])
    beta = 1.0 / dev.lipschits()
    dd, rm = [], []
    for it in range(50):
        pytvlib.run_ctvlib(dev, "SIRT", beta)
        dd.append(dev.data_distance())
        rm.append(dev.rmse())
        if it + 1 in (1, 5, 50):
            assert rel_l2(dev.get_volume(), g[f"sirt_k{it + 1}"]) < TOL, f"iterate {it + 1}"
    assert np.allclose(dd, g["sirt_dd"], rtol=1e-5)
    assert np.allclose(rm, g["sirt_rmse"], rtol=1e-5)


@pytest.mark.parametrize("N,P,Nx", SHAPES)
def test_sart_and_normalised_sirt(gpu, golden, N, P, Nx):
    A = golden(f"A_N{N}_P{P}.npz")
    g = golden(f"trace_N{N}_P{P}_Nx{Nx}.npz")
    dev = tomoengine(Nx, N, np.asarray(A["angles_deg"]) * np.pi / 180)
    dev.set_tilt_series(g["b"])
    dev.initialize_SART("sequential")
    dev.SART(0.25, 1)
    assert rel_l2(dev.get_volume(), g["sart_b025"]) < TOL
    dev.SART(1.0, 1)
    assert rel_l2(dev.get_volume(), g["sart_b1"]) < TOL
    dev.restart_recon()
    dev.initialize_SIRT()
    dev.SIRT(5)
    assert rel_l2(dev.get_volume(), g["sirtnorm_5"]) < TOL
    # un-normalised GPU-style data distance (tomoengine.cpp:410-413) vs the oracle
    ref = oracle.ctvlib(Nx, N, P)
    ref.load_A(A["A"])
    ref.set_tilt_series(g["b"])
    ref.recon[:] = g["sirtnorm_5"]
    assert abs(dev.data_distance() - ref.data_distance(normalize=False)) <= 2e-5 * ref.data_distance(normalize=False)
    assert rel_l2(dev.get_model_projections(), ref.g) < TOL


@pytest.mark.parametrize("Nx,P,niter,order", [(8, 16, 1, "sequential"), (8, 16, 3, "sequential"), (130, 5, 2, "random"),
                                               (256, 1, 2, "sequential")])
def test_fused_sart_equals_separate_fp_bp(gpu, Nx, P, niter, order):
    """The fused [BP(a_k)+FP(a_k+1)] chain performs the same per-voxel arithmetic as one FP + one BP per angle."""
    N = 64 if Nx <= 8 else 24
    ang = np.deg2rad(np.linspace(-70, 70, P)) if P > 1 else np.deg2rad([17.0])
    x = ellipsoids(Nx, N, seed=3)
    vols = []
    for fused in (1, 0):
        dev = tomoengine(Nx, N, ang)
        dev.set_option("sart_fused", fused)
        dev.set_volume(x, VOL_ORIGINAL)
        dev.create_projections()
        dev.initialize_SART(order)
        dev.SART(0.7, niter)
        dev.SART(0.3, 1)                      # a second call starts from the swapped buffer
        vols.append(dev.get_volume())
        dd = dev.data_distance()
    assert rel_l2(vols[0], vols[1]) < 2e-6
    ref = oracle.ctvlib(Nx, N, P)
    ref.load_A(oracle.parallel_ray(N, np.rad2deg(ang)))
    ref.original_volume = x.copy()
    ref.create_projections()
    if order == "sequential":
        ref.SART(0.7, niter)
        ref.SART(0.3, 1)
        assert rel_l2(vols[0], ref.recon) < TOL
        assert abs(dd - ref.data_distance(normalize=False)) <= 2e-5 * dd


def test_sart_random_order_is_a_permutation_sweep(gpu, golden):
    N, P, Nx = 32, 9, 4
    A = golden(f"A_N{N}_P{P}.npz")
    g = golden(f"trace_N{N}_P{P}_Nx{Nx}.npz")
    dev = tomoengine(Nx, N, np.asarray(A["angles_deg"]) * np.pi / 180)
    dev.set_tilt_series(g["b"])
    dev.initialize_SART("random")
    order = np.random.default_rng(0).permutation(P)     # the engine's seeded stream
    dev.SART(0.5, 1)
    ref = oracle.ctvlib(Nx, N, P)
    ref.load_A(A["A"])
    ref.set_tilt_series(g["b"])
    ref.SART(0.5, 1, order=order)
    assert rel_l2(dev.get_volume(), ref.recon) < TOL
    with pytest.raises(ValueError):
        dev.initialize_SART("zigzag")


@pytest.mark.parametrize("N,P,Nx", SHAPES[:2])
def test_art_sweep(gpu, golden, N, P, Nx):
    dev, ref, g = make_pair(golden, N, P, Nx)
    dev.set_tilt_series(g["b"])
    pytvlib.initialize_ctvlib  # noqa: B018 (documented entry point)
    dev.row_inner_product()
    dev.ART(0.5)
    assert rel_l2(dev.get_volume(), g["art_1"]) < TOL


def test_randart_permutation_sweep(gpu, golden):
    N, P, Nx = 32, 9, 4
    dev, ref, g = make_pair(golden, N, P, Nx)
    dev.set_tilt_series(g["b"])
    ref.set_tilt_series(g["b"])
    ref.row_inner_product()
    for seed in (5, 6):
        order = dev.randART(0.5, seed=seed)
        assert np.array_equal(np.sort(order), np.arange(N * P))
        ref.ART(0.5, order=order)
        assert rel_l2(dev.get_volume(), ref.recon) < TOL


@pytest.mark.parametrize("N,P,Nx", SHAPES)
@pytest.mark.parametrize("eps", [1e-8, 1e-6])
def test_tv_value_and_gradient_descent(gpu, golden, N, P, Nx, eps):
    dev, ref, g = make_pair(golden, N, P, Nx)
    dev.tv_eps = eps
    seed_volume(dev, ref, g["x_sart"])
    tv = dev.tv()
    assert abs(tv - float(g[f"tv_eps{eps:g}"])) <= 1e-5 * tv
    for ng in (1, 10):
        seed_volume(dev, ref, g["x_sart"])
        tv0 = dev.tv_gd(ng, 0.05)
        assert abs(tv0 - float(g[f"tvgd_tv0_ng{ng}_eps{eps:g}"])) <= 1e-5 * tv0
        assert rel_l2(dev.get_volume(), g[f"tvgd_ng{ng}_eps{eps:g}"]) < TOL
        assert dev.get_volume().min() >= 0


@pytest.mark.parametrize("N,P,Nx", SHAPES)
def test_tv_fgp(gpu, golden, N, P, Nx):
    dev, ref, g = make_pair(golden, N, P, Nx)
    for it, lam in [(1, 0.1), (10, 0.1), (10, 15.0)]:
        seed_volume(dev, ref, g["x_sart"])
        tv0 = dev.tv_fgp(it, lam)
        assert abs(tv0 - float(g[f"fgp_tv0_i{it}_l{lam:g}"])) <= 1e-5 * tv0
        assert rel_l2(dev.get_volume(), g[f"fgp_i{it}_l{lam:g}"]) < TOL


def test_copy_recon_matrix_2norm_rmse_l1(gpu, golden):
    N, P, Nx = 32, 9, 4
    dev, ref, g = make_pair(golden, N, P, Nx)
    seed_volume(dev, ref, g["x_sart"])
    dev.copy_recon()
    ref.copy_recon()
    assert dev.matrix_2norm() == 0.0
    seed_volume(dev, ref, g["sart_b025"])
    assert abs(dev.matrix_2norm() - ref.matrix_2norm()) <= 1e-6 * ref.matrix_2norm()
    dev.set_volume(g["x0"], VOL_ORIGINAL)
    ref.original_volume = g["x0"].copy()
    assert abs(dev.rmse() - ref.rmse()) <= 1e-6 * ref.rmse()
    assert abs(dev.l1_norm() - np.abs(g["sart_b025"]).sum(dtype=np.float64)) <= 1e-6 * dev.l1_norm()
    assert abs(dev.original_tv() - ref.original_tv()) <= 1e-5 * ref.original_tv()


def test_fista_momentum_and_soft_threshold(gpu, golden):
    N, P, Nx = 32, 9, 4
    dev, ref, g = make_pair(golden, N, P, Nx)
    seed_volume(dev, ref, g["x_sart"])
    dev.initialize_fista()
    ref.initialize_fista()
    dev.set_volume(g["sart_b025"], VOL_YK)
    ref.yk[:] = g["sart_b025"]
    dev.fista_momentum(0.37)
    ref.fista_momentum(0.37)
    assert np.array_equal(dev.get_volume(VOL_RECON), ref.recon)
    assert rel_l2(dev.get_volume(VOL_YK), ref.yk) < 1e-7
    dev.remove_momentum()
    x = g["x_sart"] - 0.2
    dev.set_volume(x)
    dev.soft_threshold(0.1)
    want = np.maximum(np.sign(x) * np.maximum(np.abs(x) - 0.1, 0), 0)
    assert rel_l2(dev.get_volume(), want) < 1e-7


def test_fista_momentum_rotation_keeps_the_three_volumes(gpu, golden):
    """tomoengine.cpp:381-384 leaves recon == recon_old == the prox result and yk = the extrapolated point.  The engine rotates
    buffers and keeps recon_old == recon as a flag (two reads, one store per step): every way of looking at or writing the three
    volumes between steps must see the reference's state."""
    from tomo_tv_amd._lib import VOL_RECON_OLD
    N, P, Nx = 32, 9, 4
    dev, ref, g = make_pair(golden, N, P, Nx)
    rng = np.random.default_rng(1)
    seed_volume(dev, ref, g["x_sart"])
    dev.initialize_fista()
    ref.initialize_fista()

    def same(tag):
        for vol, want in ((VOL_RECON, ref.recon), (VOL_YK, ref.yk), (VOL_RECON_OLD, ref.recon_old)):
            assert np.array_equal(dev.get_volume(vol), want), (tag, vol)
    for k, beta in enumerate((0.0, 0.28, 0.43, 0.53, 0.6)):
        y = (g["x_sart"] * np.float32(1 + 0.1 * k) + rng.random(g["x_sart"].shape, dtype=np.float32) * np.float32(0.01))
        dev.set_volume(y, VOL_YK)
        ref.yk[:] = y
        if k == 2:                                   # look at recon_old while it is only a flag, then carry on
            assert np.array_equal(dev.get_volume(VOL_RECON_OLD), ref.recon_old)
        if k == 3:                                   # a caller overwrites recon between two steps: recon_old must keep its content
            z = g["sart_b025"] * np.float32(0.5)
            dev.set_volume(z, VOL_RECON)
            ref.recon[:] = z
        if k == 4:                                   # ... and recon_old itself
            z = g["sart_b1"]
            dev.set_volume(z, VOL_RECON_OLD)
            ref.recon_old[:] = z
        dev.fista_momentum(beta)
        ref.fista_momentum(beta)
        assert np.array_equal(dev.get_volume(VOL_RECON), ref.recon), k
        assert rel_l2(dev.get_volume(VOL_YK), ref.yk) < 1e-7, k
        ref.yk[:] = dev.get_volume(VOL_YK)           # (FMA contraction of r + beta (r - old): carry the device's bits)
        same(k)
    dev.restart_recon()
    ref.restart_recon()
    same("restart")


def test_projection_reuse_is_bit_identical_and_never_stale(gpu, golden):
    """"fp_reuse": a SIRT / CGLS call whose volume is exactly what the model sinogram was last projected from starts from that
    sinogram instead of projecting again (the reference's drivers alternate step and data_distance: gpu/reconstructor.py:61-71).
    Same bits as always projecting -- and every way of changing the volume or the sinogram in between must drop the claim."""
    from tomo_tv_amd._lib import SINO_G, VOL_TEMP
    N, P, Nx = 32, 9, 4
    A = golden(f"A_N{N}_P{P}.npz")
    g = golden(f"trace_N{N}_P{P}_Nx{Nx}.npz")
    ang = np.asarray(A["angles_deg"]) * np.pi / 180

    def run(reuse):
        t = tomoengine(Nx, N, ang)
        t.set_option("fp_reuse", reuse)
        t.set_tilt_series(g["b"])
        out = []
        for k in range(4):                                   # the plain driver loop: step, data_distance, step, ...
            t.SIRT(1)
            out.append(t.data_distance())
        out.append(t.get_volume())
        t.CGLS(2)                                            # restart projects the volume the last data_distance projected
        out.append(t.get_volume())
        hazards = [lambda: t.set_volume(g["x_sart"]), lambda: t.set_recon(g["x_sart"][1], 1), lambda: t.tv_gd(2, 0.05),
                   lambda: t.tv_fgp(2, 0.05), lambda: t.SART(0.5, 1), lambda: t.restart_recon(), lambda: t.positivity(),
                   lambda: (t.set_volume(g["sart_b025"], VOL_TEMP), t.be.c("forward_projection", VOL_TEMP, SINO_G)),
                   lambda: t.be.c("set_sinogram", SINO_G, g["b"].ctypes.data), lambda: t.soft_threshold(0.01),
                   lambda: (t.initialize_fista(), t.fista_momentum(0.3), t.remove_momentum())]
        for h in hazards:                                    # data_distance, THEN something that changes recon or G, then a step
            t.set_volume(g["sart_b1"])
            t.data_distance()
            h()
            t.SIRT(1)
            out.append(t.get_volume())
        t.set_volume(g["sart_b1"])                           # a copy inherits the claim (multimodal::data_fusion's pattern)
        t.data_distance()
        t.be.c("copy_volume", VOL_TEMP, VOL_RECON)
        t.be.c("sirt_data", VOL_TEMP, 0, 2)
        out.append(t.get_volume(VOL_TEMP))
        t.data_distance_begin(VOL_TEMP)                      # projected on the second stream, reused on the main one
        t.be.c("sirt_data", VOL_TEMP, 0, 1)
        out.append(t.get_volume(VOL_TEMP))
        return out
    a, b = run(1), run(0)
    for i, (u, v) in enumerate(zip(a, b)):
        assert np.array_equal(np.asarray(u), np.asarray(v)), i


def test_fista_projects_the_extrapolated_point_by_linearity(gpu, golden):
    """tomo_fista_project_yk: A yk = (1 + beta) A r - beta A r_old from the two projections the cost evaluations made.  The shortcut
    must (a) actually be taken in the driver loop from the second iteration on, (b) stay within the parity tolerance of the loop
    that projects yk every time, and (c) refuse whenever a volume it relies on was touched in between."""
    from tomo_tv_amd._lib import SINO_G, SINO_YK_MODEL, VOL_RECON, VOL_YK, VOL_RECON_OLD
    N, P, Nx = 32, 9, 4
    A = golden(f"A_N{N}_P{P}.npz")
    g = golden(f"trace_N{N}_P{P}_Nx{Nx}.npz")
    ang = np.asarray(A["angles_deg"]) * np.pi / 180

    def loop(reuse, niter=6, meddle=None):
        t = tomoengine(Nx, N, ang)
        t.set_option("fp_reuse", reuse)
        t.set_tilt_series(g["b"])
        pytvlib.initialize_algorithm(t, "fista")
        t0, taken, costs = 1.0, [], []
        for k in range(niter):
            pytvlib.run(t, "fista")
            t.tv_fgp(5, 0.1, vol=VOL_YK)
            tk = 0.5 * (1 + np.sqrt(1 + 4 * t0 ** 2))
            t.fista_momentum((t0 - 1) / tk)
            t0 = tk
            costs.append(0.5 * t.data_distance() ** 2 + 0.1 * t.tv())
            if meddle is not None and k == 3:
                meddle(t, "before")
            taken.append(t.fista_project_yk())
            if meddle is not None and k == 3:
                meddle(t, "after")
        return t.get_volume(), t.get_volume(VOL_YK), np.array(costs), taken
    x1, y1, c1, taken = loop(1)
    x0, y0, c0, never = loop(0)
    assert taken == [True] * 6 and never == [False] * 6      # (iteration 1: beta = 0, yk is r itself)
    assert rel_l2(x1, x0) < TOL and rel_l2(y1, y0) < TOL and np.allclose(c1, c0, rtol=1e-5)
    # the projection it leaves IS the projection of yk, to rounding
    t = tomoengine(Nx, N, ang)
    t.set_tilt_series(g["b"])
    pytvlib.initialize_algorithm(t, "fista")
    for beta in (0.0, 0.28, 0.43):
        pytvlib.run(t, "fista")
        t.tv_fgp(3, 0.1, vol=VOL_YK)
        t.fista_momentum(beta)
        t.data_distance()
        g_before = t.get_model_projections()             # (reading G drops its claim: project recon again)
        t.data_distance()
        assert t.fista_project_yk()
        lin = np.zeros((Nx, N * P), np.float32)
        t.be.c("get_sinogram", SINO_YK_MODEL, lin.ctypes.data)
        # the model sinogram stays A * recon after the shortcut (tomoengine.cpp:410-427,459), bit for bit
        assert np.array_equal(t.get_model_projections(), g_before)
        t.be.c("forward_projection", VOL_RECON, SINO_G)
        assert np.array_equal(t.get_model_projections(), g_before)
        t.be.c("forward_projection", VOL_YK, SINO_G)
        direct = np.zeros_like(lin)
        t.be.c("get_sinogram", SINO_G, direct.ctypes.data)
        assert rel_l2(lin, direct) < 1e-6
    # hazards: anything that changes recon / yk / recon_old between the pieces makes the next shortcut refuse (and the loop
    # that follows equals the always-project loop given the same meddling)
    def hazard(fn, when):
        def meddle(t, at):
            if at == when:
                fn(t)
        a = loop(1, meddle=meddle)
        b = loop(0, meddle=meddle)
        assert rel_l2(a[0], b[0]) < TOL and rel_l2(a[1], b[1]) < TOL, (fn, when)
        return a[3]
    tk_before = hazard(lambda t: t.set_volume(g["sart_b1"]), "before")            # recon changed after its projection
    assert tk_before[3] is False
    tk = hazard(lambda t: t.set_volume(g["sart_b1"], VOL_YK), "before")           # yk is no longer the extrapolated point
    assert tk[3] is False
    tk = hazard(lambda t: t.set_volume(g["sart_b1"]), "after")                    # recon (-> recon_old) changed after saving A r
    assert tk[3] is True and tk[4] is False and tk[5] is True
    tk = hazard(lambda t: t.set_volume(g["sart_b025"], VOL_RECON_OLD), "after")   # recon_old is not the iterate that was projected
    assert tk[4] is False
    tk = hazard(lambda t: t.set_volume(g["sart_b025"], VOL_YK), "after")          # yk replaced after A yk was formed: SIRT must project
    assert tk[3] is True


def ulp_noise(x, seed):
    """x moved by one float32 ulp in a random direction per element: the smallest possible input change."""
    rng = np.random.default_rng(seed)
    x = np.asarray(x, np.float32)
    return np.nextafter(x, np.where(rng.random(x.shape) < 0.5, -np.inf, np.inf).astype(np.float32))


def asd_loop(t, niter, norm, dd_fn):
    """examples/sim_ASD.py:66-94 with the defaults of gpu/reconstructor.py:158-161."""
    beta, dPOCS = 0.25, 0.0
    dd, tv = [], []
    for i in range(niter):
        t.copy_recon()
        t.SART(beta, 1)
        beta *= 0.9985
        dp = t.matrix_2norm()
        if i == 0:
            dPOCS = dp * 0.2
        dd.append(dd_fn(t) / norm)
        t.copy_recon()
        tv.append(t.tv_gd(10, dPOCS))
        dg = t.matrix_2norm()
        if dg > dp * 0.95 and dd[-1] > 0.025:
            dPOCS *= 0.95
    return np.array(dd), np.array(tv)


@pytest.mark.parametrize("N,P,Nx", SHAPES)
def test_asd_pocs_free_running_trace(gpu, golden, N, P, Nx):
    """examples/sim_ASD.py:66-94 loop through TomoGPU.asd_pocs, 20 iterations, defaults.

    TV descent with eps=1e-8 follows sign-like gradients (Lipschitz constant ~1/sqrt(eps)) with a normalised step of
    fixed length dPOCS, so the loop is ill-conditioned: it amplifies ANY fp32 rounding difference.  The bound is
    therefore conditioning-aware: the oracle is run again on tilt series moved by ONE ulp per sample (eight seeds), and
    the HIP path must stay within 3x of the farthest the oracle moves itself (profiles/r03_asd_parity.md: HIP sits at
    0.03 ... 1.2x of that in every row).  The first iteration is held to 2e-5."""
    from tomo_tv_amd.reconstructor import TomoGPU
    A = golden(f"A_N{N}_P{P}.npz")
    g = golden(f"trace_N{N}_P{P}_Nx{Nx}.npz")
    ts = g["b"].reshape(Nx, P, N).transpose(0, 2, 1)             # (Nslice, Nray, Nangles)
    rec = TomoGPU(A["angles_deg"], ts)
    rec.tomo.tv_eps = 1e-8
    dd, tv = rec.asd_pocs(Niter=20)
    got = rec.get_recon()
    self_move = self_dd = self_tv = 0.0
    for seed in range(8):                # the oracle against itself, tilt series moved by one ulp: eight seeds, the largest deviation
        ref = oracle.ctvlib(Nx, N, P)
        ref.load_A(A["A"])
        ref.tv_eps = 1e-8
        ref.set_tilt_series(ulp_noise(g["b"], seed))
        dd2, tv2 = asd_loop(ref, 20, Nx * N * P, lambda t: t.data_distance(normalize=False))
        self_move = max(self_move, rel_l2(ref.recon, g["asd_final"]))
        self_dd = max(self_dd, np.max(np.abs(dd2 - g["asd_dd"]) / g["asd_dd"]))
        self_tv = max(self_tv, np.max(np.abs(tv2 - g["asd_tv"]) / g["asd_tv"]))
    assert np.allclose(dd[:1], g["asd_dd"][:1], rtol=2e-5) and np.allclose(tv[:1], g["asd_tv"][:1], rtol=2e-5)
    assert np.max(np.abs(dd - g["asd_dd"]) / g["asd_dd"]) <= max(2e-5, 3 * self_dd)
    assert np.max(np.abs(tv - g["asd_tv"]) / g["asd_tv"]) <= max(2e-5, 3 * self_tv)
    assert rel_l2(got, g["asd_final"]) <= max(5e-5, 3 * self_move), (rel_l2(got, g["asd_final"]), self_move)


@pytest.mark.parametrize("N,P,Nx", SHAPES)
def test_asd_pocs_teacher_forced(gpu, golden, N, P, Nx):
    """Every ASD-POCS outer iteration restarted from the oracle's iterate.  SART sweep: 1e-5.  After the 10 TV
    descent steps: 1e-5, or 5x the oracle's own response to a one-ulp change of its input (see above)."""
    A = golden(f"A_N{N}_P{P}.npz")
    g = golden(f"trace_N{N}_P{P}_Nx{Nx}.npz")
    dev = tomoengine(Nx, N, np.asarray(A["angles_deg"]) * np.pi / 180)
    dev.set_tilt_series(g["b"])
    dev.tv_eps = 1e-8
    ref, ref2 = oracle.ctvlib(Nx, N, P), oracle.ctvlib(Nx, N, P)
    for r in (ref, ref2):
        r.load_A(A["A"])
        r.set_tilt_series(g["b"])
        r.tv_eps = 1e-8
    beta, dPOCS = 0.25, 0.0
    for i in range(12):
        dev.set_volume(ref.recon)
        ref2.recon[:] = ref.recon
        out = []
        for t in (dev, ref, ref2):
            t.copy_recon()
            t.SART(beta, 1)
            dp = t.matrix_2norm()
            dd = t.data_distance() if t is dev else t.data_distance(normalize=False)
            if i == 0 and t is ref:
                dPOCS = dp * 0.2
            out.append([dp, dd])
        assert np.allclose(out[0], out[1], rtol=1e-5), f"iteration {i}: {out}"
        assert rel_l2(dev.get_volume(), ref.recon) < TOL, f"SART sweep {i}"
        ref2.recon[:] = ulp_noise(ref2.recon, i)
        for t, o in zip((dev, ref, ref2), out):
            t.copy_recon()
            o.append(t.tv_gd(10, dPOCS))
            o.append(t.matrix_2norm())
        self_move = rel_l2(ref2.recon, ref.recon)
        assert abs(out[0][2] - out[1][2]) <= 1e-5 * out[1][2]          # TV before descent
        assert rel_l2(dev.get_volume(), ref.recon) <= max(TOL, 5 * self_move), (i, self_move)
        beta *= 0.9985
        if out[1][3] > out[1][0] * 0.95 and out[1][1] / (Nx * N * P) > 0.025:
            dPOCS *= 0.95


@pytest.mark.parametrize("N,P,Nx", SHAPES[1:])
def test_fista_trace(gpu, golden, N, P, Nx):
    from tomo_tv_amd.reconstructor import TomoGPU
    A = golden(f"A_N{N}_P{P}.npz")
    g = golden(f"trace_N{N}_P{P}_Nx{Nx}.npz")
    ts = g["b"].reshape(Nx, P, N).transpose(0, 2, 1)
    rec = TomoGPU(A["angles_deg"], ts)
    rec.tomo.tv_eps = 1e-8
    cost = rec.fista(Niter=10, lambda_param=0.01, nTViter=5)
    assert np.allclose(cost, g["fista_cost"], rtol=2e-5)
    assert rel_l2(rec.get_recon(), g["fista_final"]) < 2e-5


def test_config1_sirt50(gpu, golden):
    """BASELINE config 1: 2-D 256x256 Shepp-Logan, 50 tilts, SIRT x50 (Nslice = 1: one real slice, 63 padded)."""
    from tomo_tv_amd.phantom import shepp_logan
    g = golden("trace_config1_sirt50.npz")
    N, P = 256, 50
    dev = ctvlib(1, N, P)
    pytvlib.initialize_ctvlib(dev, "SIRT", N, np.linspace(-70, 70, P))
    assert abs(dev.lipschits() - float(g["lipschitz"])) <= 1e-6 * float(g["lipschitz"])
    pytvlib.create_projections(dev, shepp_logan(N)[None])
    beta = 1.0 / dev.lipschits()
    dd = []
    for _ in range(50):
        dev.SIRT(beta)
        dd.append(dev.data_distance())
    assert np.allclose(dd, g["dd"], rtol=2e-5)
    assert rel_l2(dev.get_recon(0)[None], g["recon"]) < TOL


@pytest.mark.parametrize("Nx", [1, 3, 64, 100, 128, 256])
def test_ragged_slice_counts_match_oracle(gpu, Nx):
    """Every vector width (sx%256 -> float4, %128 -> float2, else scalar) and padded slabs give the same numbers."""
    N, P = 32, 7
    ang = np.linspace(-60, 60, P)
    A = oracle.parallel_ray(N, ang)
    x = ellipsoids(Nx, N, seed=7)
    ref = oracle.ctvlib(Nx, N, P)
    ref.load_A(A)
    ref.original_volume = x.copy()
    ref.create_projections()
    dev = tomoengine(Nx, N, ang * np.pi / 180)
    dev.set_volume(x, VOL_ORIGINAL)
    dev.create_projections()
    assert rel_l2(dev.get_projections(), ref.b) < TOL
    dev.SART(0.5, 1)
    ref.SART(0.5, 1)
    assert rel_l2(dev.get_volume(), ref.recon) < TOL
    ref.tv_eps = dev.tv_eps
    assert abs(dev.tv() - ref.tv()) <= 1e-5 * ref.tv()
    tv_d, tv_r = dev.tv_gd(3, 0.1), ref.tv_gd(3, 0.1)
    assert abs(tv_d - tv_r) <= 1e-5 * tv_r
    assert rel_l2(dev.get_volume(), ref.recon) < TOL
    dev.tv_fgp(3, 0.05)
    ref.tv_fgp(3, 0.05)
    assert rel_l2(dev.get_volume(), ref.recon) < TOL
    for s in {0, Nx - 1}:
        assert np.array_equal(dev.get_recon(s), dev.get_volume()[s])


@pytest.mark.parametrize("N,P,Nx", [(9, 4, 5), (15, 3, 70), (33, 5, 130), (24, 5, 130)])
def test_odd_sizes_whose_grids_round_up(gpu, N, P, Nx):
    """Sizes where (pixels/4 x chunks) or (rays/4 x chunks) is not a multiple of the 4 waves of a workgroup."""
    ang = np.linspace(-63.5, 58.0, P)
    A = oracle.parallel_ray(N, ang)
    x = ellipsoids(Nx, N, seed=13)
    ref = oracle.ctvlib(Nx, N, P)
    ref.load_A(A)
    ref.original_volume = x.copy()
    ref.create_projections()
    for rep in range(3):                                   # stale device memory must not matter
        dev = ctvlib(Nx, N, P)
        dev.load_A(A)
        pytvlib.create_projections(dev, x)
        assert rel_l2(dev.get_projections(), ref.b) < TOL
    dev.SIRT(1.0 / dev.lipschits())
    r2 = oracle.ctvlib(Nx, N, P)
    r2.load_A(A)
    r2.set_tilt_series(ref.b)
    r2.SIRT(1.0 / r2.lipschits())
    assert rel_l2(dev.get_volume(), r2.recon) < TOL
    t = tomoengine(Nx, N, ang * np.pi / 180)
    t.set_tilt_series(ref.b)
    t.SART(0.5, 2)
    t.SIRT(2)
    r2.restart_recon()
    r2.SART(0.5, 2)
    r2.SIRT_norm(2)
    assert rel_l2(t.get_volume(), r2.recon) < TOL
    assert abs(t.data_distance() - r2.data_distance(normalize=False)) <= 2e-5 * t.data_distance()


def test_slice_set_get_roundtrip_and_errors(gpu):
    N, P, Nx = 16, 3, 5
    dev = tomoengine(Nx, N, np.deg2rad([-30.0, 0.0, 30.0]))
    rng = np.random.default_rng(0)
    vol = rng.random((Nx, N, N), dtype=np.float32)
    for s in range(Nx):
        dev.set_recon(vol[s], s)
    assert np.array_equal(dev.get_volume(), vol)
    assert np.array_equal(dev.get_recon(3), vol[3])
    with pytest.raises(IndexError):
        dev.get_recon(Nx)
    with pytest.raises(ValueError):
        dev.set_tilt_series(np.zeros((Nx, N * P + 1)))
    with pytest.raises(ValueError):
        dev.set_recon(np.zeros((N, N + 1)), 0)
    dev.restart_recon()
    assert dev.get_volume().max() == 0
    with pytest.raises(TomoError):
        tomoengine(0, N, np.zeros(1))


def test_poisson_ml_step(gpu, golden):
    N, P, Nx = 32, 9, 4
    A = golden(f"A_N{N}_P{P}.npz")
    g = golden(f"trace_N{N}_P{P}_Nx{Nx}.npz")
    dev = tomoengine(Nx, N, np.asarray(A["angles_deg"]) * np.pi / 180)
    b = g["b"] / g["b"].max()
    dev.set_tilt_series(b)
    dev.initialize_poisson_ML()
    ref = oracle.ctvlib(Nx, N, P)
    ref.load_A(A["A"])
    ref.set_tilt_series(b)
    x = g["x_sart"]
    dev.set_volume(x)
    ref.recon[:] = x
    L = ref.lipschits()
    for _ in range(3):
        c_dev = dev.poisson_ML(0.5)
        c_ref = ref.poisson_ML(0.5, L)
        assert abs(c_dev - c_ref) <= 1e-5 * abs(c_ref)
    assert rel_l2(dev.get_volume(), ref.recon) < TOL


def test_load_A_rejects_out_of_range_and_unsupported(gpu):
    N, P = 8, 2
    A = oracle.parallel_ray(N, np.array([0.0, 30.0]))
    bad = A.copy()
    bad[1, 0] = N * N
    t = ctvlib(2, N, P)
    with pytest.raises(TomoError):
        t.load_A(bad)
    # three rays of one angle through a pixel is outside the supported geometry
    extra = np.concatenate([A, np.array([[2.0], [A[1, 0]], [0.5]], np.float32),
                            np.array([[3.0], [A[1, 0]], [0.5]], np.float32)], axis=1)
    with pytest.raises(TomoError, match="more than two rays"):
        ctvlib(2, N, P).load_A(extra)


def test_cgls_matches_per_slice_textbook_cgls(gpu, golden):
    """CGLS restarted per call, independently per slice (alpha, beta per slice), positivity at the end."""
    N, P, Nx = 32, 9, 4
    A = golden(f"A_N{N}_P{P}.npz")
    g = golden(f"trace_N{N}_P{P}_Nx{Nx}.npz")
    dev = tomoengine(Nx, N, np.asarray(A["angles_deg"]) * np.pi / 180)
    dev.set_tilt_series(g["b"])
    ref = oracle.ctvlib(Nx, N, P)
    ref.load_A(A["A"])
    F = np.float32

    def fp(v):
        ref.recon[:] = v
        ref.forward_projection()
        return ref.g.copy()
    x = np.zeros((Nx, N, N), F)
    for call in range(2):
        niter = 4
        dev.CGLS(niter)
        r = (g["b"] - fp(x)).astype(F)
        z = ref.back_projection(r)
        p = z.copy()
        gam = (z.astype(np.float64) ** 2).sum(axis=(1, 2))
        for _ in range(niter):
            w = fp(p)
            den = (w.astype(np.float64) ** 2).sum(axis=1)
            alpha = np.where(den > 0, gam / np.where(den > 0, den, 1), 0).astype(F)      # empty slices: 0/0 := 0
            x = (x + alpha[:, None, None] * p).astype(F)
            r = (r - alpha[:, None] * w).astype(F)
            z = ref.back_projection(r)
            gnew = (z.astype(np.float64) ** 2).sum(axis=(1, 2))
            beta = np.where(gam > 0, gnew / np.where(gam > 0, gam, 1), 0).astype(F)
            gam = gnew
            p = (z + beta[:, None, None] * p).astype(F)
        x = np.maximum(x, 0)
        assert rel_l2(dev.get_volume(), x) < 5e-5, call
    assert dev.data_distance() < 0.4 * np.linalg.norm(g["b"])


@pytest.mark.parametrize("name", ["ram-lak", "shepp-logan", "hamming", "kaiser"])
def test_wbp_reconstructs_full_angular_range(gpu, name):
    """recon = pi/P A^T (h * b): with 0..180 degree coverage it approximates the object (scale and shape)."""
    from tomo_tv_amd import pytvlib as ptl
    from tomo_tv_amd.engine import fbp_filter_taps
    N, P, Nx = 64, 90, 3
    ang = np.linspace(-90, 88, P)
    x = ellipsoids(Nx, N, seed=2, k=6)
    dev = tomoengine(Nx, N, np.deg2rad(ang))
    dev.set_volume(x, VOL_ORIGINAL)
    dev.create_projections()
    b = dev.get_projections()
    ptl.initialize_algorithm(dev, "FBP", name)
    ptl.run(dev, "FBP")
    rec = dev.get_volume()
    # reference computation of the same definition with numpy + the oracle's A^T
    taps = fbp_filter_taps(N, name)
    idx = np.abs(np.arange(N)[:, None] - np.arange(N)[None, :])
    Hm = taps[idx].astype(np.float64)
    filt = np.einsum("jk,spk->spj", Hm, b.reshape(Nx, P, N).astype(np.float64)).reshape(Nx, -1).astype(np.float32)
    ref = oracle.ctvlib(Nx, N, P)
    ref.load_A(oracle.parallel_ray(N, ang))
    want = np.maximum(ref.back_projection(filt) * np.float32(np.pi / P), 0)
    assert rel_l2(rec, want) < 2e-5
    inner = x > 0
    corr = np.corrcoef(rec.ravel(), x.ravel())[0, 1]
    assert corr > 0.9 and 0.7 < rec[inner].mean() / x[inner].mean() < 1.3, (corr, rec[inner].mean() / x[inner].mean())


def _independent_fbp_taps(n, name):
    """The 12 filters of tomofusion/pytvlib.py:33-36 written out here, NOT taken from the product: the band-limited ramp of Kak &
    Slaney (h[0] = 1/4, h[k odd] = -1/(pi k)^2, h[k even] = 0) on a periodic extension of L = 4 * 2^ceil(log2 n) samples, its
    cosine transform multiplied by the window W(f), f = frequency / Nyquist, and transformed back -- by explicit cosine sums
    (no FFT, no shared helper).  Window definitions: the textbook ones (Harris 1978 for the cosine-sum windows)."""
    L = 4 * 2 ** int(np.ceil(np.log2(max(n, 2))))
    k = np.arange(L)
    kk = np.minimum(k, L - k).astype(np.float64)
    h = np.where(kk == 0, 0.25, np.where(kk % 2 == 1, -1.0 / (np.pi * np.maximum(kk, 1)) ** 2, 0.0))
    m = np.arange(L // 2 + 1)
    C = np.cos(2 * np.pi * np.outer(m, k) / L)                  # (L/2+1, L)
    H = C @ h                                                   # real-even sequence -> real cosine transform
    f = m / (L / 2.0)
    W = {
        "ram-lak": np.ones_like(f),
        "shepp-logan": np.where(f == 0, 1.0, np.sin(np.pi * f / 2) / np.where(f == 0, 1, np.pi * f / 2)),
        "cosine": np.cos(np.pi * f / 2),
        "hamming": 0.54 + 0.46 * np.cos(np.pi * f),
        "lanczos": np.where(f == 0, 1.0, np.sin(np.pi * f) / np.where(f == 0, 1, np.pi * f)),
        "triangular": 1 - f,
        "gaussian": np.exp(-0.5 * (f / 0.4) ** 2),
        "blackman": 0.42 + 0.5 * np.cos(np.pi * f) + 0.08 * np.cos(2 * np.pi * f),
        "nuttall": 0.355768 + 0.487396 * np.cos(np.pi * f) + 0.144232 * np.cos(2 * np.pi * f) + 0.012604 * np.cos(3 * np.pi * f),
        "blackman-harris": 0.35875 + 0.48829 * np.cos(np.pi * f) + 0.14128 * np.cos(2 * np.pi * f) + 0.01168 * np.cos(3 * np.pi * f),
        "kaiser": np.i0(8.6 * np.sqrt(np.clip(1 - f * f, 0, 1))) / np.i0(8.6),
        "parzen": np.where(f <= 0.5, 1 - 6 * f ** 2 * (1 - f), 2 * (1 - f) ** 3),
    }[name]
    G = H * W
    wgt = np.full(L // 2 + 1, 2.0)
    wgt[0] = wgt[-1] = 1.0                                      # the mirrored half of the spectrum
    taps = (C[:, :n] * (G * wgt)[:, None]).sum(axis=0) / L
    return taps


@pytest.mark.parametrize("name", ["ram-lak", "shepp-logan", "hamming", "cosine", "parzen", "lanczos", "triangular", "gaussian",
                                  "blackman", "nuttall", "blackman-harris", "kaiser"])
def test_wbp_all_twelve_filters_against_independent_taps(gpu, name):
    """Every filter name of tomofusion/pytvlib.py:33-36 through TomoGPU.wbp's path, against recon = pi/P A^T (h * b) evaluated
    with taps built in THIS file (explicit cosine sums) and the oracle's A^T (tomoengine.cpp:317-347; ASTRA's own filter
    construction is absent: parity unpinned, the definition is DESIGN.md's)."""
    from tomo_tv_amd import pytvlib as ptl
    assert name in ptl.wbp_filters()
    N, P, Nx = 48, 24, 2
    ang = np.linspace(-90, 82.5, P)
    x = ellipsoids(Nx, N, seed=6, k=5)
    dev = tomoengine(Nx, N, np.deg2rad(ang))
    dev.set_volume(x, VOL_ORIGINAL)
    dev.create_projections()
    b = dev.get_projections()
    ptl.initialize_algorithm(dev, "FBP", name)
    ptl.run(dev, "FBP")
    taps = _independent_fbp_taps(N, name)
    idx = np.abs(np.arange(N)[:, None] - np.arange(N)[None, :])
    filt = np.einsum("jk,spk->spj", taps[idx], b.reshape(Nx, P, N).astype(np.float64)).reshape(Nx, -1).astype(np.float32)
    ref = oracle.ctvlib(Nx, N, P)
    ref.load_A(oracle.parallel_ray(N, ang))
    want = np.maximum(ref.back_projection(filt) * np.float32(np.pi / P), 0)
    assert rel_l2(dev.get_volume(), want) < 2e-5, name


def test_update_projection_angles_and_poisson_noise(gpu):
    """Dynamic tilt append keeps the reconstruction (tomoengine.cpp:128-149); seeded Poisson noise keeps the total."""
    N, Nx = 32, 5
    x = ellipsoids(Nx, N, seed=9)
    a1, a2 = np.linspace(-60, 0, 5), np.linspace(-60, 60, 9)
    dev = tomoengine(Nx, N, np.deg2rad(a1))
    dev.set_volume(x, VOL_ORIGINAL)
    dev.create_projections()
    dev.SART(0.5, 2)
    before = dev.get_volume()
    dev.update_projection_angles(np.deg2rad(a2))
    assert np.array_equal(dev.get_volume(), before) and np.array_equal(dev.get_volume(VOL_ORIGINAL), x)
    dev.create_projections()
    ref = oracle.ctvlib(Nx, N, 9)
    ref.load_A(oracle.parallel_ray(N, a2))
    ref.original_volume = x.copy()
    ref.create_projections()
    assert rel_l2(dev.get_projections(), ref.b) < TOL
    dev.SART(0.5, 1)
    ref.recon[:] = before
    ref.SART(0.5, 1)
    assert rel_l2(dev.get_volume(), ref.recon) < TOL
    b0 = dev.get_projections()
    dev.poisson_noise(50, seed=1)
    b1 = dev.get_projections()
    assert abs(b1.sum(dtype=np.float64) / b0.sum(dtype=np.float64) - 1) < 0.02 and not np.array_equal(b0, b1)
    dev.set_tilt_series(b0)
    dev.poisson_noise(50, seed=1)
    assert np.array_equal(dev.get_projections(), b1)            # reproducible


def test_failed_geometry_rebuild_leaves_the_engine_whole(gpu):
    """update_proj_angles / update_projection_angles with a matrix the engine rejects must raise AND leave the
    engine exactly as it was -- same geometry, same volumes, still usable (ctvlib.cpp:317-333 keeps recon across the swap)."""
    N, Nx = 16, 3
    a1, a2 = np.linspace(-60, 60, 5), np.linspace(-60, 60, 7)
    x = ellipsoids(Nx, N, seed=2)
    dev = ctvlib(Nx, N, 5)
    dev.load_A(oracle.parallel_ray(N, a1))
    dev.set_volume(x, VOL_ORIGINAL)
    dev.create_projections()
    dev.SIRT(0.01)
    before, b = dev.get_volume(), dev.get_projections()
    bad = oracle.parallel_ray(N, a2)
    bad[1, 3] = N * N                                   # column index out of range
    with pytest.raises(TomoError):
        dev.update_proj_angles(bad, 7)
    assert (dev.Nproj, dev.Nrow) == (5, 5 * N)
    assert np.array_equal(dev.get_volume(), before) and np.array_equal(dev.get_projections(), b)
    dev.SIRT(0.01)                                      # still a working engine on the old geometry
    ref = oracle.ctvlib(Nx, N, 5)
    ref.load_A(oracle.parallel_ray(N, a1))
    ref.set_tilt_series(b)
    ref.recon[:] = before
    ref.SIRT(0.01)
    assert rel_l2(dev.get_volume(), ref.recon) < TOL
    dev.update_proj_angles(oracle.parallel_ray(N, a2), 7)      # and a good matrix still goes in
    assert dev.Nproj == 7 and np.array_equal(dev.get_volume(), dev.get_volume())


def test_rebuilding_the_stream_owner_of_a_multimodal_pair(gpu):
    """the HAADF engine of `multimodal` runs on the chemical engine's stream; rebuilding the chemical engine's
    geometry must hand the HAADF engine the new stream before the old one is destroyed."""
    from tomo_tv_amd.chemistry import multimodal
    N, Nx = 16, 4
    mm = multimodal(Nx, N, 1, np.deg2rad(np.linspace(-60, 60, 5)), np.deg2rad(np.linspace(-50, 50, 4)))
    x = ellipsoids(Nx, N, seed=4)
    mm.he.set_volume(x, VOL_ORIGINAL)
    mm.he.create_projections()
    want = mm.he.get_projections()
    mm.ce.update_projection_angles(np.deg2rad(np.linspace(-50, 50, 6)))
    mm.he.create_projections()                          # would run on a destroyed stream without the hand-over
    mm.he.synchronize()
    assert np.array_equal(mm.he.get_projections(), want)
    mm.he.update_projection_angles(np.deg2rad(np.linspace(-60, 60, 7)))
    mm.he.create_projections()
    assert mm.he.get_projections().shape == (Nx, 7 * N)


@pytest.mark.parametrize("N,P,Nx", [(40, 7, 70), (96, 13, 128), (33, 5, 256)])
def test_tile_projectors_match_row_and_pixel_driven_forms(gpu, N, P, Nx):
    """k_fp_tile/k_fp_tile_reduce and k_bp_tile against the ray-driven FP and the pixel-driven BP they replace:
    BP does the same FMAs in the same order (bit-identical); FP sums tile partials in a different order (<= 1e-6)."""
    ang = np.linspace(-68, 71, P)
    x = ellipsoids(Nx, N, seed=5)
    out = {}
    for tile in (0, 1):
        t = tomoengine(Nx, N, ang * np.pi / 180)
        t.set_option("fp_tile", tile)
        t.set_option("bp_tile", tile)
        t.set_volume(x, VOL_ORIGINAL)
        t.create_projections()
        b = t.get_projections()
        t.SIRT(3)
        out[tile] = (b, t.get_volume(), t.data_distance())
    assert rel_l2(out[1][0], out[0][0]) < 1e-6
    assert rel_l2(out[1][1], out[0][1]) < 1e-5
    # BP alone on identical input: bit-identical
    t0 = tomoengine(Nx, N, ang * np.pi / 180); t1 = tomoengine(Nx, N, ang * np.pi / 180)
    for t, tile in ((t0, 0), (t1, 1)):
        t.set_option("fp_tile", 0); t.set_option("bp_tile", tile)
        t.set_tilt_series(out[0][0]); t.SIRT(2)
    assert np.array_equal(t0.get_volume(), t1.get_volume())


@pytest.mark.parametrize("N,Nx,iters", [(40, 70, 7), (33, 130, 4), (64, 64, 3), (17, 5, 6), (96, 128, 10), (8, 200, 5), (72, 1, 8)])
def test_two_fgp_iterations_per_pass_equal_two_passes(gpu, N, Nx, iters):
    """k_fgp_fused2 (round 4: P carried through two FGP iterations on chip, halo cells recomputed) == two passes of k_fgp_fused,
    bit for bit: odd and even iteration counts (the odd one out runs the one-iteration kernel), tiles cut by every face, one slice."""
    x = ellipsoids(Nx, N, seed=9) + np.float32(0.1) * np.random.default_rng(2).random((Nx, N, N), dtype=np.float32)
    out = {}
    for pair in (1, 0):
        t = tomoengine(Nx, N, np.deg2rad(np.array([-20.0, 35.0])))
        t.set_option("fgp_pair", pair)
        t.set_volume(x, VOL_RECON)
        tv0 = t.tv_fgp(iters, 0.07)
        out[pair] = (tv0, t.get_volume())
    assert out[1][0] == out[0][0]
    assert np.array_equal(out[1][1], out[0][1])


@pytest.mark.parametrize("N,P,Nx", [(40, 7, 70), (96, 13, 128), (33, 5, 256), (16, 1, 128), (128, 31, 384), (50, 4, 100), (64, 9, 192), (24, 200, 128)])
def test_wave_uniform_back_projector_is_bit_identical(gpu, N, P, Nx):
    """k_bp_list (round 4: a wave = 128 slices of 32 pixels, the matrix as scalar-loaded entry lists of nonzero weights, accumulators
    picked by the VGPR index mode) does the nonzero FMAs of k_bp_tile and of the pixel-driven k_bp_all in the same order: the same
    bits, also where the slice count is not a multiple of 128 (the engine then falls back to k_bp_tile for the whole slab)."""
    ang = np.linspace(-68, 71, P) if P > 1 else np.array([-23.0])
    b = np.random.default_rng(3).standard_normal((Nx, P * N)).astype(np.float32)
    vols = {}
    for form in ("wave", "tile", "pixel"):
        t = tomoengine(Nx, N, ang * np.pi / 180)
        t.set_option("fp_tile", 0)
        t.set_option("bp_tile", 0 if form == "pixel" else 1)
        t.set_option("bp_list", 1 if form == "wave" else 0)
        assert t.get_option("bp_list_ready") == (1 if P <= 192 else 0)      # (more angles than the kernel keeps list bounds for: k_bp_tile runs)
        t.set_tilt_series(b)
        t.SIRT(2)
        vols[form] = t.get_volume()
    assert np.array_equal(vols["wave"], vols["tile"])
    # (round 5: also k_bp_all's one-float-per-lane build, what odd chunk counts run: its accumulation is two written-out FMAs now --
    # the compiler had packed one of them into v_pk_mul_f32 + v_pk_add_f32, the 1-ulp exception this test carried in round 4)
    assert np.array_equal(vols["tile"], vols["pixel"])


@pytest.mark.parametrize("N,P,Nx,amax", [(40, 7, 70, 70), (96, 13, 128, 68), (33, 5, 256, 60), (16, 1, 64, 0), (64, 16, 64, 89), (128, 31, 64, 70),
                                          (50, 4, 100, 45), (24, 200, 128, 80)])
def test_strip_forward_projector_matches_tile_and_row_forms(gpu, monkeypatch, N, P, Nx, amax):
    """k_fp_strip (sheared strips, ray sums resident in registers; round 4: the all-angle forward projector of large slabs) against
    the tile-stationary and the ray-driven forms: the same matrix entries summed in a different order (<= 1e-6), in every epilogue
    mode the reduce kernel has (store, normalised residual inside SIRT, data distance, Poisson step), and against the oracle."""
    monkeypatch.setenv("TOMO_FP_STRIP", "1")        # (the engine builds the strip tables by itself only for large slabs)
    monkeypatch.setenv("TOMO_FP_LIST", "1")         # ... and their wave-uniform list form (k_fp_list; runs where the slab is whole 128-slice pieces)
    ang = np.linspace(-amax, amax + 1.5, P) if P > 1 else np.array([17.0])
    x = ellipsoids(Nx, N, seed=5)
    out = {}
    for form in ("list", "strip", "tile", "rows"):
        t = tomoengine(Nx, N, ang * np.pi / 180)
        assert t.get_option("fp_strip_ready") == 1 and t.get_option("fp_list_ready") == 1
        if form == "strip":
            t.set_option("fp_list", 0)
        elif form != "list":
            t.set_option("fp_tile", 1 if form == "tile" else 0)
        t.set_volume(x, VOL_ORIGINAL)
        t.create_projections()
        b = t.get_projections()
        t.SIRT(3)
        v = t.get_volume()
        dd = t.data_distance()
        t.initialize_poisson_ML()
        t.restart_recon()
        cost = t.poisson_ML(0.7)
        out[form] = (b, v, dd, t.get_volume(), cost)
    for k in range(5):                              # the list form against the strip form
        a, b = out["list"][k], out["strip"][k]
        assert (rel_l2(a, b) < 1e-6) if isinstance(a, np.ndarray) else (abs(a - b) <= 1e-5 * abs(b)), k
    for other in ("tile", "rows"):
        assert rel_l2(out["strip"][0], out[other][0]) < 1e-6, other
        assert rel_l2(out["strip"][1], out[other][1]) < 1e-5, other
        assert abs(out["strip"][2] - out[other][2]) <= 1e-5 * out[other][2], other
        assert rel_l2(out["strip"][3], out[other][3]) < 1e-5, other
        assert abs(out["strip"][4] - out[other][4]) <= 1e-5 * abs(out[other][4]), other
    ref = oracle.ctvlib(Nx, N, P)
    ref.load_A(oracle.parallel_ray(N, ang))
    ref.original_volume = x.copy()
    ref.create_projections()
    assert rel_l2(out["strip"][0], ref.b) < 1e-6
    assert rel_l2(out["list"][0], ref.b) < 1e-6


@pytest.mark.parametrize("N,P,Nx", [(40, 7, 70), (100, 9, 128), (16, 1, 64)])
def test_tile_sart_step_matches_ray_walk_form_and_oracle(gpu, N, P, Nx):
    """k_sart_tile (streamed tiles) against k_sart_seg (ray walk) and the oracle, natural and permuted angle order."""
    ang = np.linspace(-72, 66, P) if P > 1 else np.array([12.0])
    x = ellipsoids(Nx, N, seed=11)
    ref = oracle.ctvlib(Nx, N, P)
    ref.load_A(oracle.parallel_ray(N, ang))
    ref.original_volume = x.copy()
    ref.create_projections()
    order = np.random.default_rng(3).permutation(P).astype(np.int32)
    vols = {}
    for tile in (0, 1):
        t = tomoengine(Nx, N, ang * np.pi / 180)
        t.set_option("sart_resident", 0)          # the two STREAMED forms (the resident sweep: tests/test_gpu_sart_resident.py)
        t.set_option("sart_tile", tile)
        t.set_tilt_series(ref.b)
        t.SART(0.7, 2)
        a = t.get_volume()
        t.be.c("sart", VOL_RECON, 0.7, 1, order.ctypes.data)              # permuted sweep through the C ABI
        vols[tile] = (a, t.get_volume())
    ref.SART(0.7, 2)
    assert rel_l2(vols[1][0], ref.recon) < 1e-5 and rel_l2(vols[0][0], ref.recon) < 1e-5
    assert rel_l2(vols[1][0], vols[0][0]) < 2e-6
    assert rel_l2(vols[1][1], vols[0][1]) < 2e-6


def test_matrix_whose_ray_windows_defeat_the_tile_kernels(gpu):
    """A user matrix (load_A) with the rays of every angle shuffled: still <= 2 rays per pixel and angle, but the two rays
    of a pixel are far apart, so no tile's ray window fits the LDS budget of k_bp_tile / k_sart_tile.  The engine must
    fall back to the pixel-driven / ray-walk kernels; the tile forward projector handles any matrix."""
    N, P, Nx = 48, 4, 66
    A = oracle.parallel_ray(N, np.array([-60.0, -10.0, 25.0, 70.0]))
    rng = np.random.default_rng(9)
    rows = A[0].astype(np.int64)
    perm = np.concatenate([i * N + rng.permutation(N) for i in range(P)])
    Ash = A.copy()
    Ash[0] = perm[rows].astype(np.float32)
    dev, ref = ctvlib(Nx, N, P), oracle.ctvlib(Nx, N, P)
    dev.load_A(Ash); ref.load_A(Ash)
    x = ellipsoids(Nx, N, seed=2)
    dev.set_volume(x, VOL_ORIGINAL); ref.original_volume = x.copy()
    dev.create_projections(); ref.create_projections()
    assert rel_l2(dev.get_projections(), ref.b) < 1e-5
    dev.be.c("sart", VOL_RECON, 0.8, 2, None); ref.SART(0.8, 2)        # tomo_sart through the C ABI
    assert rel_l2(dev.get_volume(), ref.recon) < 1e-5
    L = dev.get_lipschitz()
    for _ in range(3):
        dev.SIRT(1.0 / L); ref.SIRT(1.0 / L)
    assert rel_l2(dev.get_volume(), ref.recon) < 1e-5
    dev.row_inner_product(); ref.row_inner_product()
    dev.ART(0.6); ref.ART(0.6)                   # neighbouring rays are not neighbours here: the row-sequential kernel runs
    assert rel_l2(dev.get_volume(), ref.recon) < 1e-5


@pytest.mark.parametrize("opts", [{}, {"sart_resident": 0}, {"sart_resident": 0, "sart_tile": 0}, {"sart_fused": 0}])
def test_tracked_sart_and_tv_equal_the_separate_calls(gpu, opts):
    """tomo_sart_tracked / tomo_tv_gd_tracked = SART; matrix_2norm; copy_recon and tv_gd; matrix_2norm; copy_recon
    (examples/sim_ASD.py:68-88) with the norm and the snapshot formed inside the last pass: same volumes bit for bit."""
    N, P, Nx = 48, 7, 130
    ang = np.linspace(-70, 70, P)
    x = ellipsoids(Nx, N, seed=8)
    a, b = tomoengine(Nx, N, ang * np.pi / 180), tomoengine(Nx, N, ang * np.pi / 180)
    for t in (a, b):
        for k, v in opts.items():
            t.set_option(k, v)
        t.set_volume(x, VOL_ORIGINAL)
        t.create_projections()
        t.restart_recon()
        t.copy_recon()
    for it in range(3):
        a.SART(0.5, 1); dp_a = a.matrix_2norm(); a.copy_recon()
        dp_b = b.SART_tracked(0.5, 1)
        assert abs(dp_a - dp_b) <= 1e-10 * dp_a
        assert np.array_equal(a.get_volume(VOL_TEMP), b.get_volume(VOL_TEMP))
        b.data_distance_begin()                          # reads TEMP on the second stream while TV runs
        tv_a = a.tv_gd(4, 0.2 * dp_a); dg_a = a.matrix_2norm(); a.copy_recon()
        tv_b, dg_b = b.tv_gd_tracked(4, 0.2 * dp_b)
        dd_b = b.data_distance_end()
        assert abs(tv_a - tv_b) <= 1e-10 * tv_a and abs(dg_a - dg_b) <= 1e-10 * dg_a
        assert np.array_equal(a.get_volume(), b.get_volume())
        assert np.array_equal(a.get_volume(VOL_TEMP), b.get_volume(VOL_TEMP))
        assert dd_b > 0
    # zero TV steps: positivity only, still tracked
    tv_b, dg_b = b.tv_gd_tracked(0, 0.1)
    assert dg_b == 0.0 or dg_b < 1e-3


@pytest.mark.parametrize("ncp", [1, 2, 3, 5])
def test_tile_forward_projection_in_chunk_passes(gpu, ncp):
    """A slab whose partial sums exceed the scratch cap is projected in passes over groups of 64-slice chunks; every
    grouping (lane-group widths 16/32/64 in the reduce kernel, ragged last pass) gives the single-pass sinogram."""
    N, P, Nx = 40, 6, 300                       # 5 chunks
    ang = np.linspace(-70, 65, P)
    x = ellipsoids(Nx, N, seed=4)
    a, b = tomoengine(Nx, N, ang * np.pi / 180), tomoengine(Nx, N, ang * np.pi / 180)
    b.set_option("fp_tile_chunks_per_pass", ncp)
    for t in (a, b):
        t.set_volume(x, VOL_ORIGINAL)
        t.create_projections()
    assert np.array_equal(a.get_projections(), b.get_projections())
    a.SIRT(2); b.SIRT(2)
    assert np.array_equal(a.get_volume(), b.get_volume())
    assert abs(a.data_distance() - b.data_distance()) <= 1e-12 * a.data_distance()


@pytest.mark.parametrize("form", ["strip", "list"])
@pytest.mark.parametrize("Nx,ncp", [(384, 1), (384, 2), (384, 3), (640, 5), (640, 3), (320, 2)])
def test_strip_and_list_forward_projection_in_chunk_passes(gpu, monkeypatch, form, Nx, ncp):
    """The multi-pass path of the strip and the list projector (launch_fp_strip / launch_fp_list with a first chunk c0 > 0 and a
    ragged last pass): the scratch cap splits the chunks only on large slabs (1024^3), so the tables are forced at a small image
    (TOMO_FP_STRIP = 1 / TOMO_FP_LIST = 1) and the chunks per pass set by hand.  The list form works on PAIRS of chunks (an odd
    chunks-per-pass is made even; a slab of an odd number of chunks -- 320 slices -- keeps the strips).  Every grouping gives the
    single-pass sinogram bit for bit."""
    from tomo_tv_amd import _lib
    monkeypatch.setenv("TOMO_FP_STRIP", "1")
    monkeypatch.setenv("TOMO_FP_LIST", "1" if form == "list" else "0")
    N, P = 48, 7
    ang = np.linspace(-70, 65, P) * np.pi / 180
    x = ellipsoids(Nx, N, seed=6)
    a, b = tomoengine(Nx, N, ang), tomoengine(Nx, N, ang)
    want = "list" if (form == "list" and Nx % 128 == 0) else "strip"
    assert _lib.FORM_FP[a.get_option("form_fp")] == want
    b.set_option("fp_tile_chunks_per_pass", ncp)
    for t in (a, b):
        t.set_volume(x, VOL_ORIGINAL)
        t.create_projections()
    pa, pb = a.get_projections(), b.get_projections()
    assert np.isfinite(pa).all() and pa.max() > 0
    assert np.array_equal(pa, pb)
    a.SIRT(2); b.SIRT(2)
    assert np.array_equal(a.get_volume(), b.get_volume())
    assert abs(a.data_distance() - b.data_distance()) <= 1e-12 * a.data_distance()
    # ... and the tile form of the same slab agrees to rounding (another order of the partial sums)
    monkeypatch.setenv("TOMO_FP_STRIP", "0")
    monkeypatch.setenv("TOMO_FP_LIST", "0")
    c = tomoengine(Nx, N, ang)
    assert _lib.FORM_FP[c.get_option("form_fp")] == "tile"
    c.set_volume(x, VOL_ORIGINAL)
    c.create_projections()
    pc = c.get_projections()
    assert rel_l2(pc, pa) < 1e-6


@pytest.mark.parametrize("N,Nx", [(40, 70), (64, 64), (33, 130), (100, 300), (7, 5), (3, 1)])
def test_tv_gradient_kernels_are_bit_identical(gpu, N, Nx):
    """Register march without row rotation (k_tv_march4, default), register march (k_tv_grad_reg), LDS march (k_tv_grad_lds)
    and the TV value they fold into the first pass: the same arithmetic in the same order, so three descent steps leave the
    same bits; the direct-global stencil (4 sqrt + 4 div per voxel) agrees to rounding."""
    x = np.random.default_rng(N + Nx).random((Nx, N, N), dtype=np.float32)
    out = {}
    for opt in (1, "reg", 8, 0):
        t = tomoengine(Nx, N, np.array([10.0, 40.0]) * np.pi / 180)
        if opt == "reg":
            t.set_option("tv_march4", 0)
        else:
            t.set_option("tv_lds", opt)
        t.set_volume(x, VOL_RECON)
        tv = t.tv_gd(3, 0.05)
        out[opt] = (tv, t.get_volume(), t.tv())
    assert np.array_equal(out[1][1], out[8][1]) and out[1][0] == out[8][0] and out[1][2] == out[8][2]
    assert np.array_equal(out[1][1], out["reg"][1]) and out[1][0] == out["reg"][0]
    assert np.abs(out[1][1] - out[0][1]).max() < 1e-6 and abs(out[1][0] - out[0][0]) <= 1e-6 * out[0][0]


@pytest.mark.parametrize("N,Nx", [(40, 70), (64, 256)])
def test_tv_march_segment_length_does_not_change_a_bit(gpu, N, Nx):
    """"tv_yseg" (rows a wave of the TV march walks; chosen by slab size) only changes the work partition."""
    x = np.random.default_rng(N * Nx).random((Nx, N, N), dtype=np.float32)
    out = {}
    for yseg in (0, 32, 8, 4):
        t = tomoengine(Nx, N, np.array([10.0, 40.0]) * np.pi / 180)
        t.set_option("tv_yseg", yseg)
        t.set_volume(x, VOL_RECON)
        t.tv_gd(3, 0.05)
        out[yseg] = t.get_volume()
    assert all(np.array_equal(out[0], v) for v in out.values())


@pytest.mark.parametrize("N,P,Nx", [(48, 7, 70), (33, 5, 130), (16, 1, 3)])
def test_chained_art_equals_row_sequential_art(gpu, N, P, Nx):
    """tomo_art in natural order: fused tile steps (BP_art(prev) + FP(next), k_sart_tile ART) with the recurrence along the
    rays (k_art_chain) in between -- and its unfused form, per-angle FP + recurrence + BP -- against the row-by-row kernel
    (k_art) and the oracle; two sweeps."""
    ang = np.linspace(-75, 72, P) if P > 1 else np.array([33.0])
    A = oracle.parallel_ray(N, ang)
    x = ellipsoids(Nx, N, seed=6)
    ref = oracle.ctvlib(Nx, N, P); ref.load_A(A); ref.original_volume = x.copy(); ref.create_projections()
    ref.row_inner_product()
    vols = {}
    for chain in (1, "rows", 0):
        dev = ctvlib(Nx, N, P); dev.load_A(A)
        if chain == "rows":
            dev.set_option("art_tile", 0)
        else:
            dev.set_option("art_chain", chain)
        dev.set_tilt_series(ref.b)
        dev.row_inner_product()
        dev.ART(0.8); dev.ART(0.8)
        vols[chain] = dev.get_volume()
    ref.ART(0.8); ref.ART(0.8)
    assert rel_l2(vols[1], vols[0]) < 2e-6 and rel_l2(vols["rows"], vols[0]) < 2e-6
    assert rel_l2(vols[1], ref.recon) < 1e-5 and rel_l2(vols[0], ref.recon) < 1e-5 and rel_l2(vols["rows"], ref.recon) < 1e-5
