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


@pytest.mark.parametrize("N,P,Nx", SHAPES)
def test_asd_pocs_trace(gpu, golden, N, P, Nx):
    """examples/sim_ASD.py:66-94 loop, 20 iterations, defaults; dd/tv/dPOCS traces and the final volume."""
    from tomo_tv_amd.reconstructor import TomoGPU
    A = golden(f"A_N{N}_P{P}.npz")
    g = golden(f"trace_N{N}_P{P}_Nx{Nx}.npz")
    ts = g["b"].reshape(Nx, P, N).transpose(0, 2, 1)             # (Nslice, Nray, Nangles)
    rec = TomoGPU(A["angles_deg"], ts)
    # parallelRay takes deg*pi/180; TomoGPU uses np.deg2rad like the reference -- same to 1 ulp of the angle
    rec.tomo.tv_eps = 1e-8
    dd, tv = rec.asd_pocs(Niter=20)
    assert np.allclose(dd, g["asd_dd"], rtol=2e-5)
    assert np.allclose(tv, g["asd_tv"], rtol=2e-5)
    assert rel_l2(rec.get_recon(), g["asd_final"]) < 5e-5        # 20 sweeps x (P updates + 10 TV steps)


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
