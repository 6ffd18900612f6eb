"""Pin the oracle: bit-exact against the reference's own parallelRay output, and against independent fp64 maths."""
import hashlib
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp

import oracle
from conftest import GOLDEN, rel_l2


@pytest.mark.parametrize("name", ["A_N16_P5.npz", "A_N32_P9.npz", "A_N64_P16.npz", "A_axis_N8.npz", "A_odd_N9.npz"])
def test_parallel_ray_bit_exact(name):
    g = np.load(os.path.join(GOLDEN, name))
    A = oracle.parallel_ray(int(g["N"]), g["angles_deg"])
    assert A.shape == g["A"].shape and np.array_equal(A, g["A"])


def test_parallel_ray_digests():
    dig = json.load(open(os.path.join(GOLDEN, "A_digest.json")))
    A = oracle.parallel_ray(128, np.linspace(-70, 70, 31))
    assert hashlib.sha256(A.tobytes()).hexdigest() == dig["N128_P31_lin70"]["sha256"]


@pytest.mark.parametrize("N,P", [(256, 60), (512, 90), (512, 70), (1024, 120)])
def test_parallel_ray_digests_baseline_geometries(N, P):
    """The oracle's builder at the BASELINE.json geometries against the imported reference's digests (round 3): the oracle
    the full-size GPU parity tests compare with starts from the reference's own matrix, bit for bit."""
    d = json.load(open(os.path.join(GOLDEN, "A_digest.json")))[f"N{N}_P{P}_lin70"]
    A = oracle.parallel_ray(N, np.linspace(-70, 70, P))
    assert A.shape[1] == d["nnz"] and hashlib.sha256(A.tobytes()).hexdigest() == d["sha256"]


def test_geometry_conventions():
    """theta=0 -> ray j is image column j with weight 1; +90 -> ray 0 is the last row."""
    A = oracle.parallel_ray(8, np.array([0.0, 90.0]))
    r, c, v = A[0].astype(int), A[1].astype(int), A[2]
    for j in range(8):
        cols = np.sort(c[r == j])
        assert np.array_equal(cols, np.arange(8) * 8 + j) and np.allclose(v[r == j], 1.0)
    assert np.array_equal(np.sort(c[r == 8]), 7 * 8 + np.arange(8))


@pytest.fixture(scope="module")
def case():
    g = np.load(os.path.join(GOLDEN, "trace_N32_P9_Nx4.npz"))
    A = np.load(os.path.join(GOLDEN, "A_N32_P9.npz"))["A"]
    N, P, Nx = 32, 9, 4
    M = sp.csr_matrix((A[2].astype(np.float64), (A[0].astype(int), A[1].astype(int))), shape=(N * P, N * N))
    t = oracle.ctvlib(Nx, N, P)
    t.load_A(A)
    return g, M, t, (N, P, Nx)


def test_load_A_sorted_rows(case):
    _, M, t, _ = case
    A = t.A
    assert A.nnz == M.nnz
    for r in range(0, A.nrow, 7):
        idx = A.idx[A.ptr[r]:A.ptr[r + 1]]
        assert np.all(np.diff(idx) > 0)


def test_forward_back_lipschitz_vs_fp64(case):
    g, M, t, (N, P, Nx) = case
    x0 = g["x0"]
    t.initialize_original_volume()
    t.original_volume[:] = x0
    t.create_projections()
    b64 = (M @ x0.reshape(Nx, -1).astype(np.float64).T).T
    assert rel_l2(t.b, b64) < 2e-7
    assert np.array_equal(t.b, g["b"])
    atb = (M.T @ b64.T).T.reshape(Nx, N, N)
    assert rel_l2(t.back_projection(t.b), atb) < 3e-7
    L = (M.T @ (M @ np.ones(N * N))).max()
    assert abs(t.lipschits() - L) / L < 1e-6


def test_sirt_sart_tv_vs_fp64(case):
    g, M, t, (N, P, Nx) = case
    b = g["b"].astype(np.float64)
    beta = 1.0 / float(g["lipschitz"])
    x = np.zeros((Nx, N * N))
    for _ in range(5):
        x = np.maximum(0, x + beta * (M.T @ (b.T - M @ x.T)).T)
    assert rel_l2(g["sirt_k5"].reshape(Nx, -1), x) < 1e-6
    v = g["x_sart"].astype(np.float64)
    tv = np.sqrt(1e-8 + (v - np.roll(v, -1, 0)) ** 2 + (v - np.roll(v, -1, 1)) ** 2 + (v - np.roll(v, -1, 2)) ** 2).sum()
    assert abs(tv - float(g["tv_eps1e-08"])) / tv < 1e-6


def test_tv_gradient_is_derivative_of_tv(case):
    """tv_gd's gradient tensor is d(TV)/dx: finite-difference check in fp64 on a tiny volume."""
    rng = np.random.default_rng(3)
    v = rng.uniform(0.2, 1.0, (3, 4, 5)).astype(np.float32)
    eps = 1e-3

    def tv64(u):
        return np.sqrt(eps + (u - np.roll(u, -1, 0)) ** 2 + (u - np.roll(u, -1, 1)) ** 2 + (u - np.roll(u, -1, 2)) ** 2).sum()
    t = oracle.ctvlib(3, 4, 1)
    # gradient via one tv_gd step: x1 = x0 - d * g/||g||  ->  g/||g|| = (x0 - x1)/d  (no clamping: values stay > 0)
    import ctypes
    vol = np.zeros((3, 4, 5), np.float32)
    vol[:] = v
    scratch = np.empty_like(vol)
    oracle.lib().orc_tv_gd(3, 4, 5, vol.ctypes.data_as(ctypes.c_void_p), scratch.ctypes.data_as(ctypes.c_void_p), 1, 1e-3, eps)
    gdir = (v.astype(np.float64) - vol) / 1e-3
    num = np.zeros_like(gdir)
    h = 1e-4
    for idx in np.ndindex(*v.shape):
        u = v.astype(np.float64).copy(); u[idx] += h
        d = v.astype(np.float64).copy(); d[idx] -= h
        num[idx] = (tv64(u) - tv64(d)) / (2 * h)
    num /= np.linalg.norm(num)
    assert rel_l2(gdir, num) < 2e-3


def test_traces_regression(case):
    """The committed trace vectors are what the oracle produces today (guards the restatement against drift)."""
    g, M, t, (N, P, Nx) = case
    t.set_tilt_series(g["b"])
    t.restart_recon()
    beta = 1.0 / t.lipschits()
    for _ in range(5):
        t.SIRT(beta)
    assert np.array_equal(t.recon, g["sirt_k5"])
    t.restart_recon()
    t.SART(0.25, 1)
    assert np.array_equal(t.recon, g["sart_b025"])
    t.recon[:] = g["x_sart"]
    t.tv_eps = 1e-8
    tv0 = t.tv_gd(10, 0.05)
    assert np.array_equal(t.recon, g["tvgd_ng10_eps1e-08"]) and tv0 == float(g["tvgd_tv0_ng10_eps1e-08"])
    t.recon[:] = g["x_sart"]
    t.tv_fgp(10, 0.1)
    assert np.array_equal(t.recon, g["fgp_i10_l0.1"])


def test_fgp_reduces_tv_and_keeps_nonneg(case):
    g, _, t, _ = case
    t.recon[:] = g["x_sart"]
    t.tv_eps = 1e-6
    before = t.tv()
    t.tv_fgp(10, 0.1)
    assert t.tv() < before and t.recon.min() >= 0


def test_config1_trace_is_converging():
    g = np.load(os.path.join(GOLDEN, "trace_config1_sirt50.npz"))
    assert g["recon"].shape == (1, 256, 256)
    assert np.all(np.diff(g["dd"]) < 0) and g["rmse"][-1] < g["rmse"][0]


@pytest.mark.parametrize("N,P,Nx", [(16, 5, 2), (32, 9, 4)])
def test_harness_mirror_equals_the_imported_reference_helpers(N, P, Nx):
    """tests/golden/trace_refharness_*.npz: the reference's OWN harness helpers (cpu/utils/pytvlib.py:171-213, imported by
    tools/gen_golden.py --only-refharness) executed on the oracle class.  The product's mirror of those helpers
    (tomo_tv_amd.cpu_harness: same names and signatures) executed on the same class must give the same bits: what is compared
    is the helper logic -- which engine methods are called, in which order, with which arguments."""
    from tomo_tv_amd import cpu_harness as H
    g = np.load(os.path.join(GOLDEN, f"trace_refharness_N{N}_P{P}_Nx{Nx}.npz"))
    ang = g["angles_deg"]
    for alg in ("SIRT", "ART", "cimminoSIRT"):
        for snr in (0, 100):
            t = oracle.ctvlib(Nx, N, P)
            H.initialize_algorithm(t, alg, N, ang)
            beta0 = 0.5
            if alg == "SIRT":
                beta0 = 1 / t.lipschits()
            if alg == "cimminoSIRT":
                beta0 = N * P / t.lipschits()
            H.create_projections(t, g["x0"].copy(), snr)
            key = f"{alg}_snr{snr}"
            assert np.array_equal(t.b, g[f"b_{key}"]), key
            beta = beta0
            for i in range(8):
                H.run(t, alg, beta)
                if alg != "SIRT":
                    beta *= 0.995
                # (the oracle's fp64 sums are OpenMP reductions: the last bit depends on the team)
                assert abs(t.data_distance() - g[f"dd_{key}"][i]) <= 1e-12 * g[f"dd_{key}"][i], (key, i)
                assert abs(t.rmse() - g[f"rmse_{key}"][i]) <= 1e-12 * g[f"rmse_{key}"][i], (key, i)
            assert np.array_equal(t.recon, g[f"recon_{key}"]), key
    t = oracle.ctvlib(Nx, N, P)
    H.load_exp_tilt_series(t, g["exp_ts"])
    assert np.array_equal(t.b, g["exp_b"])
