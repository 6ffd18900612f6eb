#!/usr/bin/env python3
"""Generate tests/golden/*.npz.  Runs ONLY in the build container (needs /root/reference).

Two kinds of vectors are written, and every file says which is which in its ``provenance`` field:

* ``A_N{N}_P{P}.npz`` -- output of the REFERENCE ITSELF: ``parallelRay`` imported from
  ``/root/reference/tomofusion/cpu/utils/pytvlib.py`` (its unused top-level imports ``skimage``/``h5py`` are
  stubbed in ``sys.modules``).  These pin the oracle's and the product's system-matrix builders.
  ``A_digest.json`` holds (nnz, sha256, fp64 sum) for shapes too large to commit (config 1: 256x256, 50 tilts).
* ``sigma_*.npz`` -- ``create_weighted_summation_matrix`` imported from the reference's
  ``tomofusion/chemistry/utils/fusion_helper.py`` (config 5's summation weights incl. fp16 rounding).
* ``trace_*.npz`` -- outputs of the repo's restatement (``oracle/``) driven in the reference's harness order
  (``tomofusion/cpu/sim_tomo.py:35-61``, ``tomofusion/cpu/sim_ASD.py:47-96``, ``demo.ipynb`` FISTA cell).  The
  reference has no tests and its C++ cannot be built here (Eigen absent), so these are regression vectors of
  the restatement, not reference outputs.
"""
import hashlib
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


def import_reference():
    for m in ("skimage", "skimage.io", "h5py"):
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.modules["skimage"].io = sys.modules["skimage.io"]
    sys.path.insert(0, os.path.join(REF, "tomofusion", "cpu", "utils"))
    import pytvlib as ref_pytvlib  # noqa
    sys.path.insert(0, os.path.join(REF, "tomofusion", "chemistry", "utils"))
    import fusion_helper as ref_fusion  # noqa
    return ref_pytvlib, ref_fusion


def digest(A):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(A, dtype=np.float32).tobytes())
    return {"nnz": int(A.shape[1]), "sha256": h.hexdigest(), "sum_vals_f64": float(A[2].astype(np.float64).sum())}


def phantom(nx, n, seed=1234, k=20):
    """Seeded ellipsoid phantom; same definition as tomo_tv_amd.phantom.ellipsoids."""
    from tomo_tv_amd.phantom import ellipsoids
    return ellipsoids(nx, n, seed=seed, k=k)


SIM_ASD = dict(alg="ART", beta0=0.5, beta_red=0.985, eps=0.02, alpha=0.2, alpha_red=0.95, r_max=0.95, ng=10, SNR=100)
"""The parameter block of tomofusion/cpu/sim_ASD.py:14-31 (its own defaults)."""


def sim_asd_loop(tomo, Niter, p=SIM_ASD, on_iter=None):
    """The main loop of tomofusion/cpu/sim_ASD.py:64-96 on any object with the ``ctvlib`` method table (the oracle
    here, the GPU facade in tests/test_gpu_cpu_harness.py).  Returns the per-iteration traces."""
    beta = p["beta0"]
    tr = {k: np.zeros(Niter) for k in ("dd", "tv", "rmse", "dp", "dg", "dPOCS")}
    dPOCS = 0.0
    for i in range(Niter):
        tomo.copy_recon()
        tomo.ART(beta)                                   # run(tomo, 'ART', beta)
        beta *= p["beta_red"]
        if i == 0:
            dPOCS = tomo.matrix_2norm() * p["alpha"]
            dp = dPOCS / p["alpha"]
        else:
            dp = tomo.matrix_2norm()
        tr["dd"][i] = tomo.data_distance()
        tr["rmse"][i] = tomo.rmse()
        tomo.copy_recon()
        tr["tv"][i] = tomo.tv()
        tomo.tv_gd(p["ng"], dPOCS)
        dg = tomo.matrix_2norm()
        if dg > dp * p["r_max"] and tr["dd"][i] > p["eps"]:
            dPOCS *= p["alpha_red"]
        tr["dp"][i], tr["dg"][i], tr["dPOCS"][i] = dp, dg, dPOCS
        if on_iter is not None:
            on_iter(i, tomo)
    return tr


def asd_art_traces(shapes, Niter=20):
    """ASD-POCS in the CPU reference's own form: alg = 'ART', the defaults of cpu/sim_ASD.py:14-31, SNR = 100 (background
    lifted to 1, Poisson noise -- drawn from a seeded numpy generator, the reference's engine is unseedable: Q13).
    Every line of the loop is arithmetic of ctvlib.cpp restated in oracle/: ART :137-155, true-copy matrix_2norm
    :254-269 (Q1), data_distance :272-276, rmse :296-306, tv_3D :336-367, tv_gd_3D :406-462."""
    import oracle
    for N, P, Nx in shapes:
        A = np.load(os.path.join(GOLD, f"A_N{N}_P{P}.npz"))["A"]
        x0 = phantom(Nx, N).copy()
        x0[x0 == 0] = 1                                   # cpu/utils/pytvlib.py:194-195 (SNR != 0)
        out = {"x0": x0, "params": json.dumps(SIM_ASD)}
        for eps in (1e-8, 1e-6):
            t = oracle.ctvlib(Nx, N, P)
            t.load_A(A)
            t.row_inner_product()
            t.initialize_recon_copy()
            t.initialize_original_volume()
            for s in range(Nx):
                t.set_original_volume(x0[s], s)
            t.create_projections()
            t.poisson_noise(SIM_ASD["SNR"], seed=4321)
            out["b"] = t.b.copy()
            t.tv_eps = eps
            out[f"tv0_eps{eps:g}"] = np.float64(t.original_tv())
            snaps = {}
            tr = sim_asd_loop(t, Niter, on_iter=lambda i, tt: snaps.__setitem__(i, tt.recon.copy()) if i in (0, 4) else None)
            for k, v in tr.items():
                out[f"{k}_eps{eps:g}"] = v
            out[f"final_eps{eps:g}"] = t.recon.copy()
            out[f"iter1_eps{eps:g}"], out[f"iter5_eps{eps:g}"] = snaps[0], snaps[4]
        out["provenance"] = ("oracle/ restatement driven by the loop of tomofusion/cpu/sim_ASD.py:64-96 with its own "
                             "defaults (alg ART); noise from numpy default_rng(4321) (not reference output)")
        np.savez_compressed(os.path.join(GOLD, f"trace_asd_art_N{N}_P{P}_Nx{Nx}.npz"), **out)
        # Cimmino branch of ctvlib::SIRT (ctvlib.cpp:212-216, 245-251) on the same data: 10 iterations at 1/L
        t = oracle.ctvlib(Nx, N, P)
        t.load_A(A)
        t.cimminos_method()
        t.set_tilt_series(out["b"])
        L = t.lipschits()
        for _ in range(10):
            t.SIRT(t.Nrow / L)
        np.savez_compressed(os.path.join(GOLD, f"trace_cimmino_N{N}_P{P}_Nx{Nx}.npz"), b=out["b"], lipschitz=np.float32(L),
                            recon10=t.recon.copy(),
                            provenance="oracle/ restatement of the Cimmino branch (ctvlib.cpp:198-199,212-216,245-251)")


def refharness_traces(ref):
    """Round 3: the harness helpers of tomofusion/cpu/utils/pytvlib.py -- initialize_algorithm (:178-189), create_projections
    (:191-206), run (:171-176), load_exp_tilt_series (:208-213) -- IMPORTED and executed here, acting on the oracle class (which
    carries the ctvlib method table).  Only the loop around them is typed (cpu/sim_tomo.py:35-61).  What the fixtures pin: which
    engine methods each helper calls, in which order and with which arguments (background lift before the projection, noise after
    it, the transpose of load_exp_tilt_series, row_inner_product / cimminos_method by algorithm name)."""
    import io
    import contextlib
    import oracle
    for N, P, Nx in [(32, 9, 4), (16, 5, 2)]:
        ang = np.linspace(-70, 70, P)
        out = {"angles_deg": ang}
        x0 = phantom(Nx, N)
        out["x0"] = x0
        for alg in ("SIRT", "ART", "cimminoSIRT"):
            for snr in (0, 100):
                t = oracle.ctvlib(Nx, N, P)
                with contextlib.redirect_stdout(io.StringIO()):              # the reference prints a banner
                    ref.initialize_algorithm(t, alg, N, ang)
                beta0 = 0.5
                if alg == "SIRT":
                    beta0 = 1 / t.lipschits()
                if alg == "cimminoSIRT":
                    beta0 = N * P / t.lipschits()
                vol = x0.copy()
                ref.create_projections(t, vol, snr)                          # lifts the background in place when snr != 0
                key = f"{alg}_snr{snr}"
                out[f"b_{key}"] = t.b.copy()
                beta, dd, rm = beta0, np.zeros(8), np.zeros(8)
                for i in range(8):
                    ref.run(t, alg, beta)
                    if alg != "SIRT":
                        beta *= 0.995
                    dd[i], rm[i] = t.data_distance(), t.rmse()
                out[f"dd_{key}"], out[f"rmse_{key}"], out[f"recon_{key}"] = dd, rm, t.recon.copy()
        # load_exp_tilt_series: (Nslice, Nray, Nproj) -> the engine's (Nslice, Nray*Nproj) layout
        ts = np.random.default_rng(5).random((Nx, N, P)).astype(np.float32)
        t = oracle.ctvlib(Nx, N, P)
        ref.load_exp_tilt_series(t, ts)
        out["exp_ts"], out["exp_b"] = ts, t.b.copy()
        out["provenance"] = ("helpers initialize_algorithm / create_projections / run / load_exp_tilt_series IMPORTED from the reference's "
                             "tomofusion/cpu/utils/pytvlib.py and executed on the oracle class; loop typed from cpu/sim_tomo.py:35-61; "
                             "noise from numpy default_rng(4321) (quirk Q13)")
        np.savez_compressed(os.path.join(GOLD, f"trace_refharness_N{N}_P{P}_Nx{Nx}.npz"), **out)
        print("wrote", f"trace_refharness_N{N}_P{P}_Nx{Nx}.npz")


def method_tables():
    """Method NAMES of the reference's five pybind11 classes (interface data for tests/test_abi_cpu.py)."""
    import re

    def defs(path):
        src = open(os.path.join(REF, "tomofusion", path)).read()
        return sorted(set(re.findall(r'\.def\(\s*"([A-Za-z_0-9]+)"', src[src.index("PYBIND11_MODULE"):])))
    tables = {"tomoengine": defs("gpu/utils/tomoengine.cpp"), "multigpuengine": defs("gpu/utils/multigpuengine.cpp"),
              "ctvlib": defs("cpu/utils/ctvlib.cpp"), "multimodal": defs("chemistry/utils/multimodal.cpp"),
              "multigpufusion": defs("chemistry/utils/multigpufusion.cpp")}
    json.dump({"provenance": "method NAMES of the reference's pybind11 tables (the .def(\"name\", ...) strings of tomoengine.cpp:487-534, "
                             "multigpuengine.cpp:385-421, ctvlib.cpp:486-520, multimodal.cpp:520-564, multigpufusion.cpp:463-474), "
                             "extracted by tools/gen_golden.py", "tables": tables},
              open(os.path.join(GOLD, "method_tables.json"), "w"), indent=1, sort_keys=True)


def main():
    os.makedirs(GOLD, exist_ok=True)
    if "--only-method-tables" in sys.argv:
        method_tables()
        return
    if "--only-asd-art" in sys.argv:       # the traces added in round 2; everything else stays byte-identical
        asd_art_traces([(16, 5, 2), (32, 9, 4), (64, 16, 8)])
        return
    ref, ref_fusion = import_reference()
    import oracle

    if "--only-refharness" in sys.argv:
        refharness_traces(ref)
        return
    if "--only-baseline-digests" in sys.argv:
        # round 3: the imported reference's matrix at the BASELINE geometries (config 3: 512 x 90, config 5: 512 x 70,
        # config 4: 1024 x 120) -- digests only; the other entries of A_digest.json stay as they are
        path = os.path.join(GOLD, "A_digest.json")
        digests = json.load(open(path))
        for N, P in [(256, 60), (512, 90), (512, 70), (1024, 120)]:
            A = ref.parallelRay(N, np.linspace(-70, 70, P))
            digests[f"N{N}_P{P}_lin70"] = digest(A)
            print(N, P, digests[f"N{N}_P{P}_lin70"], flush=True)
            del A
        with open(path, "w") as f:
            json.dump(digests, f, indent=1, sort_keys=True)
        return

    digests = {}
    shapes = [(16, 5, 2), (32, 9, 4), (64, 16, 8)]
    extra_angle_sets = {
        "axis": (8, np.array([0.0, 45.0, 90.0, -90.0, 30.0, 60.0, -45.0])),
        "odd": (9, np.array([-63.5, 0.0, 12.25, 90.0])),
    }
    for N, P, Nx in shapes:
        ang = np.linspace(-70, 70, P)
        A = ref.parallelRay(N, ang)
        np.savez_compressed(os.path.join(GOLD, f"A_N{N}_P{P}.npz"), A=A, angles_deg=ang, N=N,
                            provenance="reference parallelRay (tomofusion/cpu/utils/pytvlib.py:8-121), imported")
        digests[f"N{N}_P{P}_lin70"] = digest(A)
    for name, (N, ang) in extra_angle_sets.items():
        A = ref.parallelRay(N, ang)
        np.savez_compressed(os.path.join(GOLD, f"A_{name}_N{N}.npz"), A=A, angles_deg=ang, N=N,
                            provenance="reference parallelRay (tomofusion/cpu/utils/pytvlib.py:8-121), imported")
        digests[f"{name}_N{N}"] = digest(A)
    # digests only (too large to commit)
    for N, P in [(128, 31), (256, 50)]:
        ang = np.linspace(-70, 70, P)
        A = ref.parallelRay(N, ang)
        digests[f"N{N}_P{P}_lin70"] = digest(A)
        if (N, P) == (256, 50):
            A256 = A
    with open(os.path.join(GOLD, "A_digest.json"), "w") as f:
        json.dump(digests, f, indent=1, sort_keys=True)

    # summation matrix for config 5 (fusion_helper.py:5-32), methods 1 and 3
    for method, zs in [(1, [31, 8]), (3, [31, 8]), (3, [22, 38, 8])]:
        S = ref_fusion.create_weighted_summation_matrix(6, 6, len(zs), zs, 1.6, method)
        S = S.tocsr()
        np.savez_compressed(os.path.join(GOLD, f"sigma_m{method}_nz{len(zs)}.npz"), indptr=S.indptr,
                            indices=S.indices, data=S.data.astype(np.float32), shape=np.array(S.shape), zs=np.array(zs),
                            provenance="reference create_weighted_summation_matrix (fusion_helper.py:5-32), imported")

    # ---- restatement traces ------------------------------------------------------------------
    for N, P, Nx in shapes:
        ang = np.linspace(-70, 70, P)
        A = np.load(os.path.join(GOLD, f"A_N{N}_P{P}.npz"))["A"]
        x0 = phantom(Nx, N)
        out = {"x0": x0, "angles_deg": ang}
        t = oracle.ctvlib(Nx, N, P)
        t.load_A(A)
        t.initialize_original_volume()
        for s in range(Nx):
            t.set_original_volume(x0[s], s)
        t.create_projections()
        out["b"] = t.b.copy()
        out["ATb"] = t.back_projection(t.b)
        out["lipschitz"] = np.float32(t.lipschits())
        # SIRT (cpu/sim_tomo.py:35-61 with alg='SIRT')
        beta = 1.0 / t.lipschits()
        dd, rm = [], []
        for it in range(50):
            t.SIRT(beta)
            dd.append(t.data_distance())
            rm.append(t.rmse())
            if it + 1 in (1, 5, 50):
                out[f"sirt_k{it + 1}"] = t.recon.copy()
        out["sirt_dd"], out["sirt_rmse"] = np.array(dd), np.array(rm)
        # ART: one sweep
        t.restart_recon()
        t.row_inner_product()
        t.ART(0.5)
        out["art_1"] = t.recon.copy()
        # SART: two sweeps, beta 0.25 then 1
        t.restart_recon()
        t.SART(0.25, 1)
        out["sart_b025"] = t.recon.copy()
        t.SART(1.0, 1)
        out["sart_b1"] = t.recon.copy()
        xs = t.recon.copy()
        # normalised SIRT, 5 iterations from zero
        t.restart_recon()
        t.SIRT_norm(5)
        out["sirtnorm_5"] = t.recon.copy()
        # TV value / TV-GD / FGP on the SART iterate
        for eps in (1e-8, 1e-6):
            t.tv_eps = eps
            t.recon[:] = xs
            out[f"tv_eps{eps:g}"] = np.float64(t.tv())
            for ng in (1, 10):
                t.recon[:] = xs
                out[f"tvgd_tv0_ng{ng}_eps{eps:g}"] = np.float64(t.tv_gd(ng, 0.05))
                out[f"tvgd_ng{ng}_eps{eps:g}"] = t.recon.copy()
        for it_, lam in [(1, 0.1), (10, 0.1), (10, 15.0)]:
            t.recon[:] = xs
            out[f"fgp_tv0_i{it_}_l{lam:g}"] = np.float64(t.tv_fgp(it_, lam))
            out[f"fgp_i{it_}_l{lam:g}"] = t.recon.copy()
        out["x_sart"] = xs
        # ASD-POCS trace, 20 iterations (cpu/sim_ASD.py:47-96 / examples/sim_ASD.py:66-94, alg SART)
        t.tv_eps = 1e-8
        t.restart_recon()
        t.initialize_recon_copy()
        beta, beta_red, eps_dd, r_max, alpha, alpha_red, ng = 0.25, 0.9985, 0.025, 0.95, 0.2, 0.95, 10
        tr = {k: [] for k in ("dd", "tv", "dPOCS", "beta", "dp", "dg")}
        for i in range(20):
            t.copy_recon()
            t.SART(beta, 1)
            beta *= beta_red
            if i == 0:
                dPOCS = t.matrix_2norm() * alpha
                dp = dPOCS / alpha
            else:
                dp = t.matrix_2norm()
            dd_ = t.data_distance()
            t.copy_recon()
            tv_ = t.tv_gd(ng, dPOCS)
            dg = t.matrix_2norm()
            if dg > dp * r_max and dd_ > eps_dd:
                dPOCS *= alpha_red
            for k_, v_ in (("dd", dd_), ("tv", tv_), ("dPOCS", dPOCS), ("beta", beta), ("dp", dp), ("dg", dg)):
                tr[k_].append(v_)
        for k_, v_ in tr.items():
            out[f"asd_{k_}"] = np.array(v_)
        out["asd_final"] = t.recon.copy()
        # FISTA (textbook form, quirk Q6): y-step normalised SIRT, FGP prox, momentum; 10 iterations
        t.restart_recon()
        t.initialize_fista()
        t.restart_recon()
        t0 = 1.0
        cost = []
        for k_ in range(10):
            t.SIRT_norm(1, target="yk")
            t.recon[:] = t.yk
            t.tv_fgp(5, 0.01)
            t.yk[:] = t.recon
            tk = 0.5 * (1 + np.sqrt(1 + 4 * t0 ** 2))
            t.fista_momentum((t0 - 1) / tk)
            t0 = tk
            cost.append(0.5 * t.data_distance(normalize=False) ** 2 + 0.01 * t.tv())
        out["fista_final"] = t.recon.copy()
        out["fista_cost"] = np.array(cost)
        out["provenance"] = "oracle/ restatement driven in the reference's harness order (not reference output)"
        np.savez_compressed(os.path.join(GOLD, f"trace_N{N}_P{P}_Nx{Nx}.npz"), **out)

    # config 1: 2-D 256x256 Shepp-Logan, 50 tilts, SIRT x50 (final image + traces only)
    from tomo_tv_amd.phantom import shepp_logan
    N, P = 256, 50
    ang = np.linspace(-70, 70, P)
    t = oracle.ctvlib(1, N, P)
    t.load_A(A256)
    img = shepp_logan(N)[None]
    t.initialize_original_volume()
    t.set_original_volume(img[0], 0)
    t.create_projections()
    beta = 1.0 / t.lipschits()
    dd, rm = [], []
    for it in range(50):
        t.SIRT(beta)
        dd.append(t.data_distance())
        rm.append(t.rmse())
    np.savez_compressed(os.path.join(GOLD, "trace_config1_sirt50.npz"), recon=t.recon.astype(np.float32),
                        dd=np.array(dd), rmse=np.array(rm), lipschitz=np.float32(t.lipschits()),
                        provenance="oracle/ restatement, config 1 (256x256 Shepp-Logan, 50 tilts, SIRT x50)")
    asd_art_traces(shapes)
    method_tables()
    print("golden written to", GOLD)


if __name__ == "__main__":
    main()
