#!/usr/bin/env python3
"""Free-running ASD-POCS against the oracle, characterised (runs on the GPU box; prints the table of DESIGN.md section 5).

usage: python tests/measure_asd_parity.py [libtomo_hip.so] [--big]      (--big adds the 128 x 31 x 16 shape)

For every shape, for the ART loop of tomofusion/cpu/sim_ASD.py:64-96 (through the ``ctvlib`` facade) and the SART loop of
examples/sim_ASD.py:66-94 (through ``tomoengine``), at eps = 1e-8 and 1e-6, 20 iterations:

  * HIP vs oracle: relative L2 of the final volume, worst dd / tv deviation, last iteration whose VOLUME is within 1e-5;
  * the oracle against ITSELF on a tilt series moved by one float32 ulp per sample, EIGHT seeds: median and max of the same
    figures (one seed is a sample, not a distribution: VERDICT r2);
  * both against the same loop evaluated in binary64 (numpy / scipy, below): which fp32 path ends nearer exact arithmetic.

Writes gpurun_out/asd_parity.json and prints markdown.
"""
import json
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
if args:
    from tomo_tv_amd import _lib
    _lib.LIB_PATH = os.path.abspath(args[0])

import oracle  # noqa: E402
from conftest import rel_l2  # noqa: E402
from tomo_tv_amd.engine import ctvlib, tomoengine  # noqa: E402
from tomo_tv_amd.phantom import ellipsoids  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
ART = dict(beta0=0.5, beta_red=0.985, eps=0.02, alpha=0.2, alpha_red=0.95, r_max=0.95, ng=10)       # cpu/sim_ASD.py:14-31
SART = dict(beta0=0.25, beta_red=0.9985, eps=0.025, alpha=0.2, alpha_red=0.95, r_max=0.95, ng=10)   # gpu/reconstructor.py:158-161
SEEDS = range(8)


def ulp_noise(x, seed):
    rng = np.random.default_rng(seed)
    x = np.asarray(x, np.float32)
    return np.nextafter(x, np.where(rng.random(x.shape) < 0.5, -np.inf, np.inf).astype(np.float32))


class F64:
    """The same loops in binary64 on the float32 matrix (ctvlib.cpp:137-155 ART, the SART of oracle/tomo_oracle.c, :272-276,
    :336-367, :406-462): the exact-arithmetic trajectory both fp32 paths approximate."""

    def __init__(self, A3, Nx, N, P, b, eps):
        r, c, v = A3[0].astype(np.int64), A3[1].astype(np.int64), A3[2].astype(np.float64)
        self.A = sp.csr_matrix((v, (r, c)), shape=(N * P, N * N))
        self.A.sum_duplicates()
        self.A.sort_indices()
        self.Nx, self.N, self.P, self.eps = Nx, N, P, eps
        self.b = np.asarray(b, np.float64).T.copy()                 # (Nrow, Nx)
        self.x = np.zeros((N * N, Nx))
        self.inner = np.asarray(self.A.multiply(self.A).sum(axis=1)).ravel()
        self.rowsum = np.asarray(self.A.sum(axis=1)).ravel()
        self.blocks = [self.A[i * N:(i + 1) * N] for i in range(P)]
        self.den = [np.asarray(B.sum(axis=0)).ravel() for B in self.blocks]

    @property
    def recon(self):
        return self.x.T.reshape(self.Nx, self.N, self.N)

    def copy_recon(self):
        self.temp = self.x.copy()

    def matrix_2norm(self):
        return float(np.sqrt(((self.x - self.temp) ** 2).sum()))

    def ART(self, beta):
        ip, ix, va, x = self.A.indptr, self.A.indices, self.A.data, self.x
        for j in range(self.A.shape[0]):
            if not self.inner[j] > 0:
                continue
            k = slice(ip[j], ip[j + 1])
            a = (self.b[j] - va[k] @ x[ix[k]]) / self.inner[j]
            x[ix[k]] += va[k, None] * a[None, :] * beta
        np.maximum(x, 0, out=x)

    def SART(self, beta):
        N = self.N
        for i, B in enumerate(self.blocks):
            rs = self.rowsum[i * N:(i + 1) * N]
            res = np.where(rs[:, None] > 0, (self.b[i * N:(i + 1) * N] - B @ self.x) / np.where(rs > 0, rs, 1)[:, None], 0.0)
            num, den = B.T @ res, self.den[i]
            upd = np.where(den[:, None] > 0, num / np.where(den > 0, den, 1)[:, None], 0.0)
            self.x = np.maximum(self.x + beta * upd, 0)

    def dd_raw(self):
        return float(np.sqrt(((self.A @ self.x - self.b) ** 2).sum()))

    def _D(self, x):
        return np.sqrt(self.eps + (x - np.roll(x, -1, 0)) ** 2 + (x - np.roll(x, -1, 1)) ** 2 + (x - np.roll(x, -1, 2)) ** 2)

    def tv(self):
        return float(self._D(self.recon).sum())

    def tv_gd(self, ng, dPOCS):
        x = self.recon.copy()
        tv0 = float(self._D(x).sum())
        for _ in range(ng):
            R = 1.0 / self._D(x)
            g = (3 * x - np.roll(x, -1, 0) - np.roll(x, -1, 1) - np.roll(x, -1, 2)) * R
            for ax in range(3):
                g += (x - np.roll(x, 1, ax)) * np.roll(R, 1, ax)
            x = x - dPOCS * g / np.sqrt((g * g).sum())
        self.x = np.maximum(x, 0).reshape(self.Nx, -1).T.copy()
        return tv0


def loop(t, kind, p, norm, Niter=20):
    """cpu/sim_ASD.py:64-96 (kind 'ART') / examples/sim_ASD.py:66-94 (kind 'SART'); returns dd, tv traces and the iterates."""
    beta, dPOCS = p["beta0"], 0.0
    dd, tv, vols = np.zeros(Niter), np.zeros(Niter), []
    is64 = isinstance(t, F64)
    for i in range(Niter):
        t.copy_recon()
        if kind == "ART":
            t.ART(beta)
        else:
            t.SART(beta) if is64 else t.SART(beta, 1)
        beta *= p["beta_red"]
        dp = t.matrix_2norm()
        if i == 0:
            dPOCS = dp * p["alpha"]
        if is64:
            dd[i] = t.dd_raw() / norm
        elif isinstance(t, oracle.ctvlib):
            dd[i] = t.data_distance(normalize=False) / norm
        else:
            dd[i] = t.data_distance() / (1 if kind == "ART" else norm)   # the ctvlib facade divides by the size itself
        t.copy_recon()
        tv[i] = t.tv_gd(p["ng"], dPOCS)
        dg = t.matrix_2norm()
        if dg > dp * p["r_max"] and dd[i] > p["eps"]:
            dPOCS *= p["alpha_red"]
        vols.append(np.array(t.get_volume() if hasattr(t, "get_volume") else t.recon, dtype=np.float32))
    return dd, tv, vols


def last_ok(vols, ref_vols, tol=1e-5):
    n = 0
    for a, b in zip(vols, ref_vols):
        if rel_l2(a, b) > tol:
            break
        n += 1
    return n


def worst(a, b):
    return float(np.max(np.abs(a - b) / np.abs(b)))


shapes = [(16, 5, 2), (32, 9, 4), (64, 16, 8)] + ([(128, 31, 16)] if "--big" in sys.argv else [])
rows = []
for N, P, Nx in shapes:
    ang = np.linspace(-70, 70, P)
    A3 = oracle.parallel_ray(N, ang)
    fixture = os.path.join(GOLD, f"trace_asd_art_N{N}_P{P}_Nx{Nx}.npz")
    if os.path.exists(fixture):
        b_art, b_sart = np.load(fixture)["b"], np.load(os.path.join(GOLD, f"trace_N{N}_P{P}_Nx{Nx}.npz"))["b"]
    else:                                                       # no fixture at this shape: same recipe (tools/gen_golden.py)
        x0 = ellipsoids(Nx, N)
        r = oracle.ctvlib(Nx, N, P)
        r.load_A(A3)
        r.original_volume = x0.copy()
        r.create_projections()
        b_sart = r.b.copy()
        x1 = x0.copy()
        x1[x1 == 0] = 1
        r.original_volume = x1
        r.create_projections()
        r.poisson_noise(100, seed=4321)
        b_art = r.b.copy()
    norm = float(Nx * N * P)
    for kind, p, b in (("ART", ART, b_art), ("SART", SART, b_sart)):
        for eps in (1e-8, 1e-6):
            def make_oracle(bb):
                r = oracle.ctvlib(Nx, N, P)
                r.load_A(A3)
                r.row_inner_product()
                r.set_tilt_series(bb)
                r.tv_eps = eps
                return r
            if kind == "ART":
                d = ctvlib(Nx, N, P)
                d.load_A(A3)
                d.row_inner_product()
            else:
                d = tomoengine(Nx, N, np.deg2rad(ang))
            d.set_tilt_series(b)
            d.tv_eps = eps
            dd_d, tv_d, v_d = loop(d, kind, p, norm)
            dd_o, tv_o, v_o = loop(make_oracle(b), kind, p, norm)
            dd_f, tv_f, v_f = loop(F64(A3, Nx, N, P, b, eps), kind, p, norm)
            selfs = [loop(make_oracle(ulp_noise(b, s)), kind, p, norm) for s in SEEDS]
            s_l2 = [rel_l2(v[-1], v_o[-1]) for _, _, v in selfs]
            s_dd = [worst(a, dd_o) for a, _, _ in selfs]
            s_tv = [worst(a, tv_o) for _, a, _ in selfs]
            s_ok = [last_ok(v, v_o) for _, _, v in selfs]
            rows.append(dict(loop=kind, shape=f"{N}x{P}x{Nx}", eps=eps,
                             hip_l2=rel_l2(v_d[-1], v_o[-1]), hip_dd=worst(dd_d, dd_o), hip_tv=worst(tv_d, tv_o), hip_last_ok=last_ok(v_d, v_o),
                             self_l2_med=float(np.median(s_l2)), self_l2_max=float(np.max(s_l2)), self_dd_max=float(np.max(s_dd)),
                             self_tv_max=float(np.max(s_tv)), self_last_ok_min=int(np.min(s_ok)), self_last_ok_max=int(np.max(s_ok)),
                             hip_vs_f64=rel_l2(v_d[-1], v_f[-1]), oracle_vs_f64=rel_l2(v_o[-1], v_f[-1]),
                             hip_f64_last_ok=last_ok(v_d, v_f), oracle_f64_last_ok=last_ok(v_o, v_f),
                             hip_l2_iter1=rel_l2(v_d[0], v_o[0]), hip_l2_iter5=rel_l2(v_d[4], v_o[4]),
                             self_l2_iter5_max=float(np.max([rel_l2(v[4], v_o[4]) for _, _, v in selfs]))))
            print(rows[-1], file=sys.stderr, flush=True)

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "asd_parity.json"), "w"), indent=1)
print("| loop | N x P x Nx | eps | HIP vs oracle: rel-L2 after 20 it | worst dd / tv | iterations within 1e-5 | oracle vs itself, b +-1 ulp, 8 seeds: rel-L2 median / max | worst dd / tv (max) | iterations within 1e-5 (min..max) | vs binary64: HIP / oracle | iterations within 1e-5 of binary64: HIP / oracle |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for r in rows:
    print(f"| {r['loop']} | {r['shape']} | {r['eps']:g} | {r['hip_l2']:.1e} | {r['hip_dd']:.1e} / {r['hip_tv']:.1e} | {r['hip_last_ok']} | "
          f"{r['self_l2_med']:.1e} / {r['self_l2_max']:.1e} | {r['self_dd_max']:.1e} / {r['self_tv_max']:.1e} | "
          f"{r['self_last_ok_min']}..{r['self_last_ok_max']} | {r['hip_vs_f64']:.1e} / {r['oracle_vs_f64']:.1e} | "
          f"{r['hip_f64_last_ok']} / {r['oracle_f64_last_ok']} |")
