#!/usr/bin/env python3
"""Measured parity of free-running ASD-POCS loops against the oracle (runs on the GPU box; golden fixtures + oracle).

For every fixture shape, for the ART-based loop of tomofusion/cpu/sim_ASD.py (through the ``ctvlib`` facade) and the
SART-based loop of examples/sim_ASD.py (through ``tomoengine``), at eps = 1e-8 and 1e-6: relative L2 of the final volume
after 20 iterations, the worst relative deviation of the dd / tv traces, and -- as the yardstick of conditioning -- how
far the ORACLE itself moves when its tilt series changes by one float32 ulp.  Prints a markdown table (DESIGN.md
section 5) and writes gpurun_out/asd_parity.json.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import oracle  # noqa: E402
from conftest import rel_l2  # noqa: E402
from gen_golden import SIM_ASD, sim_asd_loop  # noqa: E402
from test_gpu_parity import asd_loop, ulp_noise  # noqa: E402
from tomo_tv_amd.engine import ctvlib, tomoengine  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
rows = []
for N, P, Nx in [(16, 5, 2), (32, 9, 4), (64, 16, 8)]:
    A = np.load(os.path.join(GOLD, f"A_N{N}_P{P}.npz"))
    ga = np.load(os.path.join(GOLD, f"trace_asd_art_N{N}_P{P}_Nx{Nx}.npz"))
    gs = np.load(os.path.join(GOLD, f"trace_N{N}_P{P}_Nx{Nx}.npz"))
    for eps in (1e-8, 1e-6):
        # ---- ART loop (cpu/sim_ASD.py) ----
        def art(make, b):
            t = make()
            t.load_A(A["A"])
            t.row_inner_product()
            t.initialize_recon_copy()
            t.initialize_original_volume()
            for s in range(Nx):
                t.set_original_volume(ga["x0"][s], s)
            t.set_tilt_series(b)
            t.tv_eps = eps
            tr = sim_asd_loop(t, 20)
            return tr, (t.get_volume() if hasattr(t, "get_volume") else t.recon.copy())
        tr_d, v_d = art(lambda: ctvlib(Nx, N, P), ga["b"])
        tr_o, v_o = art(lambda: oracle.ctvlib(Nx, N, P), ulp_noise(ga["b"], 5))
        key = f"eps{eps:g}"
        want = ga[f"final_{key}"]
        dev = lambda tr, k: float(np.max(np.abs(tr[k] - ga[f"{k}_{key}"]) / np.abs(ga[f"{k}_{key}"])))  # noqa: E731
        first_bad = lambda tr: int(np.argmax(np.abs(tr["dd"] - ga[f"dd_{key}"]) / ga[f"dd_{key}"] > 1e-5)) or 20  # noqa: E731
        rows.append(dict(loop="ART (cpu/sim_ASD.py)", shape=f"{N}x{P}x{Nx}", eps=eps, hip_l2=rel_l2(v_d, want),
                         hip_first_iter_over_1e5=first_bad(tr_d), oracle_first_iter_over_1e5=first_bad(tr_o),
                         hip_dd=dev(tr_d, "dd"), hip_tv=dev(tr_d, "tv"), oracle_ulp_l2=rel_l2(v_o, want),
                         oracle_ulp_dd=dev(tr_o, "dd"), oracle_ulp_tv=dev(tr_o, "tv"),
                         hip_l2_iter1=None))
        # ---- SART loop (examples/sim_ASD.py) ----
        ang = np.asarray(A["angles_deg"]) * np.pi / 180
        d = tomoengine(Nx, N, ang)
        d.set_tilt_series(gs["b"])
        d.tv_eps = eps
        dd_d, tv_d = asd_loop(d, 20, Nx * N * P, lambda t: t.data_distance())
        outs = []
        for b in (gs["b"], ulp_noise(gs["b"], 5)):
            r = oracle.ctvlib(Nx, N, P)
            r.load_A(A["A"])
            r.set_tilt_series(b)
            r.tv_eps = eps
            dd_r, tv_r = asd_loop(r, 20, Nx * N * P, lambda t: t.data_distance(normalize=False))
            outs.append((dd_r, tv_r, r.recon.copy()))
        (dd0, tv0, v0), (dd1, tv1, v1) = outs
        m = lambda a, b: float(np.max(np.abs(a - b) / np.abs(b)))  # noqa: E731
        fb = lambda a, b: int(np.argmax(np.abs(a - b) / b > 1e-5)) or 20  # noqa: E731
        rows.append(dict(loop="SART (examples/sim_ASD.py)", shape=f"{N}x{P}x{Nx}", eps=eps, hip_l2=rel_l2(d.get_volume(), v0),
                         hip_first_iter_over_1e5=fb(dd_d, dd0), oracle_first_iter_over_1e5=fb(dd1, dd0),
                         hip_dd=m(dd_d, dd0), hip_tv=m(tv_d, tv0), oracle_ulp_l2=rel_l2(v1, v0), oracle_ulp_dd=m(dd1, dd0),
                         oracle_ulp_tv=m(tv1, tv0), hip_l2_iter1=None))

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "asd_parity.json"), "w"), indent=1)
print("| loop | N x P x Nx | eps | HIP vs oracle: rel-L2 after 20 it | dd | tv | first it. with dd off by > 1e-5 | oracle vs itself (+1 ulp on b): rel-L2 | dd | tv | first it. |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for r in rows:
    print(f"| {r['loop']} | {r['shape']} | {r['eps']:g} | {r['hip_l2']:.1e} | {r['hip_dd']:.1e} | {r['hip_tv']:.1e} | "
          f"{r['hip_first_iter_over_1e5'] + 1 if r['hip_first_iter_over_1e5'] < 20 else 'none'} | "
          f"{r['oracle_ulp_l2']:.1e} | {r['oracle_ulp_dd']:.1e} | {r['oracle_ulp_tv']:.1e} | "
          f"{r['oracle_first_iter_over_1e5'] + 1 if r['oracle_first_iter_over_1e5'] < 20 else 'none'} |")
