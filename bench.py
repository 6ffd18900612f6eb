#!/usr/bin/env python3
"""Headline benchmark: SART+TV (one ASD-POCS outer iteration) at 512^3 x 90 tilts.

    python bench.py --gpus N --steps K --warmup W [--scaling strong|weak] [--n 1024 --nslice 1024 --nproj 120]

A step = one ASD-POCS outer iteration (examples/sim_ASD.py:66-94): copy_recon, one SART sweep over all tilts, step
norm, data distance (a full forward projection), copy_recon, 10 TV gradient-descent steps, step norm (the engine forms
the two norms and snapshot copies inside the sweep's last back-projection and the last descent step; the iteration's
scalars are read back together).  Synthetic phantom + tilt series are resident in HBM before the timed region.

N > 1: one rank per GPU over RCCL.  Launched by ``torch.distributed.run`` the ranks come from the environment; launched
plainly (``python bench.py --gpus 4``) this process starts the N ranks itself as fresh child processes before anything
touches the GPU (the role of multigpuengine.cpp:140-193, which starts its own workers).  ``--scaling strong`` (default,
BASELINE's metric: ONE 512^3 x 90 volume on 1/2/4/8 GPUs) shards the --nslice slices of the volume over the ranks;
``--scaling weak`` gives every rank its own --nslice slices.  The only cross-rank traffic is one scalar all-reduce per
iteration + one per TV step, and the TV halo planes.  Rank 0 prints ONE JSON line.

At N = 1 the same process then times, as sub-objects of that line (``secondary``): config 2 (256^3 x 60 SART), config 3
(512^3 x 90 FISTA, roofline of the fused FGP iteration), the 8-GPU shard of config 4 (128 x 1024^2 x 120 ASD-POCS),
ASD-POCS in the CPU reference's ART form, a normalised-SIRT iteration with the tile projectors' three roofs -- and the
CPU baselines (``cpu_baseline`` + configs 1 and 2).  ``--quick`` skips the secondary part.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0        # HBM3E 8.0 TB/s spec
VALU_PEAK_TFLOPS = 157.3     # fp32 vector peak
LDS_PEAK_GBS = 256 * 256 * 2.4   # 256 CUs x 256 B/clk (ds_read_b128) x 2.4 GHz = 157 TB/s
RMW_CEILING_GBS = 5860.0     # measured: tools/micro/copy_patterns.hip (in-place read + write of the slab by 16x16x64 tiles, non-temporal
                             # loads and stores; 5340 with plain ones) -- informative


def asd_pocs_step(t, st):
    """One outer iteration, state dict st carries beta / dPOCS (defaults of gpu/reconstructor.py:158-161)."""
    if hasattr(t, "SART_tracked"):
        from tomo_tv_amd._lib import S_DD, S_DIFF2
        # engine (= TomoGPU.asd_pocs): the step norms and snapshot copies ride on the last back-projection / last
        # descent pass; the residual of the SART result runs on the second stream under the TV steps; one read-back
        # (the read-back of an iteration's scalars is collected after the NEXT sweep has been enqueued: they steer nothing before
        # that iteration's TV steps, and the device does not wait for the host between iterations; asd_pocs_flush ends the run)
        if st["i"] == 0:
            t.copy_recon()
            dp0 = t.SART_tracked(st["beta"], 1)
            st["dPOCS"] = dp0 * 0.2
        else:
            dp0 = None
            t.SART_tracked(st["beta"], 1, defer=True)       # its step norm is read with the other scalars of the iteration
            asd_pocs_flush(t, st)
        st["beta"] *= 0.9985
        t.data_distance_begin()
        st["pending"] = (dp0, t.tv_gd_tracked(10, st["dPOCS"], extra=(S_DD,) if dp0 is not None else (S_DD, S_DIFF2), defer=True))
        st["i"] += 1
        return st.get("dd"), st.get("tv")
    else:                                      # oracle (cpu_baseline): the same work as separate calls
        t.copy_recon()
        t.SART(st["beta"], 1)
        st["beta"] *= 0.9985
        if st["i"] == 0:
            st["dPOCS"] = t.matrix_2norm() * 0.2
            dp = st["dPOCS"] / 0.2
        else:
            dp = t.matrix_2norm()
        t.copy_recon()
        dd = t.data_distance(normalize=False) / st["norm"]
        tv = t.tv_gd(10, st["dPOCS"])
        dg = t.matrix_2norm()
    if dg > dp * 0.95 and dd > 0.025:
        st["dPOCS"] *= 0.95
    st["i"] += 1
    st["dd"], st["tv"] = dd, tv
    return dd, tv


def asd_pocs_flush(t, st):
    """Collect the scalars of the last enqueued iteration (engine form) and apply the step-length rule they feed."""
    if st.get("pending") is None:
        return st.get("dd"), st.get("tv")
    dp, get = st.pop("pending")
    if dp is None:
        tv, dg, dd2, dp2 = get()
        dp = dp2 ** 0.5
    else:
        tv, dg, dd2 = get()
    dd = dd2 ** 0.5 / st["norm"]
    if dg > dp * 0.95 and dd > 0.025:
        st["dPOCS"] *= 0.95
    st["dd"], st["tv"] = dd, tv
    return dd, tv


# ---- CPU baselines (the oracle, timed build, on the host cores the box grants) ---------------------------------------
def _oracle_setup(ns, n, nproj, shepp=False):
    import oracle
    from tomo_tv_amd.phantom import ellipsoids, shepp_logan, tilt_angles
    ang = tilt_angles(nproj)
    ref = oracle.ctvlib(ns, n, nproj)
    ref.load_A(oracle.parallel_ray(n, ang))
    ref.original_volume = shepp_logan(n)[None].astype(np.float32) if shepp else ellipsoids(ns, n)
    ref.create_projections()
    ref.initialize_recon_copy()
    ref.tv_eps = 1e-6
    return ref


def _timed(fn, budget_s, max_iters, min_iters=1):
    t0 = time.perf_counter()
    it = 0
    while True:
        fn()
        it += 1
        el = time.perf_counter() - t0
        if (el > budget_s and it >= min_iters) or it >= max_iters:
            return it, el


def cpu_baselines(n, nproj, budget_s=14.0):
    """Headline sample + configs 1 and 2.  Returns (cpu_baseline, {config: ...})."""
    import oracle
    oracle.select_build("timed")
    if "OMP_NUM_THREADS" not in os.environ:
        oracle.set_num_threads(oracle.usable_cpus())   # the baseline uses every CPU the host grants
    threads = oracle.num_threads()
    build = f"{oracle.BUILD_FLAGS['timed']}; OpenMP over slices like ctvlib.cpp:207; oracle/tomo_oracle.c"
    ns = max(8, 2 * threads)
    ref = _oracle_setup(ns, n, nproj)
    st = {"beta": 0.25, "i": 0, "dPOCS": 0.0, "norm": float(ns * n * nproj)}
    iters, el = _timed(lambda: asd_pocs_step(ref, st), budget_s, 5)
    head = {"value": ns * n * n * iters / el / 1e9, "unit": "Gvoxel-updates/s", "cores": threads, "kind": "port", "build": build,
            "sample": f"{iters} ASD-POCS iterations (SART sweep + 10 TV-GD steps) on a {ns}x{n}x{n} slab of the workload, "
                      f"{nproj} tilts",
            "iters_per_s_full_volume_equiv": iters / el * ns / n}
    out = {}
    # config 1: 2-D 256x256 Shepp-Logan, 50 tilts, SIRT (the reference's CPU-runnable case; Landweber at 1/L)
    ref = _oracle_setup(1, 256, 50, shepp=True)
    beta = 1.0 / ref.lipschits()
    iters, el = _timed(lambda: ref.SIRT(beta), 2.0, 50)
    out["config1_sirt_256sq_x50tilts"] = {"iters_per_s": iters / el, "ms_per_iter": el / iters * 1e3, "cores": 1,
                                          "sample": f"{iters} Landweber-SIRT iterations, 1 slice (OpenMP is over slices)"}
    # config 2: 256^3, 60 tilts: SIRT and SART+TV on a slab sample
    ns2 = max(8, 2 * threads)
    ref = _oracle_setup(ns2, 256, 60)
    beta = 1.0 / ref.lipschits()
    iters, el = _timed(lambda: ref.SIRT(beta), 3.0, 20)
    out["config2_sirt_256cube_x60tilts"] = {"iters_per_s_full_volume_equiv": iters / el * ns2 / 256, "cores": threads,
                                            "gvoxel_updates_per_s": ns2 * 256 * 256 * iters / el / 1e9,
                                            "sample": f"{iters} Landweber-SIRT iterations on a {ns2}x256x256 slab"}
    st = {"beta": 0.25, "i": 0, "dPOCS": 0.0, "norm": float(ns2 * 256 * 60)}
    iters, el = _timed(lambda: asd_pocs_step(ref, st), 4.0, 5)
    out["config2_sart_tv_256cube_x60tilts"] = {"iters_per_s_full_volume_equiv": iters / el * ns2 / 256, "cores": threads,
                                               "gvoxel_updates_per_s": ns2 * 256 * 256 * iters / el / 1e9,
                                               "sample": f"{iters} SART+TV iterations on a {ns2}x256x256 slab"}
    # config 3: 512^3, 90 tilts, FISTA (normalised SIRT step + 10 FGP-TV iterations at lambda 0.1 + Nesterov step + cost)
    ns3 = max(8, 2 * threads)
    ref = _oracle_setup(ns3, 512, 90)
    ref.initialize_fista()
    fs = {"t0": 1.0}

    def fista_iter():
        ref.SIRT_norm(1, target="yk")
        ref.recon, ref.yk = ref.yk, ref.recon               # the oracle's tv_fgp acts on .recon
        ref.tv_fgp(10, 0.1)
        ref.recon, ref.yk = ref.yk, ref.recon
        tk = 0.5 * (1 + np.sqrt(1 + 4 * fs["t0"] ** 2))
        ref.fista_momentum((fs["t0"] - 1) / tk)
        fs["t0"] = tk
        return 0.5 * ref.data_distance(normalize=False) ** 2 + 0.1 * ref.tv()
    iters, el = _timed(fista_iter, 6.0, 5, min_iters=3)      # at least three iterations
    out["config3_fista_512cube_x90tilts"] = {"iters_per_s_full_volume_equiv": iters / el * ns3 / 512, "cores": threads,
                                             "gvoxel_updates_per_s": ns3 * 512 * 512 * iters / el / 1e9,
                                             "sample": f"{iters} FISTA iterations (SIRT step + 10 FGP-TV + momentum + cost) on a {ns3}x512x512 slab"}
    # config 4: 1024^3, 120 tilts, ASD-POCS -- a sample of the 128-slice slab one of 8 GPUs owns ("a 1/8 slab")
    ns4 = max(8, threads)
    ref = _oracle_setup(ns4, 1024, 120)
    st = {"beta": 0.25, "i": 0, "dPOCS": 0.0, "norm": float(ns4 * 1024 * 120)}
    iters, el = _timed(lambda: asd_pocs_step(ref, st), 5.0, 3)
    out["config4_asd_pocs_1024sq_x120tilts"] = {"ms_per_step_128_slice_slab_equiv": el / iters * 1e3 * 128 / ns4, "cores": threads,
                                                "iters_per_s_full_volume_equiv": iters / el * ns4 / 1024,
                                                "gvoxel_updates_per_s": ns4 * 1024.0 * 1024 * iters / el / 1e9,
                                                "sample": f"{iters} ASD-POCS iterations on a {ns4}x1024x1024 sample of the 128-slice slab "
                                                          "(1/8 of config 4), scaled by slices (OpenMP is over slices)"}
    oracle.select_build("parity")
    return head, out


# ---- per-kernel roofline from the engine's HIP-event log ----------------------------------------------------------------
class KernelLog:
    """The engine's HIP-event log of the named kernels.  A sub-slab group (tomoengine(..., sub_slabs=K)) logs per sub-slab
    engine; the launches of all of them are put on one time base (same device) and merged."""

    def __init__(self, t, ids, stride=None):
        """stride: name -> N brackets every N-th launch of that kernel only (an event pair costs ~3 us of command-processor
        time: with every launch of an ASD-POCS step bracketed the step is 0.7 ms = 3 % slower than unobserved)."""
        from tomo_tv_amd import _lib
        self._lib, self.ids = _lib, ids
        self.stride = dict(stride or {})
        self.kids = list(getattr(t.be, "kids", [t.be]))
        for kid in self.kids:
            for name, kid_id in ids.items():
                _lib.check(kid.L.tomo_profile_enable(kid.h, kid_id, max(1, int(self.stride.get(name, 1)))))

    def read(self):
        """name -> (launches, summed launch durations in ms, ms during which at least one launch was executing)."""
        import ctypes
        out = {}
        L = self.kids[0].L
        for name, kid_id in self.ids.items():
            if len(self.kids) == 1:
                launches, total_ms, busy_ms = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
                self._lib.check(L.tomo_profile_read2(self.kids[0].h, kid_id, ctypes.byref(launches), ctypes.byref(total_ms),
                                                     ctypes.byref(busy_ms)))
                out[name] = (int(launches.value), float(total_ms.value), float(busy_ms.value))
            else:
                iv = []
                for kid in self.kids:
                    n = ctypes.c_int(0)
                    self._lib.check(L.tomo_profile_intervals(kid.h, kid_id, self.kids[0].h, None, None, 0, ctypes.byref(n)))
                    t0, t1 = np.zeros(max(n.value, 1)), np.zeros(max(n.value, 1))
                    self._lib.check(L.tomo_profile_intervals(kid.h, kid_id, self.kids[0].h, t0.ctypes.data_as(ctypes.c_void_p),
                                                             t1.ctypes.data_as(ctypes.c_void_p), n.value, ctypes.byref(n)))
                    iv += list(zip(t0[:n.value], t1[:n.value]))
                iv.sort()
                busy, cur_a, cur_b = 0.0, 0.0, -1.0
                for a, b in iv:
                    if cur_b < cur_a or a > cur_b:
                        if cur_b >= cur_a:
                            busy += cur_b - cur_a
                        cur_a, cur_b = a, b
                    else:
                        cur_b = max(cur_b, b)
                if cur_b >= cur_a:
                    busy += cur_b - cur_a
                out[name] = (len(iv), float(sum(b - a for a, b in iv)), float(busy))
            for kid in self.kids:
                self._lib.check(L.tomo_profile_enable(kid.h, kid_id, 0))
        return out


def roof(name, cnt, tot_ms, alg_bytes, flops=None, lds_bytes=None, busy_ms=None):
    """``alg_bytes``: algorithmic bytes of ONE launch.  Launches of one kernel may overlap (the SART sweep runs two sub-slabs
    on two streams): ``achieved`` is then all launches' bytes over the time at least one of them was executing
    (``busy_ms``); ``avg_ms`` stays the mean duration of a launch (what rocprofv3 prints), ``achieved_per_launch`` the
    bytes of a launch over that duration (a launch that shares the chip with its twin)."""
    avg_ms = tot_ms / cnt if cnt else 0.0
    busy = busy_ms if busy_ms else tot_ms
    ach = alg_bytes * cnt / (busy * 1e-3) / 1e9 if busy > 0 else 0.0
    r = {"kernel": name, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
         "traffic": None, "launches": cnt, "avg_ms": avg_ms, "total_ms": tot_ms, "busy_ms": busy,
         "launches_in_flight": tot_ms / busy if busy > 0 else 0.0, "algorithmic_bytes_per_launch": alg_bytes,
         "achieved_per_launch": alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0}
    eff_ms = busy / cnt if cnt else 0.0
    if flops is not None and eff_ms > 0:      # kernels that are not HBM-bound: the other two roofs (SURVEY.md 8d caveat)
        r["valu_tflops"] = flops / (eff_ms * 1e-3) / 1e12
        r["valu_frac"] = r["valu_tflops"] / VALU_PEAK_TFLOPS
    if lds_bytes is not None and eff_ms > 0:
        r["lds_gbs"] = lds_bytes / (eff_ms * 1e-3) / 1e9
        r["lds_frac"] = r["lds_gbs"] / LDS_PEAK_GBS
    return r


def fracs_above_one(doc, path=""):
    """Every key named ``frac`` / ``*_frac`` / ``frac_*`` of the record whose value exceeds 1 (none may)."""
    bad = []
    if isinstance(doc, dict):
        for k, v in doc.items():
            here = f"{path}.{k}" if path else str(k)
            if isinstance(v, (int, float)) and not isinstance(v, bool) and (k == "frac" or k.endswith("_frac") or k.startswith("frac_")) and v > 1.0:
                bad.append({"key": here, "value": v})
            else:
                bad += fracs_above_one(v, here)
    elif isinstance(doc, list):
        for i, v in enumerate(doc):
            bad += fracs_above_one(v, f"{path}[{i}]")
    return bad


def resident_sweep(t):
    """True when a SART sweep of this engine runs as ONE launch of the volume-resident kernel (k_sart_resident) -- the engine's own
    selection (tomo_engine.hip: select_forms) answers, under the options in force (not "sart_resident_active", which
    ignores "sart_fused")."""
    try:
        return t.get_option("form_sart") == 2
    except Exception:  # noqa: BLE001 -- the numpy slab double of the launcher test has no such option
        return False


def resident_roof(cnt, tot_ms, busy_ms, nslice, n, nproj, tracked):
    """Roofline record of k_sart_resident: one launch = one whole sweep of the slab, the volume resident in registers.

    The kernel sits on NEITHER roof: a step (one angle of one 64-slice chunk) is two latency-bound loops and two hand-offs of ray
    sums between the workgroups -- ``bound`` says "exchange latency" and ``frac`` is the LARGER of its two roof fractions, both of
    them honest and small:
      hbm   the bytes a launch MUST move -- slab in + out once (snapshot in + out too when tracked), the measured rows, one 16-byte
            cell per pixel, angle and chunk -- over the launch time, against 8 TB/s.  Strictly algorithmic (16V + 4S, SURVEY 8d
            without the tables) is given beside it.
      valu  11 flops per voxel and angle (7 in the back projection: mul, fma, mul, fma, max; 2 FMAs forward) against the fp32 peak.
    ``vs_streamed_form`` is what round 5 mistakenly reported as ``frac``: the bytes the STREAMED form would move for the same work
    (P x (8V + 12 Nx N)) over this launch's time, relative to the HBM peak -- a speed-up over another algorithm form, not a roofline
    fraction.  The phase timeline behind "exchange latency" is profiles/r06_resident_phases.txt (tools/resident_phases.sh)."""
    V = float(nslice) * n * n
    S = float(nslice) * n * nproj
    chunks = (nslice + 63) // 64
    avg_ms = tot_ms / cnt if cnt else 0.0
    sec = avg_ms * 1e-3
    slab = (16.0 if tracked else 8.0) * V + 4.0 * S
    cells = chunks * nproj * 16.0 * n * n
    must = slab + cells
    flops = 11.0 * V * nproj
    hbm_gbs = must / sec / 1e9 if sec > 0 else 0.0
    valu_tf = flops / sec / 1e12 if sec > 0 else 0.0
    hbm_frac, valu_frac = hbm_gbs / HBM_PEAK_GBS, valu_tf / VALU_PEAK_TFLOPS
    streamed = nproj * (8.0 * V + 12.0 * nslice * n)
    by_valu = valu_frac >= hbm_frac
    r = {"kernel": "k_sart_resident", "bound": "exchange latency",
         "achieved": valu_tf if by_valu else hbm_gbs, "peak": VALU_PEAK_TFLOPS if by_valu else HBM_PEAK_GBS,
         "unit": "TFLOP/s" if by_valu else "GB/s", "frac": max(hbm_frac, valu_frac),
         "frac_is": "the larger of the kernel's two roof fractions (%s); it is on neither roof: 'bound' names what binds" % ("fp32 VALU" if by_valu else "HBM on the bytes it must move"),
         "traffic": None, "launches": cnt, "avg_ms": avg_ms, "total_ms": tot_ms, "busy_ms": busy_ms if busy_ms else tot_ms,
         "launches_in_flight": 1.0,
         "us_per_angle_and_chunk": avg_ms * 1e3 / nproj / chunks if avg_ms > 0 else 0.0,
         "hbm": {"bytes_the_launch_must_move": must, "of_which_cells": cells, "achieved_GBs": hbm_gbs, "peak_GBs": HBM_PEAK_GBS, "frac": hbm_frac,
                 "strictly_algorithmic_bytes": slab, "frac_on_strictly_algorithmic_bytes": slab / sec / 1e9 / HBM_PEAK_GBS if sec > 0 else 0.0,
                 "what": "slab in + out once (+ snapshot in + out when tracked) + measured rows + one 16-byte cell per pixel, angle and "
                         "64-slice chunk; the exchange's granules are traffic, not in here"},
         "valu": {"flops_per_launch": flops, "achieved_TFLOPs": valu_tf, "peak_TFLOPs": VALU_PEAK_TFLOPS, "frac": valu_frac},
         "vs_streamed_form": streamed / sec / 1e9 / HBM_PEAK_GBS if sec > 0 else 0.0,
         "vs_streamed_form_is": "bytes the streamed form (k_sart_tile, SURVEY 8d: P x (8V + 12 Nx N)) would move for this work, over this launch's "
                                "time, relative to the HBM peak: a speed-up over another algorithm form, NOT a roofline fraction",
         "phase_profile": "profiles/r06_resident_phases.txt (tools/resident_phases.sh: RS_PROF build of the kernel)"}
    return r


def facade_overhead(world):
    """Host cost of the single-process multi-GPU facade (tomo_tv_amd/inprocess.py): a method call on ``InProcessMultiGPU`` is
    forwarded to ``world`` persistent worker threads through a job queue each and one result queue.  Measured here on the host
    alone (no GPU work): the wall time of an EMPTY forwarded call.  A loop written against the engine's methods pays it per call
    (~7 per ASD-POCS iteration: copy_recon, SART_tracked, data_distance_begin, tv_gd_tracked, its deferred read, two state reads);
    TomoGPU's driver loops run ON the rank threads (reconstructor._on_rank_threads) and pay it once per driver call."""
    from tomo_tv_amd.inprocess import InProcWorld
    w = InProcWorld(max(1, int(world)))
    try:
        for _ in range(50):
            w.run(lambda r: None)
        t0 = time.perf_counter()
        n = 400
        for _ in range(n):
            w.run(lambda r: None)
        us = (time.perf_counter() - t0) / n * 1e6
    finally:
        w.close()
    calls = 7
    return {"world": int(world), "empty_forwarded_call_us": us, "calls_per_asd_pocs_iteration_of_a_host_written_loop": calls,
            "host_overhead_ms_per_iteration_of_a_host_written_loop": us * calls * 1e-3,
            "crossings_per_TomoGPU_driver_call": 1,
            "what": "InProcWorld.run(lambda r: None): queue hand-off to one host thread per device and back; host only"}


def engine_facts(t):
    """What creating this engine cost: device bytes of the tables it built (every kernel family's, eagerly) and the wall clock of
    the creation (tomo_get_option "table_kib" / "create_ms"; VERDICT r5 'what is missing' 6)."""
    try:
        return {"table_MiB": t.get_option("table_kib") / 1024.0, "create_ms": t.get_option("create_ms")}
    except Exception:  # noqa: BLE001 -- the numpy slab double of the launcher test
        return None


def engine_forms(t):
    """Which kernel family the engine runs its all-angle forward projection, its all-angle back projection and its SART sweep as
    (tomo_get_option "form_fp" / "form_bp" / "form_sart": the engine's own selection function answers, nothing is restated here)."""
    from tomo_tv_amd import _lib
    try:
        return {"fp": _lib.FORM_FP[t.get_option("form_fp")], "bp": _lib.FORM_BP[t.get_option("form_bp")], "sart": _lib.FORM_SART[t.get_option("form_sart")]}
    except Exception:  # noqa: BLE001 -- the numpy slab double of the launcher test
        return None


def sart_chains(t):
    """Launch chains a SART sweep of this engine's slab runs as -- asked of the engine (tomo_sart_chain_count: the rule lives in
    tomo_engine.hip: chain_count, under the options in force), not restated here.  A sub-slab group answers per sub-slab engine."""
    import ctypes
    kid = list(getattr(t.be, "kids", [t.be]))[0]
    L = getattr(kid, "L", None)
    if L is None or not hasattr(L, "tomo_sart_chain_count"):      # the numpy slab double of the launcher test
        return 1
    n = ctypes.c_int(0)
    from tomo_tv_amd import _lib
    _lib.check(L.tomo_sart_chain_count(kid.h, ctypes.byref(n)))
    return int(n.value)


def attach_traffic(roofs, shape):
    """HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc, separate runs, FETCH_SIZE x2), null if the
    profile is absent or was taken at another shape.  Measured outside this run by construction (the counters need the
    profiler); the file names the command."""
    for fn in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        path = os.path.join(ROOT, "profiles", fn)
        try:
            doc = json.load(open(path))
        except (OSError, ValueError):
            continue
        if tuple(doc.get("shape", (512, 512, 90))) != tuple(shape):
            continue
        def same_kernel(label, prof_name):
            """bench label (k_name<a,b>: leading template arguments, * = any, TVM_NORM / TVM_UPDATE = 1 / 2) vs a profiler name"""
            def split(nm):
                nm = nm.split("::")[-1]
                base, _, args = nm.partition("<")
                return base.strip(), [a.strip() for a in args.rstrip(">").split(",")] if args else []
            lb, la = split(label)
            pb, pa = split(prof_name)
            la = [{"TVM_NORM": "1", "TVM_UPDATE": "2"}.get(a, a) for a in la]
            return lb == pb and len(la) <= len(pa) and all(a == "*" or a == b for a, b in zip(la, pa))
        for r in roofs.values():
            hit = [v for k, v in doc.get("kernels", {}).items() if same_kernel(r["kernel"], k)]
            if hit and r.get("traffic") is None:
                r["traffic"] = hit[0]["hbm_bytes_per_launch"]
                r["traffic_source"] = f"profiles/{fn} (rocprofv3 --pmc, separate passes, FETCH_SIZE x2)"
                raw = hit[0].get("fetch_kib_raw")
                if r["kernel"] == "k_sart_resident" and raw is not None:
                    # The guide's x2 is calibrated for 16-byte-per-lane streaming reads only.  This kernel reads by scalar loads (cells),
                    # 8-byte-per-lane polls (granules), 4-byte-per-lane loads (the chunk) and line touches; measured on those widths
                    # (tools/micro/fetch_calib.hip, profiles/r06_fetch_calibration.md): every vector read is tallied at half its bytes
                    # whatever its width, scalar loads exactly, writes exactly -- so x2 applies to all but scalar-load misses.
                    wr = hit[0].get("write_kib", 0.0) * 1024.0
                    r["traffic_fetch_raw_plus_write"] = raw * 1024.0 + wr
                    r["traffic_fetch_x2_plus_write"] = 2.0 * raw * 1024.0 + wr
                    must = r["hbm"]["bytes_the_launch_must_move"]
                    r["traffic_over_must_move"] = {"fetch_x2": r["traffic_fetch_x2_plus_write"] / must, "fetch_raw": r["traffic_fetch_raw_plus_write"] / must}
                    r["traffic_note"] = ("calibrated (profiles/r06_fetch_calibration.md): FETCH_SIZE tallies every vector read (4, 8, 16 B per lane, line "
                                         "touches) at half its bytes and scalar loads in full, WRITE_SIZE is exact; the x2 figure is right unless cell "
                                         "bytes reach the fabric through scalar-load misses (then at most one cell table less: the raw figure is a floor)")
        return


# ---- secondary configs (N = 1, same process, after the headline) -----------------------------------------------------------
def _engine(nx, n, nproj, noisy=False):
    import ctypes
    from tomo_tv_amd._lib import VOL_ORIGINAL
    from tomo_tv_amd.engine import tomoengine
    from tomo_tv_amd.phantom import ellipsoids, tilt_angles
    t = tomoengine(nx, n, np.deg2rad(tilt_angles(nproj)), device=0)
    vol = ellipsoids(nx, n)
    if noisy:
        vol[vol == 0] = 1
    t.be.c("set_volume", VOL_ORIGINAL, vol.ctypes.data_as(ctypes.c_void_p))
    del vol
    t.create_projections()
    if noisy:
        t.poisson_noise(100)
    t.restart_recon()
    return t


def _shard_step(nloc, n, nproj, steps=10):
    """ASD-POCS step of a slab of ``nloc`` slices through ``multigpuengine`` on a world-1 RCCL group (every collective call of the
    N-GPU path issued): this file's own ``--force-dist`` path in a fresh child process (torch must initialise the device before
    anything else in a process does; this one has long been running kernels through the C ABI)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.abspath(__file__), "--force-dist", "--quick", "--nslice", str(nloc), "--nray", str(n),
           "--nproj", str(nproj), "--steps", str(steps), "--warmup", "2"]
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if p.returncode != 0 or not line:
            return {"error": f"child exited {p.returncode}: {p.stderr[-400:]}"}
        d = json.loads(line[-1])
        return {"ms_per_step": d["ms_per_step"], "iters_per_s": d["iters_per_s"], "gvoxel_updates_per_s": d["value"],
                "ms_per_step_every_voxel_stored": d.get("ms_per_step_every_voxel_stored"),
                "sart_chains": d["config"].get("sart_chains_per_engine"),
                "form": "bench.py --force-dist in a child process: multigpuengine on a world-1 RCCL group, every all-reduce / ring "
                        "exchange of the N-GPU path issued"}
    except Exception as e:  # noqa: BLE001 -- a secondary figure must not take the headline down
        return {"error": f"{type(e).__name__}: {e}"}


def _sharded_fgp(nloc, n):
    """The slab-sharded FGP-TV prox per inner iteration, two iterations per pass and exchange (k_fgp_fused2 on slabs, round 6) against
    one: tools/bench_fgp_sharded.py in a child process on a world-1 RCCL group, every exchange issued."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_fgp_sharded.py"), "--json", "--nslice", str(nloc), "--n", str(n)]
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if p.returncode != 0 or not line:
            return {"error": f"child exited {p.returncode}: {p.stderr[-400:]}"}
        return json.loads(line[-1])
    except Exception as e:  # noqa: BLE001 -- a secondary figure must not take the headline down
        return {"error": f"{type(e).__name__}: {e}"}


def _time_steps(t, fn, steps, warmup=1):
    for _ in range(warmup):
        fn()
    t.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    t.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def secondary_configs(nnz_per_pixel_angle=1.22):
    from tomo_tv_amd import pytvlib
    from tomo_tv_amd._lib import K_BP_TILE, K_FGP_GRAD, K_FGP_OBJ, K_FP_REDUCE, K_FP_TILE, K_SART_FUSED, K_SART_RESIDENT, VOL_YK
    out = {}
    # ---- config 2: 256^3, 60 tilts, SART (beta 1, sequential) + data_distance per iteration
    t = _engine(256, 256, 60)
    facts = {"256x256x256_x60": engine_facts(t)}
    t.initialize_SART("sequential")
    res2 = resident_sweep(t)
    log = KernelLog(t, {"k_sart_resident": K_SART_RESIDENT} if res2 else {"k_sart_tile<true>": K_SART_FUSED})
    ms = _time_steps(t, lambda: (t.SART(1.0, 1), t.data_distance()), 5)
    V = 256.0 ** 3
    if res2:
        cnt, tot, busy = log.read()["k_sart_resident"]
        roof2 = resident_roof(cnt, tot, busy, 256, 256, 60, False)
    else:
        cnt, tot, busy = log.read()["k_sart_tile<true>"]
        roof2 = roof("k_sart_tile<true>", cnt, tot, (8 * V + 12 * 256 * 256) / sart_chains(t), busy_ms=busy)
    out["config2_sart_256cube_x60tilts"] = {"ms_per_step": ms, "iters_per_s": 1e3 / ms, "gvoxel_updates_per_s": V / ms / 1e6, "roofline": roof2}
    del t
    # ---- config 3: 512^3, 90 tilts: FISTA (lambda 0.1, 10 FGP iterations, cost), then SIRT with the tile projectors' roofs
    t = _engine(512, 512, 90)
    facts["512x512x512_x90"] = engine_facts(t)
    V, S, nx, n, P = 512.0 ** 3, 512.0 * 512 * 90, 512, 512, 90
    pytvlib.initialize_algorithm(t, "fista")
    st = {"t0": 1.0}

    def fista_iter():
        pytvlib.run(t, "fista")
        t.tv_fgp(10, 0.1, vol=VOL_YK)
        tk = 0.5 * (1 + np.sqrt(1 + 4 * st["t0"] ** 2))
        t.fista_momentum((st["t0"] - 1) / tk)
        st["t0"] = tk
        cost = 0.5 * t.data_distance() ** 2 + 0.1 * t.tv()
        t.fista_project_yk()            # what TomoGPU.fista does: A yk for the next step from this A r and the last (linearity)
        return cost
    log = KernelLog(t, {"k_fgp_fused": K_FGP_GRAD, "k_fgp_last": K_FGP_OBJ})
    ms = _time_steps(t, fista_iter, 5)
    rd = log.read()
    (cnt, tot, busy), last = rd["k_fgp_fused"], rd["k_fgp_last"]
    t.set_option("fp_reuse", 0)
    ms_noreuse = _time_steps(t, fista_iter, 5)
    t.set_option("fp_reuse", 1)
    # tv_fgp(10): 9 FGP iterations + D of the last one.  With "fgp_pair" (round 4) that is 4 launches of k_fgp_fused2 (two iterations
    # each: A and P in, P out ONCE per pair) + 1 launch of its FINAL form (the ninth iteration and D in one pass: A and P in, D out),
    # logged under the GRAD and OBJ slots; without it 9 launches of k_fgp_fused + k_fgp_obj.  The roofline prices a pair launch at the
    # bytes it has to move: 28 V whether it advances P by one iteration or by two
    pair = bool(t.get_option("fgp_pair"))
    fgp = roof("k_fgp_fused2<false> (two iterations per launch)" if pair else "k_fgp_fused", cnt, tot, 28 * V, busy_ms=busy)
    fgp["fgp_iterations_per_step"] = 9
    fgp["last_pass"] = dict(zip(("launches", "total_ms", "busy_ms"), last), kernel="k_fgp_fused2<true> (iteration + D)" if pair else "k_fgp_obj",
                            algorithmic_bytes_per_launch=20 * V)
    fgp["ms_per_fgp_iteration"] = (tot + last[1]) / max(last[0], 1) / 9          # all passes of a tv_fgp call over its 9 iterations
    out["config3_fista_512cube_x90tilts"] = {"ms_per_step": ms, "ms_per_step_every_projection_recomputed": ms_noreuse,
                                             "iters_per_s": 1e3 / ms, "gvoxel_updates_per_s": V / ms / 1e6, "roofline": fgp}
    t.remove_momentum()
    t.restart_recon()
    log = KernelLog(t, {"k_fp_tile": K_FP_TILE, "k_fp_tile_reduce": K_FP_REDUCE, "k_bp_tile": K_BP_TILE})
    ms = _time_steps(t, lambda: (t.SIRT(1), t.data_distance()), 5)
    pr = log.read()
    # the step's first forward projection is the one the previous data_distance already made ("fp_reuse", bit-identical); the
    # same loop with every projection recomputed, for the record:
    t.set_option("fp_reuse", 0)
    ms_noreuse = _time_steps(t, lambda: (t.SIRT(1), t.data_distance()), 5)
    t.set_option("fp_reuse", 1)
    nnz = nnz_per_pixel_angle * n * n * P
    avg = lambda k: pr[k][1] / max(pr[k][0], 1)  # noqa: E731
    fp_ms = avg("k_fp_tile") + avg("k_fp_tile_reduce")
    # which form projected: sheared strips (k_fp_strip; large slabs, round 4) or image tiles (k_fp_tile) -- both log under the same slot
    strip = bool(t.get_option("fp_strip_ready") and t.get_option("fp_strip"))
    fp_kernel = ("k_fp_list" if (t.get_option("fp_list_ready") and t.get_option("fp_list") and nx % 128 == 0) else "k_fp_strip") if strip else "k_fp_tile"
    bp_list = bool(t.get_option("bp_list_ready") and t.get_option("bp_list") and nx % 128 == 0)
    bp_kernel = "k_bp_list" if bp_list else "k_bp_tile"
    # the other form, for the record (same engine, one option)
    t.set_option("fp_tile", 1) if strip else None
    ms_tile_form = _time_steps(t, lambda: (t.SIRT(1), t.data_distance()), 5) if strip else None
    t.set_option("fp_strip", 1) if strip else None
    out["config3_sirt_512cube_x90tilts"] = {
        "ms_per_step": ms, "iters_per_s": 1e3 / ms, "ms_per_step_every_projection_recomputed": ms_noreuse,
        "forward_projector": fp_kernel, "ms_per_step_with_the_tile_forward_projector": ms_tile_form,
        # all-angle FP = projection kernel + k_fp_tile_reduce (4V + 4S algorithmic); one entry = one FMA and one 4-byte LDS read per slice
        "roofline_fp_all": dict(roof(fp_kernel + "+k_fp_tile_reduce", 1, fp_ms, 4 * V + 4 * S, flops=2 * nnz * nx, lds_bytes=4 * nnz * nx),
                                **{fp_kernel + "_avg_ms": avg("k_fp_tile"), "k_fp_tile_reduce_avg_ms": avg("k_fp_tile_reduce")}),
        # all-angle BP (8V + 4S).  k_bp_list (round 4; logs under the k_bp_tile slot): one FMA and one LDS row read per NONZERO
        # weight and slice; k_bp_tile: two FMAs and two row reads per pixel, angle and slice
        "roofline_bp_all": roof(bp_kernel, pr["k_bp_tile"][0], pr["k_bp_tile"][1], 8 * V + 4 * S,
                                flops=(2 * nnz if bp_list else 4.0 * n * n * P) * nx,
                                lds_bytes=(4 * nnz if bp_list else 8.0 * n * n * P) * nx, busy_ms=pr["k_bp_tile"][2]),
        "back_projector": bp_kernel}
    del t, log          # (the kernel log holds the engine's handle: without this the 512^3 engine stays alive under the next configs)
    import gc
    gc.collect()
    # ---- ASD-POCS in the CPU reference's form (cpu/sim_ASD.py:64-96: ART sweep + tv + 10 TV-GD steps + 3 norms) at 512^3 x 90
    from tomo_tv_amd.engine import ctvlib
    from tomo_tv_amd.phantom import ellipsoids, tilt_angles
    c = ctvlib(512, 512, 90)
    c.load_A(pytvlib.parallelRay(512, tilt_angles(90)))
    c.set_volume(ellipsoids(512, 512), 2)
    c.create_projections()
    c.tv_eps = 1e-6
    sa = {"beta": 0.5, "dPOCS": None}

    def asd_art():
        c.copy_recon()
        c.ART(sa["beta"])
        sa["beta"] *= 0.985
        dp = c.matrix_2norm()
        if sa["dPOCS"] is None:
            sa["dPOCS"] = dp * 0.2
        dd = c.data_distance()
        c.copy_recon()
        c.tv()
        c.tv_gd(10, sa["dPOCS"])
        dg = c.matrix_2norm()
        if dg > dp * 0.95 and dd > 0.02:
            sa["dPOCS"] *= 0.95
    ms = _time_steps(c, asd_art, 3)
    out["asd_pocs_art_512cube_x90tilts"] = {"ms_per_step": ms, "iters_per_s": 1e3 / ms, "gvoxel_updates_per_s": V / ms / 1e6,
                                            "form": "cpu/sim_ASD.py:64-96 through the ctvlib facade (chained ART sweep, separate norms)"}
    del c
    # ---- one GPU's shard of config 4: 128 x 1024^2, 120 tilts, ASD-POCS
    t = _engine(128, 1024, 120)
    facts["128x1024x1024_x120"] = engine_facts(t)
    t.initialize_SART("sequential")
    st4 = {"beta": 0.25, "i": 0, "dPOCS": 0.0, "norm": float(t.Nslice_ * t.Nrow)}
    asd_pocs_step(t, st4)
    log = KernelLog(t, {"k_sart_tile<true>": K_SART_FUSED})
    ms = _time_steps(t, lambda: asd_pocs_step(t, st4), 3, warmup=0)
    cnt, tot, busy = log.read()["k_sart_tile<true>"]
    V4 = 128.0 * 1024 * 1024
    ns = sart_chains(t)
    out["config4_shard_asd_pocs_128x1024sq_x120tilts"] = {
        "ms_per_step": ms, "iters_per_s": 1e3 / ms, "gvoxel_updates_per_s": V4 / ms / 1e6,
        "roofline": roof("k_sart_tile<true>", cnt, tot, (8 * V4 + 12 * 128 * 1024) / ns, busy_ms=busy)}
    del t
    # ---- config 4 WHOLE on one GPU: 1024^3, 120 tilts -- the N = 1 anchor of the config-4 scaling curve (multigpuengine.cpp:163-193)
    t = _engine(1024, 1024, 120)
    facts["1024x1024x1024_x120"] = engine_facts(t)
    t.initialize_SART("sequential")
    st4 = {"beta": 0.25, "i": 0, "dPOCS": 0.0, "norm": float(t.Nslice_ * t.Nrow)}
    asd_pocs_step(t, st4)
    ms = _time_steps(t, lambda: asd_pocs_step(t, st4), 2, warmup=0)
    asd_pocs_flush(t, st4)
    out["config4_full_1024cube_x120tilts_1gpu"] = {"ms_per_step": ms, "iters_per_s": 1e3 / ms, "gvoxel_updates_per_s": 1024.0 ** 3 / ms / 1e6,
                                                   "sart_chains": sart_chains(t)}
    del t
    # ---- the headline on a realistic tilt series (cpu/utils/pytvlib.py:191-206 with SNR = 100: background lifted to 1, Poisson
    # noise): no exact zeros anywhere, so k_sart_tile's skipping of unchanged stores has nothing to skip
    t = _engine(512, 512, 90, noisy=True)
    t.initialize_SART("sequential")
    stn = {"beta": 0.25, "i": 0, "dPOCS": 0.0, "norm": float(t.Nslice_ * t.Nrow)}
    asd_pocs_step(t, stn)
    ms = _time_steps(t, lambda: asd_pocs_step(t, stn), 3, warmup=0)
    asd_pocs_flush(t, stn)
    out["asd_pocs_noisy_512cube_x90tilts"] = {"ms_per_step": ms, "iters_per_s": 1e3 / ms, "gvoxel_updates_per_s": 512.0 ** 3 / ms / 1e6,
                                              "data": "background 1 + Poisson noise at 100 counts per sample (cpu/sim_ASD.py:31 SNR = 100)"}
    del t
    # ---- the slab ONE rank of an 8-GPU strong-scaling run of the headline owns (64 x 512^2, 90 tilts), through the slab-sharded
    # engine with its real collective calls on a world-1 RCCL group: the compute side of the 8-GPU point of the scaling curve
    out["shard_64x512sq_x90tilts"] = _shard_step(64, 512, 90)
    out["sharded_fgp_128x512sq"] = _sharded_fgp(128, 512)
    # ---- config 5 on ONE GPU: ChemicalTomo data-fusion iteration, ADF + 2 spectral channels, 512^3, 70 tilts
    # (chemistry/reconstructor.py:182-225: sirt_data_fusion(lambdaHAADF 10, lambdaCHEM 0.05, iterSIRT 5) + tv_fgp_4D(5, 1e-4))
    from tomo_tv_amd.chemistry import create_weighted_summation_weights, multimodal
    ang5 = np.deg2rad(tilt_angles(70))
    mm = multimodal(512, 512, 2, ang5, ang5)
    mm.set_gamma(1.6)
    mm.set_weights(create_weighted_summation_weights([30, 8], 1.6, 3))
    mm.set_volume(np.stack([ellipsoids(512, 512, seed=5 + e) * np.float32(0.5 + 0.3 * e) for e in range(2)]))
    mm._mm_model()
    mm.he.be.c("forward_projection", mm.MODEL, 0)
    bh = mm.he.get_projections()
    mm.set_haadf_tilt_series(bh / bh.max())
    for e in range(2):
        mm.ce.be.c("forward_projection", int(mm._x[e]), int(mm._b[e]))
    mm.restart_recon()
    mm.set_measureChem(True)
    mm.set_measureHaadf(True)
    for _ in range(3):
        mm.poisson_ml(0.05)
    mm.rescale_tomograms(10)
    mm.rescale_projections()
    ms = _time_steps(mm.ce, lambda: (mm.sirt_data_fusion(10, 0.05, 5), mm.tv_fgp_4D(5, 1e-4)), 3)
    out["config5_chemicaltomo_fusion_512cube_x70tilts_1gpu"] = {
        "ms_per_step": ms, "iters_per_s": 1e3 / ms, "gvoxel_updates_per_s": 2 * 512.0 ** 3 / ms / 1e6,
        "form": "multimodal.sirt_data_fusion(10, 0.05, 5) + tv_fgp_4D(5, 1e-4), 2 elements + HAADF, whole volume on one GPU"}
    facts["config5_chemical_engine_512x512x512_x70"] = engine_facts(mm.ce)
    facts["config5_haadf_engine_512x512x512_x70"] = engine_facts(mm.he)
    del mm
    # what the engines of these configs cost to create: device tables (all kernel families, built eagerly) and wall clock
    out["engine_creation"] = facts
    return out


def comm_rounds(t):
    """RCCL rounds this rank's engine has enqueued so far (tomo_get_option "comm_rounds"); None off the GPU / without the library."""
    try:
        return t.get_option("comm_rounds")
    except Exception:  # noqa: BLE001 -- the numpy slab double of the launcher test
        return None


def sharded_run_record(t, comm, rank, world, nglobal, n, nproj, ang, args, rounds0, rounds1):
    """What a sharded run can prove about itself (collective: every rank calls it).  The native communicator's world and
    this rank's place in it (tomo_comm_info), the device every rank sits on, the RCCL rounds a step enqueued (counted by
    the library), and a PARITY CHECK: one ASD-POCS iteration from zero on the sharded engine against the same iteration
    on ONE engine holding the whole volume (rank 0's device), compared on rank 0's first 64 slices (bound 2e-6, the
    sharded == whole-slab bar of tests/test_gpu_sharded_tv.py)."""
    import ctypes
    from tomo_tv_amd._lib import check
    rec = {"torch_world": world, "backend": args.backend}
    w, r = ctypes.c_int(0), ctypes.c_int(0)
    try:
        check(t.be.L.tomo_comm_info(t.be.h, ctypes.byref(w), ctypes.byref(r)))
        rec["native_world"], rec["native_rank_of_rank0"] = int(w.value), int(r.value)
    except Exception as e:  # noqa: BLE001
        rec["native_world"] = None
        rec["native_error"] = str(e)
    rec["native_collectives"] = bool(t._native())
    rec["device_of_rank"] = [int(d) for d in t.get_gpu_ids()]
    if rounds0 is not None and rounds1 is not None:
        rec["rccl_rounds_per_step"] = (rounds1 - rounds0) / max(1, args.steps)
    try:
        rec["rccl_version"] = t.get_option("rccl_version")      # ncclGetVersion's code as the library read it (22707 = 2.27.7)
    except Exception:  # noqa: BLE001 -- the numpy slab double of the launcher test
        pass
    if args.no_validate:
        rec["parity"] = None
        return rec
    # the check builds a second engine that holds the WHOLE volume on rank 0's device while the other ranks wait: skipped when
    # that engine would not fit beside this rank's slab (~14 volumes + ~16 GB of tables at 1024^2 x 120) -- ADVICE r4
    try:
        import torch
        free_b = float(torch.cuda.mem_get_info()[0])
    except Exception:  # noqa: BLE001
        free_b = None
    need_b = 14.0 * 4.0 * nglobal * n * n + 16e9 * (n / 1024.0) ** 2 * (nproj / 120.0)
    skip = free_b is not None and need_b > 0.8 * free_b
    flag = t.be.tensor([1.0 if (skip and rank == 0) else 0.0])
    comm.allreduce_sum(flag)
    if float(flag.item()) > 0:
        rec["parity"] = {"skipped": f"a whole-volume engine needs ~{need_b / 1e9:.0f} GB, {0 if free_b is None else free_b / 1e9:.0f} GB are free on rank 0's device"}
        return rec
    # one iteration from zero, sharded
    b_full = t._sino(0, dst=0)                                    # the tilt series assembled on rank 0 (gather)
    t.restart_recon()
    st = {"beta": 0.25, "i": 0, "dPOCS": 0.0, "norm": float(t.Nslice_ * t.Nrow)}
    asd_pocs_step(t, st)
    asd_pocs_flush(t, st)
    take = min(64, t.nloc)
    mine = t.get_volume_local()[:take]
    if rank == 0:
        from tomo_tv_amd.engine import tomoengine
        one = tomoengine(nglobal, n, ang, device=t.gpuID)
        for o in args.opt:
            k, v = o.split("=")
            one.set_option(k, int(v))
        one.set_tilt_series(b_full)
        one.initialize_SART("sequential")
        st1 = {"beta": 0.25, "i": 0, "dPOCS": 0.0, "norm": float(one.Nslice_ * one.Nrow)}
        asd_pocs_step(one, st1)
        asd_pocs_flush(one, st1)
        ref = one.get_volume_local()[:take]
        den = float(np.linalg.norm(ref.astype(np.float64).ravel()))
        err = float(np.linalg.norm((mine.astype(np.float64) - ref).ravel())) / (den if den > 0 else 1.0)
        rec["parity"] = {"what": f"one ASD-POCS iteration from zero: sharded over {world} rank(s) vs one engine holding all {nglobal} slices, "
                                 f"slices 0..{take - 1} of rank 0", "rel_l2": err, "bound": 2e-6, "ok": bool(err <= 2e-6)}
        del one
    comm.barrier()
    return rec


# ---- launcher: a plain `python bench.py --gpus N` starts its own N ranks ---------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpu_count():
    """GPUs this process could use, WITHOUT initialising one (the launcher must stay GPU-free: its children are the ranks)."""
    # from the kernel driver's topology (a node with SIMDs is a GPU), cut by the visibility lists: no runtime is loaded for it
    # (torch.cuda.device_count() can bring HIP / HSA up in this parent on ROCm -- ADVICE r4)
    try:
        import glob
        n = 0
        for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
            props = dict(ln.split(None, 1) for ln in open(f).read().splitlines() if " " in ln)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        if n == 0:
            raise OSError("no GPU node in the kfd topology")
        for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
            v = os.environ.get(var)
            if v is not None:
                n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
        return n
    except Exception:  # noqa: BLE001 -- no kfd topology to read: ask torch in a short-lived child, never in this process
        try:
            out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=300)
            return int(out.stdout.strip().splitlines()[-1])
        except Exception:  # noqa: BLE001 -- let the ranks find out
            return 1 << 30


def spawn_ranks(n, argv):
    """N fresh child processes (this parent never touches the GPU), rank 0's stdout passed through, every rank's stderr kept.
    The children are polled: when one exits non-zero the others -- which would sit in a collective until the process group's
    timeout -- are terminated at once and the launcher exits non-zero."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        # every rank builds its own copy of the tables: share the host CPUs between the ranks
        env.setdefault("TOMO_BUILD_THREADS", str(max(2, (os.cpu_count() or 8) // n)))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    live = dict(enumerate(procs))
    rc = 0
    while live:
        for r, p in list(live.items()):
            code = p.poll()
            if code is None:
                continue
            del live[r]
            if code != 0:
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr, flush=True)
                rc = max(rc, abs(code))
                for q in live.values():
                    q.terminate()
                t_end = time.time() + 10
                for q in live.values():
                    try:
                        q.wait(timeout=max(0.1, t_end - time.time()))
                    except subprocess.TimeoutExpired:
                        q.kill()
                return rc
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", "--nray", dest="n", type=int, default=512, help="Nray = Ny = Nz (--nray: torch.distributed.run reads a bare --n as one of its own options)")
    ap.add_argument("--nslice", type=int, default=512, help="slices of the volume (strong) / per GPU (weak)")
    ap.add_argument("--nproj", type=int, default=90)
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="strong: ONE volume of --nslice slices sharded over the GPUs (BASELINE's metric); weak: --nslice per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--quick", action="store_true", help="headline only: no secondary configs, no CPU baselines")
    ap.add_argument("--no-kernel-log", action="store_true", help="experiment: time the steps without the per-launch HIP events (no roofline)")
    ap.add_argument("--force-dist", action="store_true", help="use the slab-sharded engine + its communicator even with one rank")
    ap.add_argument("--no-validate", action="store_true", help="sharded runs: skip the parity check against one whole-volume engine")
    ap.add_argument("--backend", default="nccl", help="process-group backend (nccl = RCCL); tests drive the launcher with gloo")
    ap.add_argument("--opt", action="append", default=[], help="engine option name=int (tomo_set_option), repeatable")
    ap.add_argument("--sub-slabs", type=int, default=1,
                    help="run the GPU's slab as K sub-slab engines side by side (engine.py: _GroupBackend; single GPU only). "
                         "Measured -1...-3 %% per step at K = 2 and box-dependent, so the default is one engine")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        if args.backend == "nccl":
            ndev = visible_gpu_count()
            if args.gpus > ndev:
                raise SystemExit(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) are visible on this node "
                                 "(one rank per GPU; HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES restrict the list)")
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    # the one JSON line goes to the real stdout; everything else a library prints there (RCCL's version banner) is sent
    # to stderr
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:   # launched by torch.distributed.run: the same CPU sharing as spawn_ranks
        os.environ.setdefault("TOMO_BUILD_THREADS", str(max(2, (os.cpu_count() or 8) // int(os.environ.get("LOCAL_WORLD_SIZE", world)))))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import ctypes
    from tomo_tv_amd._lib import K_BP_ANGLE, K_SART_FUSED, K_SART_RESIDENT, VOL_ORIGINAL
    from tomo_tv_amd.distributed import slab_partition
    from tomo_tv_amd.phantom import ellipsoids, tilt_angles

    n, nproj = args.n, args.nproj
    nglobal = args.nslice * (world if args.scaling == "weak" else 1)
    ang = np.deg2rad(tilt_angles(nproj))
    comm = None
    on_gpu = True
    if world > 1 or args.force_dist:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            from tomo_tv_amd.engine import multigpuengine
            if local_rank >= torch.cuda.device_count():
                raise SystemExit(f"bench.py: rank {rank} (LOCAL_RANK {local_rank}) has no GPU: {torch.cuda.device_count()} visible")
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            t = multigpuengine(nglobal, n, ang, force_collectives=args.force_dist)   # one rank: still issue the RCCL calls
        else:                                            # launcher test on a CPU box: the numpy slab double of tests/
            on_gpu = False
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from slab_double import OracleSlabBackend
            from tomo_tv_amd.distributed import SlabComm
            from tomo_tv_amd.engine import tomoengine

            class DoubleEngine(tomoengine):
                _backend_cls = OracleSlabBackend

                def synchronize(self):
                    pass
            dist.init_process_group(args.backend, rank=rank, world_size=world)
            t = DoubleEngine(nglobal, n, ang, comm=SlabComm())
        comm = t.comm
    else:
        from tomo_tv_amd.engine import tomoengine
        t = tomoengine(nglobal, n, ang, device=0, sub_slabs=max(1, args.sub_slabs))
    first, nloc = slab_partition(nglobal, world, rank) if comm is not None else (0, nglobal)
    # synthetic data: strong scaling -> this rank's slab of ONE seeded phantom; weak -> the same phantom on every rank
    if args.scaling == "strong":
        vol = ellipsoids(nglobal, n, first=first, count=nloc)      # only this rank's slab is ever generated
    else:
        vol = ellipsoids(nloc, n)
    t.be.c("set_volume", VOL_ORIGINAL, vol.ctypes.data_as(ctypes.c_void_p))
    del vol
    t.create_projections()
    t.initialize_SART("sequential")
    for o in args.opt:
        k, v = o.split("=")
        t.set_option(k, int(v))
    t.restart_recon()
    st = {"beta": 0.25, "i": 0, "dPOCS": 0.0, "norm": float(t.Nslice_ * t.Nrow)}

    def sync():
        if on_gpu:
            t.synchronize()
            if comm is not None:
                import torch
                torch.cuda.synchronize()
        if comm is not None:
            comm.barrier()

    for _ in range(args.warmup):
        asd_pocs_step(t, st)
    sync()
    # the fused step and the plain per-angle FP run as k_sart_tile<..> unless --opt sart_tile=0 selects the ray-walk form
    tile = not any(o.replace(" ", "") == "sart_tile=0" for o in args.opt)
    K_FUSED_NAME, K_FP_NAME = ("k_sart_tile<true>", "k_sart_tile<false>") if tile else ("k_sart_seg<4,8,true>", "k_sart_seg<4,8,false>")
    K_BP_NAME = "k_bp_angle<4,4,true>"   # the sweep's last back-projection, tracked form (also step norm + snapshot copy)
    K_TVN_NAME, K_TVU_NAME = "k_tv_march4<8,*,TVM_NORM>", "k_tv_march4<8,false,TVM_UPDATE>"
    # every 4th fused step (89 per sweep) and every 2nd TV pass (10 + 10 per step) are timed; the two single launches all
    # (two sub-slab chains: every fused step is timed -- "achieved" is all launches' bytes over the time at least one of them was
    # executing, which needs them all; the pairs cost less there because the other stream's kernel fills the gap: 0.2 ms per step)
    two_chains = sart_chains(t) > 1
    # round 5: the sweep as ONE launch of the volume-resident kernel (N % 8 == 0, one 32 x 32 tile per CU at most): it replaces the
    # per-angle fused steps, the first projection and the tracked last back-projection
    resident = on_gpu and resident_sweep(t)
    K_RES_NAME = "k_sart_resident"
    if resident:
        LOG_STRIDE = {K_TVN_NAME: 2, K_TVU_NAME: 2}
        log_ids = {K_RES_NAME: K_SART_RESIDENT, K_TVN_NAME: 2, K_TVU_NAME: 3}
    else:
        LOG_STRIDE = {K_FUSED_NAME: 1 if two_chains else 4, K_TVN_NAME: 2, K_TVU_NAME: 2}
        log_ids = {K_FUSED_NAME: K_SART_FUSED, K_BP_NAME: K_BP_ANGLE, K_FP_NAME: 1, K_TVN_NAME: 2, K_TVU_NAME: 3}
    log = KernelLog(t, log_ids, LOG_STRIDE) if on_gpu and not args.no_kernel_log else None
    sync()
    rounds0 = comm_rounds(t)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        asd_pocs_step(t, st)
    dd, tv = asd_pocs_flush(t, st)          # the last step's scalars (inside the timed region)
    sync()
    el = time.perf_counter() - t0
    rounds1 = comm_rounds(t)
    prof = log.read() if log else {}
    iso = None
    chains = 1 if resident else sart_chains(t)          # launch chains per sweep of this rank's slab, as the engine runs them now
    if on_gpu and getattr(t, "sub_slabs", 1) == 1 and chains > 1 and not resident:
        # the dominant kernel alone on the chip (one chain, one stream), one untimed step: the kernel's own rate
        t.set_option("sart_streams", 1)
        log1 = KernelLog(t, {K_FUSED_NAME: K_SART_FUSED})
        asd_pocs_step(t, st)
        iso = log1.read()[K_FUSED_NAME]
        mode = 0
        for o in args.opt:
            k, _, v = o.replace(" ", "").partition("=")
            if k == "sart_streams":
                mode = int(v)
        t.set_option("sart_streams", mode)
    # Transparency: k_sart_tile stores only the 256-byte pieces whose bits changed (voxels held at zero by the positivity
    # clamp, rays with a zero residual -- data-dependent).  The same step with every voxel stored, timed after the timed region:
    el_all = None
    if on_gpu and tile and not resident and not any(o.replace(" ", "").startswith("sart_skip_same=") for o in args.opt):
        t.set_option("sart_skip_same", 0)
        asd_pocs_step(t, st)
        sync()
        ta = time.perf_counter()
        for _ in range(min(3, args.steps)):
            asd_pocs_step(t, st)
        sync()
        el_all = (time.perf_counter() - ta) / min(3, args.steps)
        t.set_option("sart_skip_same", 1)
    per_rank = None
    if comm is not None:
        tt = t.be.tensor([el])
        comm.allreduce_max(tt)
        el = float(tt.item())
        if prof:   # the dominant kernel's mean launch time on every rank
            kdom = K_RES_NAME if resident else K_FUSED_NAME
            mine = prof[kdom][2] / max(prof[kdom][0], 1)   # busy time per launch
            allv = t.be.tensor([mine if r == rank else 0.0 for r in range(world)])
            comm.allreduce_sum(allv)
            per_rank = [float(v) for v in allv.tolist()]

    comm_record = None
    if comm is not None and on_gpu:
        comm_record = sharded_run_record(t, comm, rank, world, nglobal, n, nproj, ang, args, rounds0, rounds1)

    if rank == 0:
        vox_total = float(nglobal) * n * n
        # Algorithmic bytes per launch (SURVEY.md section 8d, V = voxels of this GPU's slab, fp32):
        #   k_bp_angle<..true>  single-angle voxel update that also forms the step norm and refreshes the snapshot:
        #                       slab + snapshot in, slab + snapshot out, that angle's residual rows = 16V + 4 Nx N
        #   fused step          BP(a_k)+FP(a_k+1): slab in + slab out + residual rows in + b rows in + residual rows
        #                       out = 8V + 12 Nx N   (the tile form's partial sums are traffic, not algorithmic bytes)
        #   per-angle FP        slab in + b rows in + residual rows out = 4V + 8 Nx N
        V = float(nloc) * n * n
        #   TV norm pass        reads the slab, writes nothing = 4V
        #   TV update pass      re-evaluates g, writes x_new = 8V; the last of the 10 also reads and rewrites the snapshot (16V)
        alg_bytes = {K_BP_NAME: 16.0 * V + 4.0 * nloc * n, K_FUSED_NAME: 8.0 * V + 12.0 * nloc * n,
                     K_FP_NAME: 4.0 * V + 8.0 * nloc * n, K_TVN_NAME: 4.0 * V, K_TVU_NAME: (9 * 8.0 + 16.0) / 10 * V}
        roofs = {}
        nsub = chains * max(1, getattr(t, "sub_slabs", 1) if t is not None else 1)   # launches per angle
        for name, (cnt, tot, busy) in prof.items():
            if name == K_RES_NAME:      # one launch = one sweep of (this engine's share of) the slab
                nsl = max(1, getattr(t, "sub_slabs", 1) if t is not None else 1)
                roofs[name] = resident_roof(cnt, tot, busy, nloc // nsl, n, nproj, True)
                roofs[name]["launches_per_sweep"] = nsl
                roofs[name]["sample_stride"] = 1
                continue
            per = nsub if name in (K_BP_NAME, K_FUSED_NAME, K_FP_NAME) else max(1, getattr(t, "sub_slabs", 1) if t is not None else 1)
            roofs[name] = roof(name, cnt, tot, alg_bytes[name] / per, busy_ms=busy)
            roofs[name]["launches_per_angle"] = per    # sweep chains x sub-slab engines: one launch covers 1/per of the slab
            roofs[name]["sample_stride"] = LOG_STRIDE.get(name, 1)     # every N-th launch was timed: launches / total_ms count those
            # what an in-place read-modify-write pass over the slab reaches on this part in any access pattern
            # (tools/micro/copy_patterns.hip: 5.2-5.5 TB/s) -- informative, not the peak
            roofs[name]["frac_of_measured_rmw_ceiling"] = roofs[name]["achieved"] / RMW_CEILING_GBS
        attach_traffic(roofs, (nloc, n, nproj))
        dominant = max(roofs.values(), key=lambda r: r["total_ms"] * r["sample_stride"]) if roofs else None
        if iso is not None and dominant is not None:
            r1 = roof(K_FUSED_NAME, iso[0], iso[1], alg_bytes[K_FUSED_NAME], busy_ms=iso[2])
            dominant = dict(dominant, isolated={k: r1[k] for k in ("achieved", "frac", "avg_ms", "launches", "algorithmic_bytes_per_launch")},
                            isolated_note="the same kernel with sart_streams=1 (one launch per angle over the whole slab, nothing "
                                          "else on the chip), one untimed step after the timed region")
        if dominant is not None and per_rank is not None:
            dominant = dict(dominant, avg_ms_per_rank=per_rank)
        shape = f"{nglobal}x{n}x{n}"
        out = {
            "metric": f"SART+TV Gvoxel-updates/s (ASD-POCS outer iterations x voxels, {shape} x {nproj} tilts, at 1/2/4/8 GPU)",
            "value": vox_total * args.steps / el / 1e9,
            "unit": "Gvoxel-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3,
            "iters_per_s": args.steps / el,
            # every angle of the SART sweep updates every voxel (SURVEY.md section 8d asks for this figure as well)
            "sart_gvoxel_angle_updates_per_s": vox_total * nproj * args.steps / el / 1e9,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"ASD-POCS iteration (SART sweep beta0=0.25 + 10 TV-GD steps) on ONE {shape} volume, {nproj} tilts "
                                   f"-70..70 deg" + (" (headline SART+TV 512^3x90 = BASELINE configs[2] shape)" if (nglobal, n, nproj) == (512, 512, 90) else "")
                                   + f", {nloc} slices on rank 0",
                       "volume": shape, "slices_per_gpu": nloc, "nray": n, "nproj": nproj,
                       "sharding": f"tilt-axis slabs x{world} ({args.scaling} scaling)",
                       "sub_slab_engines_per_gpu": getattr(t, "sub_slabs", 1), "sart_chains_per_engine": chains,
                       "forms": engine_forms(t), "engine": engine_facts(t)},
            # the data-INDEPENDENT figure (k_sart_tile storing every voxel; the headline skips stores of unchanged 256-byte pieces,
            # which the zero background of the synthetic phantom favours): null when the option was forced on the command line
            # (round 5, resident sweep: every voxel is loaded and stored once per sweep whatever the data -- the headline IS that figure)
            "ms_per_step_every_voxel_stored": (el / args.steps * 1e3 if resident else None) if el_all is None else el_all * 1e3,
            "sart_sweep_form": "k_sart_resident (one launch per sweep, volume resident in registers)" if resident else "k_sart_tile chain (one fused step per angle)",
            "final_dd": dd, "final_tv": tv,
            "store_skipping": None if el_all is None else {
                "ms_per_step_with_every_voxel_stored": el_all * 1e3,
                "note": "k_sart_tile (in place) skips the store of 256-byte pieces whose bits did not change; bit-identical results; the "
                        "gain depends on the data (zero background of the synthetic phantom); --opt sart_skip_same=0 times the other form"},
            # N > 1 (and --force-dist): what the run itself can prove about its communication
            "comm": comm_record,
            # the plain-process multi-GPU facade's host cost per step, at this run's GPU count and at 8 (host-only)
            "inprocess_facade": [facade_overhead(max(2, world)), facade_overhead(8)],
            "roofline": dominant,
            "roofline_bp_angle": roofs.get(K_BP_NAME),
            "roofline_fp_angle": roofs.get(K_FP_NAME),
            # the TV descent: a norm pass + a recompute-and-update pass per inner iteration (gradient never stored); these two
            # are about half instruction-bound, their launches overlap the data distance on the second stream
            "roofline_tv_norm": roofs.get(K_TVN_NAME),
            "roofline_tv_update": roofs.get(K_TVU_NAME),
        }
        if world == 1 and comm is None and not args.quick:
            del t, log
            t = log = None
            out["secondary"] = secondary_configs()
            # north_star's ">= 60 % of the HBM roofline on the back-projection kernel": the single-angle voxel update that streams the
            # slab (k_sart_tile<true> = BP(a_k) + FP(a_k+1) in one pass; the sweep form of N = 1024, where the volume cannot be resident)
            shard4 = out["secondary"].get("config4_shard_asd_pocs_128x1024sq_x120tilts", {})
            out["roofline_back_projection_update"] = dict(shard4.get("roofline") or {}, measured_on="config 4 shard: 128 x 1024^2 x 120, ASD-POCS step")
            if not args.no_cpu_baseline:
                out["cpu_baseline"], out["cpu_baseline_configs"] = cpu_baselines(n, nproj)
        out["frac_above_one"] = fracs_above_one(out)       # must be empty: a roofline fraction above 1 is not a roofline fraction
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if comm is not None:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
