#!/usr/bin/env python3
"""Headline benchmark: SART+TV (one ASD-POCS outer iteration) at 512^3 x 90 tilts per GPU.

    python bench.py --gpus N --steps K --warmup W

A step = one ASD-POCS outer iteration (examples/sim_ASD.py:66-94): copy_recon, one SART sweep over all
tilts, step norm, data distance (a full forward projection), copy_recon, 10 TV gradient-descent steps, step
norm (the engine forms the two norms and snapshot copies inside the sweep's last back-projection and the last
descent step: tomo_sart_tracked / tomo_tv_gd_tracked).  Synthetic phantom + tilt series are resident in HBM before the timed region.  N > 1: one rank per GPU,
each owns a 512-slice slab (weak scaling); the only cross-rank traffic is scalar all-reduces and TV halo planes.
Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
RMW_CEILING_GBS = 5300.0  # measured: tools/micro/copy_patterns.hip (read + write of the slab in place), DESIGN.md section 3


def asd_pocs_step(t, st):
    """One outer iteration, state dict st carries beta / dPOCS (defaults of gpu/reconstructor.py:158-161)."""
    if hasattr(t, "SART_tracked"):
        # engine (= TomoGPU.asd_pocs): the step norms and snapshot copies ride on the last back-projection / last
        # descent pass; the residual of the SART result runs on the second stream under the TV steps
        if st["i"] == 0:
            t.copy_recon()
            dp = t.SART_tracked(st["beta"], 1)
            st["dPOCS"] = dp * 0.2
        else:
            t.SART_tracked(st["beta"], 1, defer=True)       # its step norm is read with the other scalars below
        st["beta"] *= 0.9985
        t.data_distance_begin()
        if st["i"] == 0:
            tv, dg, dd2 = t.tv_gd_tracked(10, st["dPOCS"], extra=(S_DD,))
        else:
            tv, dg, dd2, dp2 = t.tv_gd_tracked(10, st["dPOCS"], extra=(S_DD, S_DIFF2))
            dp = dp2 ** 0.5
        dd = dd2 ** 0.5 / st["norm"]
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
        dd = t.data_distance() / st["norm"]
        tv = t.tv_gd(10, st["dPOCS"])
        dg = t.matrix_2norm()
    if dg > dp * 0.95 and dd > 0.025:
        st["dPOCS"] *= 0.95
    st["i"] += 1
    return dd, tv


def cpu_baseline(n, nproj, budget_s=20.0):
    """The C oracle (OpenMP over slices, like ctvlib.cpp:207) on a bounded sample of the same workload."""
    import oracle
    from tomo_tv_amd.phantom import ellipsoids, tilt_angles
    if "OMP_NUM_THREADS" not in os.environ:
        oracle.set_num_threads(oracle.usable_cpus())   # the baseline uses every CPU the host grants
    threads = oracle.num_threads()
    ns = max(8, 2 * threads)
    ang = tilt_angles(nproj)
    A = oracle.parallel_ray(n, ang)
    ref = oracle.ctvlib(ns, n, nproj)
    ref.load_A(A)
    ref.original_volume = ellipsoids(ns, n)
    ref.create_projections()
    ref.initialize_recon_copy()
    ref.tv_eps = 1e-6
    st = {"beta": 0.25, "i": 0, "dPOCS": 0.0, "norm": float(ns * n * nproj)}
    # the oracle has the same method names, so the same step function drives it
    ref.data_distance_n = ref.data_distance
    class _Shim:
        def __getattr__(self, k):
            return getattr(ref, k)
        def data_distance(self):
            return ref.data_distance(normalize=False)
    shim = _Shim()
    t0 = time.perf_counter()
    iters = 0
    while True:
        asd_pocs_step(shim, st)
        iters += 1
        el = time.perf_counter() - t0
        if el > budget_s or iters >= 5:
            break
    vox = ns * n * n
    return {"value": vox * iters / el / 1e9, "unit": "Gvoxel-updates/s", "cores": threads, "kind": "port",
            "sample": f"{iters} ASD-POCS iterations on a {ns}x{n}x{n} slab, {nproj} tilts (oracle/tomo_oracle.c, OpenMP over slices)",
            "iters_per_s_full_volume_equiv": iters / el * ns / n}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=512, help="Nray = Ny = Nz")
    ap.add_argument("--nslice", type=int, default=512, help="slices per GPU")
    ap.add_argument("--nproj", type=int, default=90)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="use the slab-sharded engine + RCCL even with one rank")
    ap.add_argument("--opt", action="append", default=[], help="engine option name=int (tomo_set_option), repeatable")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    from tomo_tv_amd import _lib
    from tomo_tv_amd._lib import K_BP_ANGLE, K_SART_FUSED, VOL_ORIGINAL
    from tomo_tv_amd.engine import multigpuengine, tomoengine
    from tomo_tv_amd.phantom import ellipsoids, tilt_angles
    import ctypes

    n, nproj, nloc = args.n, args.nproj, args.nslice
    ang = np.deg2rad(tilt_angles(nproj))
    comm = None
    if world > 1 or args.force_dist:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        t = multigpuengine(nloc * world, n, ang)
        comm = t.comm
    else:
        t = tomoengine(nloc, n, ang, device=0)
    # synthetic data: every rank's slab is the same seeded phantom (weak scaling: identical per-GPU work)
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
        t.synchronize()
        if comm is not None:
            comm.barrier()

    for _ in range(args.warmup):
        asd_pocs_step(t, st)
    sync()
    # the fused step and the plain per-angle FP run as k_sart_tile<..> unless --opt sart_tile=0 selects the ray-walk form
    tile = not any(o.replace(" ", "") == "sart_tile=0" for o in args.opt)
    K_FUSED_NAME, K_FP_NAME = ("k_sart_tile<true>", "k_sart_tile<false>") if tile else ("k_sart_seg<4,8,true>", "k_sart_seg<4,8,false>")
    K_BP_NAME = "k_bp_angle<4,4,true>"   # the sweep's last back-projection, tracked form (also step norm + snapshot copy)
    kernels = {K_FUSED_NAME: K_SART_FUSED, K_BP_NAME: K_BP_ANGLE, K_FP_NAME: 1}   # names as rocprofv3 prints them
    for kid in kernels.values():
        _lib.check(t.be.L.tomo_profile_enable(t.be.h, kid, 1))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dd, tv = asd_pocs_step(t, st)
    sync()
    el = time.perf_counter() - t0
    prof = {}
    for name, kid in kernels.items():
        launches, total_ms = ctypes.c_int64(0), ctypes.c_double(0)
        _lib.check(t.be.L.tomo_profile_read(t.be.h, kid, ctypes.byref(launches), ctypes.byref(total_ms)))
        _lib.check(t.be.L.tomo_profile_enable(t.be.h, kid, 0))
        prof[name] = (int(launches.value), float(total_ms.value))
    if comm is not None:
        import torch
        tt = torch.tensor([el], dtype=torch.float64, device="cuda")
        comm.allreduce_max(tt)
        el = float(tt.item())

    if rank == 0:
        vox_total = nloc * world * n * n
        # Algorithmic bytes per launch (SURVEY.md section 8d, V = voxels of this GPU's slab, fp32):
        #   k_bp_angle<..true>  single-angle voxel update that also forms the step norm and refreshes the snapshot:
        #                       slab + snapshot in, slab + snapshot out, that angle's residual rows = 16V + 4 Nx N
        #   fused step          BP(a_k)+FP(a_k+1): slab in + slab out + residual rows in + b rows in + residual rows
        #                       out = 8V + 12 Nx N   (the tile form's partial sums are traffic, not algorithmic bytes)
        #   per-angle FP        slab in + b rows in + residual rows out = 4V + 8 Nx N
        V = float(nloc) * n * n
        alg_bytes = {K_BP_NAME: 16.0 * V + 4.0 * nloc * n, K_FUSED_NAME: 8.0 * V + 12.0 * nloc * n,
                     K_FP_NAME: 4.0 * V + 8.0 * nloc * n}
        roofs = {}
        for name, (cnt, tot) in prof.items():
            avg_ms = tot / cnt if cnt else 0.0
            ach = alg_bytes[name] / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            roofs[name] = {"kernel": name, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": ach / HBM_PEAK_GBS, "traffic": None, "launches": cnt, "avg_ms": avg_ms,
                           "total_ms": tot, "algorithmic_bytes_per_launch": alg_bytes[name],
                           # what an in-place read-modify-write pass over the slab reaches on this part in any access
                           # pattern (tools/micro/copy_patterns.hip: 5.2-5.5 TB/s) -- informative, not the peak
                           "frac_of_measured_rmw_ceiling": ach / RMW_CEILING_GBS}
        # HBM traffic per launch from the committed PMC passes (profiles/r01_pmc_traffic.json), null if absent
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))["kernels"]
            match = {"k_sart_seg<4,8,true>": "k_sart_seg<4, 8, true>", "k_bp_angle<4,4,true>": "k_bp_angle<4, 4, true>",
                     "k_sart_seg<4,8,false>": "k_sart_seg<4, 8, false>", "k_sart_tile<true>": "k_sart_tile<true>",
                     "k_sart_tile<false>": "k_sart_tile<false>"}
            if (nloc, n, nproj) == (512, 512, 90):
                for name in roofs:
                    key = match[name]
                    hit = [v for k, v in pmc.items() if key in k]
                    if hit:
                        roofs[name]["traffic"] = hit[0]["hbm_bytes_per_launch"]
                        roofs[name]["traffic_source"] = "profiles/r01_pmc_traffic.json (rocprofv3 --pmc, separate passes, FETCH_SIZE x2)"
        except (OSError, KeyError, ValueError):
            pass
        dominant = max(roofs.values(), key=lambda r: r["total_ms"])
        out = {
            "metric": "SART+TV Gvoxel-updates/s (ASD-POCS outer iterations x voxels, 512^3 x 90 tilts per GPU)",
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
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"ASD-POCS iteration (SART sweep beta0=0.25 + 10 TV-GD steps), {nloc}x{n}x{n} voxels per GPU, "
                                   f"{nproj} tilts -70..70 deg (BASELINE configs[2] shape; headline SART+TV 512^3x90)",
                       "slices_per_gpu": nloc, "nray": n, "nproj": nproj, "sharding": f"tilt-axis slabs x{world}"},
            "final_dd": dd, "final_tv": tv,
            "roofline": dominant,
            "roofline_bp_angle": roofs[K_BP_NAME],
            "roofline_fp_angle": roofs[K_FP_NAME],
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, nproj)
        print(json.dumps(out), flush=True)
    if comm is not None:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
