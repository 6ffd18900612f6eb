#!/usr/bin/env python3
"""What-if timing of k_sart_tile<true> (library built with `make -C tomo_tv_amd/csrc -B EXTRA=-DTOMO_WHATIF`): parts of the
kernel switched off one at a time; results are wrong, times tell where a launch spends its 210 us."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomo_tv_amd import _lib
from tomo_tv_amd._lib import K_SART_FUSED, VOL_ORIGINAL
from tomo_tv_amd.engine import tomoengine
from tomo_tv_amd.phantom import ellipsoids, tilt_angles
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 512
t = tomoengine(nx, 512, np.deg2rad(tilt_angles(90)))
t.set_volume(ellipsoids(nx, 512), VOL_ORIGINAL); t.create_projections(); t.initialize_SART("sequential")
names = {0: "full kernel", 1: "no x stores", 2: "no BP arithmetic (copy through)", 4: "no FP phase", 8: "no window / cell staging",
         16: "no tile loads", 4 | 2: "no FP, no BP arithmetic (stream copy)", 4 | 2 | 8: "stream copy, no staging", 1 | 4: "loads + BP only",
         16 | 1 | 8: "FP phase only (no loads, no stores, no staging)", 16 | 1 | 8 | 2: "LDS image + FP only"}
for wi, nm in names.items():
    t.set_option("sart_whatif", wi)
    t.restart_recon(); t.SART(0.5, 1); t.synchronize()
    _lib.check(t.be.L.tomo_profile_enable(t.be.h, K_SART_FUSED, 1))
    t.SART(0.5, 2)
    n, ms = ctypes.c_int64(0), ctypes.c_double(0)
    _lib.check(t.be.L.tomo_profile_read(t.be.h, K_SART_FUSED, ctypes.byref(n), ctypes.byref(ms)))
    _lib.check(t.be.L.tomo_profile_enable(t.be.h, K_SART_FUSED, 0))
    print(f"whatif {wi:2d}  {nm:48s} {ms.value / n.value * 1e3:7.1f} us per launch")
