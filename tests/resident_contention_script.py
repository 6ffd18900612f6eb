"""Child of tests/test_gpu_sart_resident.py::test_two_processes_share_the_device: repeated resident SART sweeps on a 512^2 slab (256
workgroups, one per CU) while ANOTHER process does the same on the same GPU.  Nothing orders the launches of two processes; a sweep
whose workgroups cannot all be resident gives up, stores nothing, and is redone by the streamed chain.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tomo_tv_amd._lib import VOL_ORIGINAL, VOL_RECON  # noqa: E402
from tomo_tv_amd.engine import tomoengine  # noqa: E402
from tomo_tv_amd.phantom import ellipsoids, tilt_angles  # noqa: E402

seed, sweeps, spin, start_at = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
ns, n, nproj = 64, 512, 60


def engine(resident):
    t = tomoengine(ns, n, np.deg2rad(tilt_angles(nproj)))
    t.set_option("sart_resident", resident)
    t.set_volume(ellipsoids(ns, n, seed=seed), VOL_ORIGINAL)
    t.create_projections()
    return t


ref = engine(0)
for _ in range(sweeps):
    ref.SART(0.3, 1)
want = ref.get_volume(VOL_RECON)
del ref
t = engine(1)
t.set_option("sart_resident_spin", spin)
t.SART(0.3, 1)          # warm (tables, buffers, the angle sequence)
t.restart_recon()
while time.time() < start_at:      # both processes start their sweeps together
    time.sleep(0.001)
t0 = time.time()
for _ in range(sweeps):
    t.SART(0.3, 1)
dt = time.time() - t0
got = t.get_volume(VOL_RECON)
err = float(np.linalg.norm(got.astype(np.float64) - want) / np.linalg.norm(want))
print(json.dumps({"rel": err, "finite": bool(np.isfinite(got).all()), "fallbacks": t.get_option("sart_resident_fallbacks"),
                  "fallback_chunks": t.get_option("sart_resident_fallback_chunks"), "seconds": dt, "sweeps": sweeps}))
