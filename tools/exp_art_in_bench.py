#!/usr/bin/env python3
"""Run bench.py's secondary configs with the ctvlib facade's calls timed one by one (why is the ART-form step 28 ms there, 23 ms alone?)."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tomo_tv_amd import engine
acc = collections.defaultdict(lambda: [0.0, 0])
def wrap(name):
    f = getattr(engine.ctvlib, name)
    def g(self, *a, **k):
        self.synchronize(); t0 = time.perf_counter(); r = f(self, *a, **k); self.synchronize()
        acc[name][0] += time.perf_counter() - t0; acc[name][1] += 1
        return r
    setattr(engine.ctvlib, name, g)
for n in ("ART", "copy_recon", "matrix_2norm", "data_distance", "tv", "tv_gd"):
    wrap(n)
out = bench.secondary_configs()
print("asd_pocs_art", out["asd_pocs_art_512cube_x90tilts"]["ms_per_step"])
for k, (s, c) in acc.items():
    print(f"  {k:14s} {s / c * 1e3:8.2f} ms x {c}")
