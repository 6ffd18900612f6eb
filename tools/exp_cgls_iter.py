import sys, time, numpy as np
sys.path.insert(0, ".")
from tomo_tv_amd.engine import tomoengine
from tomo_tv_amd.phantom import ellipsoids, tilt_angles
from tomo_tv_amd._lib import VOL_ORIGINAL
t = tomoengine(512, 512, np.deg2rad(tilt_angles(90)))
t.set_volume(ellipsoids(512, 512), VOL_ORIGINAL); t.create_projections(); t.restart_recon()
def tm(n):
    t.CGLS(n); t.synchronize()
    t0 = time.perf_counter(); t.CGLS(n); t.synchronize(); return (time.perf_counter() - t0) * 1e3
a, b = tm(1), tm(11)
print(f"CGLS(1) {a:.2f} ms, CGLS(11) {b:.2f} ms -> {(b - a) / 10:.2f} ms per additional iteration")
