import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import bench
from tomo_tv_amd import pytvlib
from tomo_tv_amd._lib import VOL_YK
t = bench._engine(512, 512, 90)
pytvlib.initialize_algorithm(t, "fista")
st = {"t0": 1.0}
def it():
    pytvlib.run(t, "fista"); t.tv_fgp(10, 0.1, vol=VOL_YK)
    tk = 0.5 * (1 + np.sqrt(1 + 4 * st["t0"] ** 2)); t.fista_momentum((st["t0"] - 1) / tk); st["t0"] = tk
    c = 0.5 * t.data_distance() ** 2 + 0.1 * t.tv(); t.fista_project_yk(); return c
for r in (1, 0, 1):
    t.set_option("fp_reuse", r)
    print("fp_reuse", r, "ms", bench._time_steps(t, it, 5))
