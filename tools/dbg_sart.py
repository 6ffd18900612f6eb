import sys, numpy as np
sys.path.insert(0, '.')
from tomo_tv_amd.engine import tomoengine
from tomo_tv_amd.phantom import ellipsoids
from tomo_tv_amd._lib import VOL_ORIGINAL
def rl(a, b): return np.linalg.norm(a - b) / np.linalg.norm(b)
for (Nx, N, P, niter, order) in [(130, 24, 5, 2, "random"), (130, 24, 5, 2, "sequential"), (64, 24, 5, 2, "sequential"), (256, 24, 5, 2, "sequential")]:
    ang = np.deg2rad(np.linspace(-70, 70, P))
    x = ellipsoids(Nx, N, seed=3)
    res = {}
    for rep in range(4):
        for fused in (2, 0):
            dev = tomoengine(Nx, N, ang)
            dev.set_option("sart_fused", fused)
            dev.set_volume(x, VOL_ORIGINAL)
            dev.create_projections()
            b = dev.get_projections()
            dev.initialize_SART(order)
            dev.SART(0.7, niter)
            res[(rep, fused)] = (b, dev.get_volume())
    print(Nx, N, P, order)
    for k in sorted(res):
        print('  ', k, 'b vs ref: %.3e' % rl(res[k][0], res[(0, 0)][0]), ' vol: %.3e' % rl(res[k][1], res[(0, 0)][1]), 'pad b max', np.abs(res[k][0]).max())
