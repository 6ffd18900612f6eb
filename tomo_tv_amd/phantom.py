"""Seeded synthetic inputs: numpy only, used by tests, bench.py and the golden generator."""
import numpy as np


def ellipsoids(nx, n, seed=1234, k=20, dtype=np.float32, first=0, count=None):
    """Sum of ``k`` random axis-aligned ellipsoids in a ``(nx, n, n)`` volume (axis 0 = tilt axis).

    Centres U(-0.6, 0.6)^3, semi-axes U(0.05, 0.3), amplitudes U(0.2, 1.0), clipped to [0, 1] and zeroed
    outside the inscribed cylinder of radius 0.95*n/2 around the tilt axis.
    ``first, count``: only the slab ``[first, first + count)`` of the ``nx`` slices is generated (a rank's shard of the
    volume: identical to slicing the whole phantom, without ever holding it).
    """
    rng = np.random.default_rng(seed)
    c = rng.uniform(-0.6, 0.6, (k, 3))
    r = rng.uniform(0.05, 0.3, (k, 3))
    a = rng.uniform(0.2, 1.0, k)
    gx = ((np.arange(nx) + 0.5) / nx * 2 - 1).astype(np.float32) if nx > 1 else np.zeros(1, np.float32)
    gy = ((np.arange(n) + 0.5) / n * 2 - 1).astype(np.float32)
    if count is None:
        count = nx - first
    if first < 0 or count < 0 or first + count > nx:
        raise ValueError("slab outside the volume")
    gx = gx[first:first + count]
    nx = count
    vol = np.zeros((nx, n, n), np.float32)
    yy, zz = np.meshgrid(gy, gy, indexing="ij")
    cyl = (yy * yy + zz * zz) <= 0.95 ** 2
    for e in range(k):
        # bounding box of the ellipse footprint in (y, z): only those pixels are tested
        jy = np.nonzero(np.abs(gy - c[e, 1]) <= r[e, 1])[0]
        jz = np.nonzero(np.abs(gy - c[e, 2]) <= r[e, 2])[0]
        if jy.size == 0 or jz.size == 0:
            continue
        ys, zs = slice(jy[0], jy[-1] + 1), slice(jz[0], jz[-1] + 1)
        q = ((yy[ys, zs] - c[e, 1]) / r[e, 1]) ** 2 + ((zz[ys, zs] - c[e, 2]) / r[e, 2]) ** 2
        for s in range(nx):
            dx = (gx[s] - c[e, 0]) / r[e, 0]
            rem = 1.0 - dx * dx
            if rem <= 0:
                continue
            vol[s, ys, zs] += np.where(q <= rem, np.float32(a[e]), np.float32(0))
    vol = np.clip(vol, 0, 1) * cyl[None]
    return vol.astype(dtype)


_SL = [  # (amplitude, a, b, x0, y0, phi_deg): modified Shepp-Logan
    (1.0, .69, .92, 0, 0, 0), (-.8, .6624, .8740, 0, -.0184, 0), (-.2, .1100, .3100, .22, 0, -18),
    (-.2, .1600, .4100, -.22, 0, 18), (.1, .2100, .2500, 0, .35, 0), (.1, .0460, .0460, 0, .1, 0),
    (.1, .0460, .0460, 0, -.1, 0), (.1, .0460, .0230, -.08, -.605, 0), (.1, .0230, .0230, 0, -.606, 0),
    (.1, .0230, .0460, .06, -.605, 0)]


def shepp_logan(n, dtype=np.float32):
    """2-D modified Shepp-Logan phantom, ``(n, n)``, values in [0, 1]."""
    g = ((np.arange(n) + 0.5) / n * 2 - 1)
    x, y = np.meshgrid(g, -g, indexing="xy")
    img = np.zeros((n, n), np.float64)
    for amp, a, b, x0, y0, phi in _SL:
        p = np.deg2rad(phi)
        xr = (x - x0) * np.cos(p) + (y - y0) * np.sin(p)
        yr = -(x - x0) * np.sin(p) + (y - y0) * np.cos(p)
        img[(xr / a) ** 2 + (yr / b) ** 2 <= 1] += amp
    return np.clip(img, 0, None).astype(dtype)


def tilt_angles(nproj, lo=-70.0, hi=70.0):
    """ET-style missing-wedge tilt scheme in degrees (cf. demo.ipynb: arange(-70, 72, 2))."""
    return np.linspace(lo, hi, nproj)
