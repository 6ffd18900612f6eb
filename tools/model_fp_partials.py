#!/usr/bin/env python3
"""How many partial sums would other decompositions of the all-angle forward projector emit?  (model on the real matrix)

Counts, for the parallelRay matrix at N x P, the (ray, block) incidences = partial sums per 64-slice chunk for:
  * axis-aligned tiles TY x TZ (what k_fp_tile does: 32 x 16),
  * sheared strips: per angle group g a strip is a band of W columns (or rows, for |theta| > 45 deg) whose offset moves with the
    row by round(y * tan(theta_g)): a ray emits one partial per strip it crosses.  Cost side: the volume is re-read once per group.
"""
import sys
import numpy as np
sys.path.insert(0, ".")
import oracle

N, P = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 90)
ang = np.linspace(-70, 70, P)
A = oracle.parallel_ray(N, ang)
row, col = A[0].astype(np.int64), A[1].astype(np.int64)
y, z = col // N, col % N
nnz = row.size
print(f"N={N} P={P} nnz={nnz} rays={N * P}")


def count(block_id):
    key = row * (1 << 32) + block_id
    return np.unique(key).size


for ty, tz in ((32, 16), (16, 32), (32, 32), (64, 32)):
    c = count((y // ty) * 4096 + z // tz)
    print(f"tiles {ty:3d} x {tz:3d}: {c:9d} partials = {c / (N * P):6.2f} per ray   bytes/chunk {c * 256 / 1e6:8.1f} MB")

a_of_row = ang[(row // N)]
for W in (16, 32, 64):
    for ngroups in (2, 4, 6, 8):
        # classes: |theta| <= 45 -> column strips sheared along y; else row strips sheared along z
        steep = np.abs(a_of_row) > 45
        total = 0
        passes = 0
        for cls, mask in ((0, ~steep), (1, steep)):
            if not mask.any():
                continue
            t = np.tan(np.deg2rad(a_of_row[mask])) if cls == 0 else 1.0 / np.tan(np.deg2rad(a_of_row[mask]))
            # image convention: which way rays lean does not matter for the count as long as the shear sign is the better of the two
            edges = np.quantile(t, np.linspace(0, 1, ngroups // 2 + 1))
            u, v = (y[mask], z[mask]) if cls == 0 else (z[mask], y[mask])       # u: along the strip, v: across
            r = row[mask]
            for g in range(ngroups // 2):
                sel = (t >= edges[g]) & (t <= edges[g + 1]) if g == ngroups // 2 - 1 else (t >= edges[g]) & (t < edges[g + 1])
                if not sel.any():
                    continue
                tg = np.median(t[sel])
                best = None
                for sign in (1, -1):
                    strip = (v[sel] + np.round(sign * tg * u[sel]).astype(np.int64)) // W
                    c = np.unique(r[sel] * (1 << 32) + (strip + 100000)).size
                    best = c if best is None else min(best, c)
                total += best
                passes += 1
        print(f"sheared strips W={W:3d}, {passes} passes over the volume: {total:9d} partials = {total / (N * P):6.2f} per ray   "
              f"partial bytes/chunk {total * 256 / 1e6:8.1f} MB + volume re-reads {passes * N * N * 256 / 1e6:7.1f} MB")
