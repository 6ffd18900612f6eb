"""Host-side pieces of the multi-modal path that need no GPU."""
import os

import numpy as np

from conftest import GOLDEN
from tomo_tv_amd.chemistry import create_weighted_summation_weights, get_periodic_table


def test_summation_weights_equal_reference_matrix():
    """Weights vs the imported reference create_weighted_summation_matrix (tests/golden/sigma_*.npz)."""
    for fn, zs, method in [("sigma_m3_nz2.npz", [31, 8], 3), ("sigma_m1_nz2.npz", [31, 8], 1), ("sigma_m3_nz3.npz", [22, 38, 8], 3)]:
        g = np.load(os.path.join(GOLDEN, fn))
        w = create_weighted_summation_weights(zs, 1.6, method)
        npix, nel = int(g["shape"][0]), len(zs)
        indptr, indices, data = g["indptr"], g["indices"], g["data"]
        for p in (0, npix // 2, npix - 1):
            cols, vals = indices[indptr[p]:indptr[p + 1]], data[indptr[p]:indptr[p + 1]]
            assert np.array_equal(cols, p + npix * np.arange(nel)) and np.array_equal(vals, w)



def test_periodic_table():
    pt = get_periodic_table()          # fusion_helper.py:34-48
    assert pt["h"] == 1 and pt["zn"] == 30 and pt["o"] == 8 and pt["au"] == 79 and pt["rf"] == 104 and len(pt) == 104
