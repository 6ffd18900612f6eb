"""String -> method dispatch, mirroring tomofusion/pytvlib.py:5-39 (GPU engines) and
tomofusion/cpu/utils/pytvlib.py:171-213 (ctvlib harness helpers).  The file helpers of the reference
(pytvlib.py:57-162) live in ``tomo_tv_amd/io.py`` and are re-exported here under the same names."""
import numpy as np

from .engine import system_matrix
from .io import load_data, load_h5_data, mpi_save_results, save_gif, save_recon, save_results  # noqa: F401


def initialize_algorithm(tomo, alg, initAlg=""):
    """tomofusion/pytvlib.py:5-19."""
    a = alg.lower()
    if a == "sirt":
        tomo.initialize_SIRT()
    elif a == "cgls":
        tomo.initialize_CGLS()
    elif a == "fista":
        tomo.initialize_fista()
    elif a in ("poisson_ml", "kl-divergence"):
        tomo.initialize_poisson_ML()
    elif a in ("sart", "asd-pocs"):
        tomo.initialize_SART(initAlg if initAlg else "sequential")
    elif a in ("fbp", "wbp"):
        tomo.initialize_FBP(initAlg)
    tomo.initialize_FP()


def run(tomo, alg, beta=1, niter=1):
    """tomofusion/pytvlib.py:21-31 (plus the 'asd-pocs' -> SART branch the reference lacks, quirk Q8)."""
    a = alg.lower()
    if a in ("sirt", "fista"):
        tomo.SIRT(niter)
    elif a == "cgls":
        tomo.CGLS(niter)
    elif a in ("sart", "asd-pocs"):
        tomo.SART(beta, niter)
    elif a in ("fbp", "wbp"):
        tomo.FBP(True)
    elif a in ("poisson_ml", "kl-divergence"):
        return tomo.poisson_ML(beta)


def wbp_filters():
    return ["ram-lak", "shepp-logan", "hamming", "cosine", "parzen", "lanczos", "triangular", "gaussian",
            "blackman", "nuttall", "blackman-harris", "kaiser"]


def sart_orders():
    return ["sequential", "random"]


def check_hip():
    """Replaces check_cuda (tomofusion/pytvlib.py:42-51): raise unless a HIP device and the library exist."""
    from . import _lib
    if _lib.device_count() == 0:
        raise _lib.TomoError("no HIP device visible")


# ---- ctvlib harness helpers (tomofusion/cpu/utils/pytvlib.py) ---------------------------------------------
def parallelRay(Nside, angles):
    """cpu/utils/pytvlib.py:8-121; angles in degrees; returns float32 (3, nnz) [row, col, val]."""
    return system_matrix(Nside, angles)


def initialize_ctvlib(tomo, alg, Nray, tiltAngles, angleStart=0):
    """cpu/utils/pytvlib.py:178-189: build A; first call loads it, later calls (``angleStart != 0``: tilts were
    appended) swap it in with the reconstruction kept; then the row weights the algorithm needs."""
    A = parallelRay(Nray, np.asarray(tiltAngles))
    if angleStart == 0:
        tomo.load_A(A)
    else:
        tomo.update_proj_angles(A, np.asarray(tiltAngles).shape[0])
    if alg in ("ART", "randART"):
        tomo.row_inner_product()
    elif alg == "cimminoSIRT" and angleStart == 0:
        tomo.cimminos_method()


def run_ctvlib(tomo, alg, beta=1):
    """cpu/utils/pytvlib.py:171-176."""
    if alg in ("SIRT", "cimminoSIRT"):
        tomo.SIRT(beta)
    elif alg == "randART":
        tomo.randART(beta)
    elif alg == "ART":
        tomo.ART(beta)


def create_projections(tomo, original_volume, SNR=0):
    """cpu/utils/pytvlib.py:191-206: with ``SNR != 0`` the background is lifted to 1 (in place, like the reference)
    and Poisson noise at ``SNR`` counts per sample is applied to the projections."""
    if SNR != 0:
        original_volume[original_volume == 0] = 1
    tomo.initialize_original_volume()
    for s in range(original_volume.shape[0]):
        tomo.set_original_volume(original_volume[s], s)
    tomo.create_projections()
    if SNR != 0:
        tomo.poisson_noise(SNR)


def pack_tilt_series(tiltSeries):
    """(Nslice, Nray, Nangles) -> (Nslice, Nray*Nangles) with index angle*Nray+ray
    (gpu/reconstructor.py:54-56, cpu/utils/pytvlib.py:208-213)."""
    ts = np.asarray(tiltSeries)
    return np.ascontiguousarray(ts.transpose(0, 2, 1)).reshape(ts.shape[0], -1)


def load_exp_tilt_series(tomo, tiltSeries):
    tomo.set_tilt_series(pack_tilt_series(tiltSeries))
