"""CPU parity oracle -- TEST INFRASTRUCTURE ONLY.

ctypes wrapper over ``oracle/tomo_oracle.c`` (a plain-C restatement of the reference's CPU path,
``tomofusion/cpu/utils/ctvlib.cpp`` + ``tomofusion/cpu/utils/pytvlib.py:parallelRay``; every C function
cites the reference lines it follows).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package; the product (``tomo_tv_amd``) never does.

Data layout is the reference's: volumes ``(Nslice, Ny, Nz)`` float32 C-order, sinograms
``(Nslice, Nproj*Nray)`` with index ``angle*Nray + ray`` (``tomofusion/cpu/utils/pytvlib.py:208-213``).

Pinning status: ``parallel_ray`` is pinned bit-for-bit to ``tests/golden/A_*.npz`` (produced by importing the
reference's ``parallelRay`` -- ``tools/gen_golden.py``).  The reference holds no other vectors (it has no
tests), and SART / normalised SIRT run inside the absent ASTRA fork, so those two are parity-unpinned.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libtomo_oracle.so")


def build(force=False):
    """Compile the oracle with gcc (``make -C oracle``)."""
    src = os.path.join(_HERE, "tomo_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None
_SO_TIMED = os.path.join(_HERE, "libtomo_oracle_timed.so")
BUILD_FLAGS = {"parity": "gcc -O2 -ffp-contract=off -fopenmp", "timed": "gcc -O3 -ffast-math -fopenmp (the reference's make.inc flags)"}
_build = "parity"


def select_build(kind):
    """``"parity"`` (default: -O2, no contraction -- what every test uses) or ``"timed"`` (-O3 -ffast-math like
    tomofusion/cpu/utils/make.inc:3 -- ONLY for the timed CPU baseline of bench.py, never for parity)."""
    global _lib, _build
    if kind not in BUILD_FLAGS:
        raise ValueError(kind)
    if kind != _build:
        _build, _lib = kind, None


def lib():
    global _lib
    if _lib is None:
        so = _SO_TIMED if _build == "timed" else _SO
        if not os.path.exists(so):
            build(force=True)
        L = ctypes.CDLL(so)
        i64, i32, f32, f64 = ctypes.c_int64, ctypes.c_int32, ctypes.c_float, ctypes.c_double
        P = ctypes.c_void_p
        L.orc_parallel_ray.restype = i64
        L.orc_parallel_ray.argtypes = [i32, i32, P, P, P, P, i64]
        L.orc_csr_from_coo.restype = i64
        L.orc_csr_from_coo.argtypes = [i64, i64, i64, P, P, P, P, P, P]
        L.orc_forward.restype = None
        L.orc_forward.argtypes = [i32, i64, i64, P, P, P, P, P]
        L.orc_back.restype = None
        L.orc_back.argtypes = [i32, i64, i64, P, P, P, P, P]
        L.orc_lipschitz.restype = f32
        L.orc_lipschitz.argtypes = [i64, i64, P, P, P]
        L.orc_positivity.restype = None
        L.orc_positivity.argtypes = [i64, P]
        L.orc_sirt.restype = None
        L.orc_sirt.argtypes = [i32, i64, i64, P, P, P, P, P, f32, i32]
        L.orc_sirt_cimmino.restype = None
        L.orc_sirt_cimmino.argtypes = [i32, i64, i64, P, P, P, P, P, P, f32, i32]
        L.orc_lipschitz_cimmino.restype = f32
        L.orc_lipschitz_cimmino.argtypes = [i64, i64, P, P, P, P]
        L.orc_row_inner.restype = None
        L.orc_row_inner.argtypes = [i64, P, P, P]
        L.orc_art.restype = None
        L.orc_art.argtypes = [i32, i64, i64, P, P, P, P, P, P, f32, P]
        L.orc_sqdiff.restype = f64
        L.orc_sqdiff.argtypes = [i64, P, P]
        L.orc_sart.restype = None
        L.orc_sart.argtypes = [i32, i32, i32, i64, P, P, P, P, P, f32, i32, P]
        L.orc_sirt_norm.restype = None
        L.orc_sirt_norm.argtypes = [i32, i64, i64, P, P, P, P, P, i32]
        L.orc_tv.restype = f64
        L.orc_tv.argtypes = [i32, i32, i32, P, f32]
        L.orc_tv_gd.restype = f64
        L.orc_tv_gd.argtypes = [i32, i32, i32, P, P, i32, f32, f32]
        L.orc_tv_gd_f64.restype = None
        L.orc_tv_gd_f64.argtypes = [i32, i32, i32, P, P, P, i32, f64, f64]
        L.orc_tv_fgp.restype = f64
        L.orc_tv_fgp.argtypes = [i32, i32, i32, P, P, i32, f32]
        L.orc_fista_momentum.restype = None
        L.orc_fista_momentum.argtypes = [i64, P, P, P, f32]
        L.orc_poisson_ml.restype = f64
        L.orc_poisson_ml.argtypes = [i32, i64, i64, P, P, P, P, P, f32, f32]
        L.orc_num_threads.restype = i32
        L.orc_set_num_threads.restype = None
        L.orc_set_num_threads.argtypes = [i32]
        _lib = L
        if "OMP_NUM_THREADS" not in os.environ:
            L.orc_set_num_threads(min(usable_cpus(), 32))      # parity cases are small; bench.py raises it for its baseline
    return _lib


def usable_cpus():
    """CPUs this process may actually use: affinity mask, capped by a cgroup CPU quota when one is set (an OpenMP team
    larger than that spins against the quota: a 16 x 16 x 2 TV step was seen to take seconds on such a host)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return max(1, n)


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def num_threads():
    return int(lib().orc_num_threads())


def parallel_ray(nside, angles_deg):
    """``parallelRay(Nside, angles)`` (cpu/utils/pytvlib.py:8-121) -> float32 array (3, nnz) [row, col, val]."""
    ang = np.ascontiguousarray(angles_deg, dtype=np.float64)
    P = ang.size
    cap = 2 * nside * P * nside
    rows = np.empty(cap, np.int64)
    cols = np.empty(cap, np.int64)
    vals = np.empty(cap, np.float32)
    nnz = lib().orc_parallel_ray(nside, P, _p(ang), _p(rows), _p(cols), _p(vals), cap)
    if nnz < 0:
        raise RuntimeError("orc_parallel_ray overflow")
    return np.array([rows[:nnz].astype(np.float32), cols[:nnz].astype(np.float32), vals[:nnz]],
                    dtype=np.float32, order="C")


class CSR:
    """The Eigen RowMajor sparse matrix the reference's ``loadA`` builds (ctvlib.cpp:309-315)."""

    def __init__(self, nrow, ncol, ptr, idx, val):
        self.nrow, self.ncol, self.ptr, self.idx, self.val = nrow, ncol, ptr, idx, val

    @property
    def nnz(self):
        return int(self.ptr[-1])


def load_A(A3, nrow, ncol):
    A3 = np.asarray(A3)
    nnz = A3.shape[1]
    rows = np.ascontiguousarray(A3[0], dtype=np.int64)
    cols = np.ascontiguousarray(A3[1], dtype=np.int64)
    vals = _f32(A3[2])
    ptr = np.empty(nrow + 1, np.int64)
    idx = np.empty(max(nnz, 1), np.int32)
    val = np.empty(max(nnz, 1), np.float32)
    out = lib().orc_csr_from_coo(nrow, ncol, nnz, _p(rows), _p(cols), _p(vals), _p(ptr), _p(idx), _p(val))
    if out < 0:
        raise ValueError("load_A: row/col index out of range")
    return CSR(nrow, ncol, ptr, idx[:out].copy(), val[:out].copy())


class ctvlib:
    """Restatement of the reference's ``ctvlib`` class (method table ctvlib.cpp:486-520) on the oracle.

    Adds the GPU-engine methods that have no CPU counterpart (``tv_fgp``, ``fista_momentum``, ``SART``,
    ``SIRT_norm``, ``poisson_ML``) restated from the CUDA / tomoengine sources.
    """

    def __init__(self, Nslice, Nray, Nproj):
        self.Nslice_, self.Ny, self.Nz, self.Nproj = Nslice, Nray, Nray, Nproj
        self.Nrow, self.Ncol = Nray * Nproj, Nray * Nray
        self.recon = np.zeros((Nslice, Nray, Nray), np.float32)
        self.temp_recon = None
        self.original_volume = None
        self.yk = self.recon_old = None
        self.b = np.zeros((Nslice, self.Nrow), np.float32)
        self.g = np.zeros((Nslice, self.Nrow), np.float32)
        self.A = None
        self.innerProduct = None
        self.tv_eps = 1e-8  # ctvlib.cpp:339,408

    def Nslice(self):
        return self.Nslice_

    def Nray(self):
        return self.Ny

    # -- matrix ------------------------------------------------------------------------------
    def load_A(self, A3):
        self.A = load_A(A3, self.Nrow, self.Ncol)

    def _a(self):
        A = self.A
        return (_p(A.ptr), _p(A.idx), _p(A.val))

    def row_inner_product(self):
        self.innerProduct = np.empty(self.Nrow, np.float32)
        lib().orc_row_inner(self.Nrow, _p(self.A.ptr), _p(self.A.val), _p(self.innerProduct))

    def cimminos_method(self):
        """M = diag(A.row(i).dot(A.row(i)))  (ctvlib.cpp:245-251): SIRT and lipschits take the Cimmino branch."""
        self.M = np.empty(self.Nrow, np.float32)
        lib().orc_row_inner(self.Nrow, _p(self.A.ptr), _p(self.A.val), _p(self.M))

    def lipschits(self):
        if getattr(self, "M", None) is not None:
            return float(lib().orc_lipschitz_cimmino(self.Nrow, self.Ncol, *self._a(), _p(self.M)))
        return float(lib().orc_lipschitz(self.Nrow, self.Ncol, *self._a()))

    def poisson_noise(self, Nc, seed=4321):
        """ctvlib.cpp:118-134 with a seeded numpy generator in place of the unseeded std::default_random_engine
        (quirk Q13); the same formula as the product's ``poisson_noise``."""
        b = self.b.astype(np.float64)
        total = b.sum()
        if total <= 0:
            return
        noisy = np.random.default_rng(seed).poisson(b / total * Nc * b.size).astype(np.float64)
        self.b = (noisy / (Nc * b.size) * total).astype(np.float32)

    # -- data --------------------------------------------------------------------------------
    def set_tilt_series(self, b):
        self.b = _f32(b).reshape(self.Nslice_, self.Nrow).copy()

    def initialize_recon_copy(self):
        self.temp_recon = np.zeros_like(self.recon)

    def initialize_tv_recon(self):
        pass

    def initialize_original_volume(self):
        self.original_volume = np.zeros_like(self.recon)

    def set_original_volume(self, img, s):
        self.original_volume[s] = img

    def set_recon(self, img, s):
        self.recon[s] = img

    def get_recon(self, s):
        return self.recon[s].copy()

    def get_projections(self):
        return self.b.copy()

    def restart_recon(self):
        self.recon[:] = 0
        if self.yk is not None:
            self.yk[:] = 0
            self.recon_old[:] = 0

    def create_projections(self):
        lib().orc_forward(self.Nslice_, self.Nrow, self.Ncol, *self._a(), _p(self.original_volume), _p(self.b))

    def forward_projection(self):
        lib().orc_forward(self.Nslice_, self.Nrow, self.Ncol, *self._a(), _p(self.recon), _p(self.g))

    def back_projection(self, sino):
        sino = _f32(sino).reshape(self.Nslice_, self.Nrow)
        out = np.empty_like(self.recon)
        lib().orc_back(self.Nslice_, self.Nrow, self.Ncol, *self._a(), _p(sino), _p(out))
        return out

    # -- reconstruction ----------------------------------------------------------------------
    def SIRT(self, beta, niter=1):
        if getattr(self, "M", None) is not None:
            lib().orc_sirt_cimmino(self.Nslice_, self.Nrow, self.Ncol, *self._a(), _p(self.M), _p(self.b),
                                   _p(self.recon), beta, niter)
            return
        lib().orc_sirt(self.Nslice_, self.Nrow, self.Ncol, *self._a(), _p(self.b), _p(self.recon), beta, niter)

    def ART(self, beta, order=None):
        o = None if order is None else np.ascontiguousarray(order, dtype=np.int32)
        lib().orc_art(self.Nslice_, self.Nrow, self.Ncol, *self._a(), _p(self.innerProduct), _p(self.b),
                      _p(self.recon), beta, None if o is None else _p(o))

    def SART(self, beta, niter=1, order=None, target="recon"):
        vol = getattr(self, target)
        o = None if order is None else np.ascontiguousarray(order, dtype=np.int32)
        lib().orc_sart(self.Nslice_, self.Ny, self.Nproj, self.Ncol, *self._a(), _p(self.b), _p(vol), beta, niter,
                       None if o is None else _p(o))

    def SIRT_norm(self, niter=1, target="recon"):
        vol = getattr(self, target)
        lib().orc_sirt_norm(self.Nslice_, self.Nrow, self.Ncol, *self._a(), _p(self.b), _p(vol), niter)

    def positivity(self):
        lib().orc_positivity(self.recon.size, _p(self.recon))

    def poisson_ML(self, lam, L):
        return float(lib().orc_poisson_ml(self.Nslice_, self.Nrow, self.Ncol, *self._a(), _p(self.b),
                                          _p(self.recon), lam, L))

    # -- scalars -----------------------------------------------------------------------------
    def copy_recon(self):
        self.temp_recon = self.recon.copy()  # true copy (quirk Q1)

    def matrix_2norm(self):
        return float(np.sqrt(lib().orc_sqdiff(self.recon.size, _p(self.recon), _p(self.temp_recon))))

    def data_distance(self, normalize=True):
        self.forward_projection()
        d = float(np.sqrt(lib().orc_sqdiff(self.g.size, _p(self.g), _p(self.b))))
        return d / self.g.size if normalize else d  # ctvlib.cpp:275 vs tomoengine.cpp:412

    def rmse(self):
        return float(np.sqrt(lib().orc_sqdiff(self.recon.size, _p(self.recon), _p(self.original_volume))
                             / self.recon.size))

    def tv(self):
        return float(lib().orc_tv(self.Nslice_, self.Ny, self.Nz, _p(self.recon), self.tv_eps))

    def original_tv(self):
        return float(lib().orc_tv(self.Nslice_, self.Ny, self.Nz, _p(self.original_volume), self.tv_eps))

    def tv_gd(self, ng, dPOCS):
        scratch = np.empty_like(self.recon)
        return float(lib().orc_tv_gd(self.Nslice_, self.Ny, self.Nz, _p(self.recon), _p(scratch), ng, dPOCS,
                                     self.tv_eps))

    def tv_gd_f64(self, ng, dPOCS, start=None):
        """``tv_gd`` from ``start`` (default: recon) evaluated in binary64 (orc_tv_gd_f64): the exact-arithmetic yardstick of
        an ill-conditioned descent, returned as float32; ``recon`` is left alone."""
        start = _f32(self.recon if start is None else start)
        work = np.empty((2,) + start.shape, np.float64)
        out = np.empty_like(start)
        lib().orc_tv_gd_f64(self.Nslice_, self.Ny, self.Nz, _p(start), _p(work), _p(out), ng, float(dPOCS), float(self.tv_eps))
        return out

    def tv_fgp(self, ng, lam):
        work = np.empty((4,) + self.recon.shape, np.float32)
        return float(lib().orc_tv_fgp(self.Nslice_, self.Ny, self.Nz, _p(self.recon), _p(work), ng, lam))

    # -- FISTA (tomoengine.cpp:350-384) --------------------------------------------------------
    def initialize_fista(self):
        self.yk = self.recon.copy()
        self.recon_old = self.recon.copy()

    def fista_momentum(self, beta):
        lib().orc_fista_momentum(self.recon.size, _p(self.recon), _p(self.yk), _p(self.recon_old), beta)
