"""The CPU path's helper module under its own names: ``tomofusion/cpu/utils/pytvlib.py:171-213``.

The reference has TWO modules called ``pytvlib``: ``tomofusion/pytvlib.py`` drives the GPU engines
(``initialize_algorithm(tomo, alg, initAlg)``, ``run(tomo, alg, beta, niter)`` -- mirrored by ``tomo_tv_amd.pytvlib``)
and ``tomofusion/cpu/utils/pytvlib.py`` drives ``ctvlib`` (``initialize_algorithm(tomo, alg, Nray, tiltAngles,
angleStart=0)``, ``run(tomo, alg, beta=1)``, ``create_projections(tomo, original_volume, SNR=0)``).  This module is the
second one, signature for signature, so the loops of ``cpu/sim_tomo.py:35-61`` and ``cpu/sim_ASD.py:47-96`` run with

    from tomo_tv_amd.cpu_harness import *
    from tomo_tv_amd.engine import ctvlib

in place of ``from pytvlib import *`` / ``import ctvlib`` (``ctvlib.ctvlib(Nslice, Nray, Nproj)`` -> ``ctvlib(...)``).
"""
from .io import load_data, load_h5_data, mpi_save_results, save_gif, save_recon, save_results  # noqa: F401
from .pytvlib import create_projections, load_exp_tilt_series, parallelRay  # noqa: F401
from .pytvlib import initialize_ctvlib as initialize_algorithm  # noqa: F401
from .pytvlib import run_ctvlib as run  # noqa: F401

__all__ = ["parallelRay", "initialize_algorithm", "run", "create_projections", "load_exp_tilt_series", "load_data",
           "load_h5_data", "save_results", "save_recon", "save_gif", "mpi_save_results"]
