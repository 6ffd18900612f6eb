"""``TomoGPU`` -- the public reconstruction API of tomofusion/gpu/reconstructor.py:11-215 on the HIP engine.

Same constructor and driver methods (``sirt``, ``sart``, ``fista``, ``asd_pocs``, ``kl_divergence``, ``get_recon``);
the matplotlib/Tk viewers of the reference are not part of the hot path.  Broken reference drivers are implemented
to the semantics of their canonical loops:
``fista`` is the textbook x_k = prox_TV(SIRT(y_k)) + momentum, ``asd_pocs`` follows examples/sim_ASD.py:66-94.
"""
import numpy as np

from . import _lib, pytvlib
from ._lib import S_DD, S_DIFF2, VOL_RECON, VOL_YK
from .engine import multigpuengine, tomoengine


def device_count():
    """List of visible GPU ids (tomofusion/__init__.py:10-18)."""
    return list(range(_lib.device_count()))


def determine_gpu_config(gpu_id=-1):
    """tomofusion/__init__.py:21-34: 'multigpu' whenever no device was named (``gpu_id < 0``) and there is more than one to use --
    several visible GPUs in a plain process (the engine then spreads the slices over them by itself, one host thread per device:
    ``inprocess.py``) or a ``torch.distributed`` job of several ranks (one slab per rank).  ``TOMO_SINGLE_GPU=1`` keeps a plain
    process on one device."""
    import os
    n = len(device_count())
    if n == 0:
        raise ValueError("An AMD GPU is needed for this package!")
    if gpu_id >= 0:
        return "singleconfig"
    from .inprocess import process_group_world
    if process_group_world() > 1:
        return "multigpu"
    if n > 1 and os.environ.get("TOMO_SINGLE_GPU", "0") != "1":
        return "multigpu"
    return "singleconfig"


def _on_rank_threads(fn):
    """Run a driver loop ON the rank threads of the in-process multi-GPU facade instead of forwarding every engine call of the loop
    through it.  A forwarded call costs a queue hand-off to one host thread per device and back (bench.py `inprocess_facade`: 25 us at
    2 devices, 80 us at 8) and an ASD-POCS iteration makes ~7 of them: 0.55 ms of host time per 2.4-ms step at 8 GPUs.  Every rank
    thread runs the same Python loop on its own slab engine -- exactly what the ranks of a torchrun job do (the scalars every rank
    reads are all-reduced, so all ranks take the same decisions) -- and the facade is crossed ONCE per driver call.  The result
    vectors of rank 0's loop are copied back to this object."""
    import copy
    import functools

    @functools.wraps(fn)
    def driver(self, *args, **kw):
        from .inprocess import InProcessMultiGPU
        t = self.tomo
        if not isinstance(t, InProcessMultiGPU) or getattr(self, "_on_rank", False):
            return fn(self, *args, **kw)
        clones = []
        for eng in t._engines:
            c = copy.copy(self)
            c.tomo, c._on_rank = eng, True
            clones.append(c)
        before = dict(self.__dict__)
        out = t._world.run(lambda r: fn(clones[r], *args, **kw))
        # whatever rank 0's loop set or replaced on its clone (cost, dd_vec, tv_vec, ... -- not the engine, not the marker)
        for name, val in clones[0].__dict__.items():
            if name in ("tomo", "_on_rank"):
                continue
            if name not in before or before[name] is not val:
                setattr(self, name, val)
        return out[0]
    return driver


class TomoGPU:

    def __init__(self, tiltAngles, tiltSeries=None, gpu_id=-1, verbose=False, sub_slabs=1):
        """tiltAngles in degrees, tiltSeries (Nslice, Nray, Nangles) with axis 0 = tilt axis.  ``sub_slabs=K`` (an extension,
        single GPU): run the volume as K sub-slab engines side by side on the GPU (engine.py: ``_GroupBackend``)."""
        pytvlib.check_hip()
        tiltAngles = np.asarray(tiltAngles, dtype=np.float64)
        self.Nslice, self.Nray, self.Nangles = tiltSeries.shape
        if tiltAngles.size != self.Nangles:
            raise ValueError("tiltAngles and tiltSeries disagree on the number of projections")
        config = determine_gpu_config(gpu_id)
        if config == "singleconfig":
            self.tomo = tomoengine(self.Nslice, self.Nray, np.deg2rad(tiltAngles), device=max(gpu_id, 0), sub_slabs=sub_slabs)
        else:
            self.tomo = multigpuengine(self.Nslice, self.Nray, np.deg2rad(tiltAngles))
        self.verbose = verbose
        self.set_tilt_series(tiltSeries)
        self.recon = None
        self.cost = None

    def set_tilt_series(self, tiltSeries):
        self.Nslice, self.Nray, self.Nangles = tiltSeries.shape
        self.recon = None
        self.tomo.set_tilt_series(pytvlib.pack_tilt_series(tiltSeries))

    # ---- drivers (gpu/reconstructor.py:61-192) ---------------------------------------------------------
    @_on_rank_threads
    def _run_iterative(self, alg, Niter, show_convergence=True):
        self.cost = np.zeros(Niter)
        self.tomo.restart_recon()
        for i in range(Niter):
            pytvlib.run(self.tomo, alg)
            if show_convergence:
                self.cost[i] = self.tomo.data_distance()
        return self.cost

    def sart(self, Niter=150, init="sequential", show_convergence=True):
        if init not in pytvlib.sart_orders():
            init = "sequential"
        pytvlib.initialize_algorithm(self.tomo, "SART", init)
        return self._run_iterative("SART", Niter, show_convergence)

    def sirt(self, Niter=150, show_convergence=True):
        pytvlib.initialize_algorithm(self.tomo, "SIRT")
        return self._run_iterative("SIRT", Niter, show_convergence)

    def cgls(self, Niter=100, show_convergence=True):
        pytvlib.initialize_algorithm(self.tomo, "CGLS")
        return self._run_iterative("CGLS", Niter, show_convergence)

    @_on_rank_threads
    def wbp(self, filter="ram-lak"):
        if filter not in pytvlib.wbp_filters():
            filter = "ram-lak"
        pytvlib.initialize_algorithm(self.tomo, "FBP", filter)
        pytvlib.run(self.tomo, "FBP")

    @_on_rank_threads
    def kl_divergence(self, Niter=100, lambda_param=0.1):
        self.tomo.restart_recon()
        pytvlib.initialize_algorithm(self.tomo, "kl-divergence")
        self.cost = np.zeros(Niter)
        for i in range(Niter):
            self.cost[i] = pytvlib.run(self.tomo, "kl-divergence", lambda_param)
        return self.cost

    @_on_rank_threads
    def fista(self, Niter=100, momentum=True, lambda_param=0.1, nTViter=10, show_convergence=True):
        """gpu/reconstructor.py:121-155 with the TV prox actually feeding the iterate (quirk Q6)."""
        t = self.tomo
        pytvlib.initialize_algorithm(t, "fista")
        if not momentum:
            t.remove_momentum()
        self.cost = np.zeros(Niter)
        t0 = 1.0
        for k in range(Niter):
            pytvlib.run(t, "fista")                       # gradient step on yk (or recon without momentum)
            if momentum:
                t.tv_fgp(nTViter, lambda_param, vol=VOL_YK)   # the prox acts on the stepped point, in place ...
                tk = 0.5 * (1 + np.sqrt(1 + 4 * t0 ** 2))
                t.fista_momentum((t0 - 1) / tk)               # ... and its result is what momentum extrapolates
                t0 = tk
            else:
                t.tv_fgp(nTViter, lambda_param)
            if show_convergence:
                self.cost[k] = 0.5 * t.data_distance() ** 2 + lambda_param * t.tv()
                if momentum:
                    t.fista_project_yk()                      # the cost's A r gives the next step's A yk by linearity
        return self.cost

    @_on_rank_threads
    def asd_pocs(self, Niter=100, eps=0.025, beta0=0.25, beta_reduce=0.9985, r_max=0.95, nTViter=10, alpha=0.2,
                 alpha_reduce=0.95, show_convergence=True, normalize_dd=True, init="sequential"):
        """examples/sim_ASD.py:66-94 == demo.ipynb cell 25 (the reference method itself is broken, quirk Q7).

        ``normalize_dd``: divide the data distance by Nslice*Nrow as the CPU path does (ctvlib.cpp:275), which is
        what the default ``eps`` presumes (quirk Q5)."""
        t = self.tomo
        pytvlib.initialize_algorithm(t, "asd-pocs", init)
        t.initialize_recon_copy()
        t.restart_recon()
        beta = beta0
        self.dd_vec, self.tv_vec = np.zeros(Niter), np.zeros(Niter)
        dPOCS = 0.0
        norm = float(t.Nslice_ * t.Nrow) if normalize_dd else 1.0
        t.copy_recon()
        pending = None              # (iteration, its SART step norm or None, the deferred read of its scalars)

        def collect(dPOCS):
            """The scalars of the iteration whose read-back is in flight, and the step-length rule they feed (sim_ASD.py:90-94)."""
            j, dp, get = pending
            if dp is None:
                self.tv_vec[j], dg, dd2, dp2 = get()
                dp = float(np.sqrt(dp2))
            else:
                self.tv_vec[j], dg, dd2 = get()
            self.dd_vec[j] = float(np.sqrt(dd2)) / norm
            if dg > dp * r_max and self.dd_vec[j] > eps:
                dPOCS *= alpha_reduce
            return dPOCS
        for i in range(Niter):
            # sim_ASD.py:68-78: copy_recon; SART; dp = matrix_2norm; copy_recon -- the step norm and the new snapshot
            # (TEMP) come out of the sweep's last back-projection pass; TEMP == recon holds on entry.  Only the first
            # iteration needs dp at once (it sets the TV step length); later ones leave it on the device until the
            # iteration's scalars are read together: one all-reduce and one read-back per iteration -- and that read-back
            # is collected only AFTER the next sweep has been enqueued (the scalars of iteration i steer nothing before the
            # TV steps of iteration i+1), so the device never waits for the host between two iterations.
            if i == 0:
                dp0 = t.SART_tracked(beta)
                dPOCS = dp0 * alpha
            else:
                dp0 = None
                t.SART_tracked(beta, defer=True)
                dPOCS = collect(dPOCS)
            beta *= beta_reduce
            # the residual of the SART result is independent of the TV descent: evaluate it on the snapshot (TEMP)
            # on the engine's second stream while the TV steps run
            t.data_distance_begin()
            # sim_ASD.py:84-88: tv_gd; dg = matrix_2norm (and the copy_recon that opens the next iteration)
            pending = (i, dp0, t.tv_gd_tracked(nTViter, dPOCS, extra=(S_DD,) if i == 0 else (S_DD, S_DIFF2), defer=True))
        if pending is not None:
            collect(dPOCS)
        return self.dd_vec, self.tv_vec

    def get_recon(self):
        """(Nslice, Nray, Nray) float64 like the reference (gpu/reconstructor.py:207-215)."""
        self.recon = self.tomo.get_volume(VOL_RECON).astype(np.float64)
        return self.recon
