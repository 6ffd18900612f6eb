"""Host-side mirror of the reference's native engine classes, over the C ABI of ``libtomo_hip.so``.

* ``tomoengine``     -- method table of tomofusion/gpu/utils/tomoengine.cpp:487-534 (what ``TomoGPU`` and
  ``tomofusion/pytvlib.py`` call).
* ``multigpuengine`` -- tomofusion/gpu/utils/multigpuengine.cpp:385-421; here one process per GPU, each owning a
  contiguous slab of x-slices (``distributed.py``), instead of OpenMP threads over host memory.
* ``ctvlib``         -- method table of tomofusion/cpu/utils/ctvlib.cpp:486-520 (``load_A``, ``SIRT(beta)``, ``ART``,
  ``lipschits`` ...), running on the GPU.

Same method names, argument meaning and return values as the reference.  Scalar-returning methods synchronise.
All arithmetic happens in HIP kernels; there is no CPU path (``_lib.load`` raises without the library).
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import (FIELD_FGP_D, FIELD_FGP_P1, S_COST, S_COUNT, S_DD, S_DIFF, S_DIFF2, S_GNORM, S_GNORM_ALL, S_L1, S_RMSE, S_TV, SINO_B,
                   SINO_G, VOL_ORIGINAL, VOL_RECON, VOL_RECON_OLD, VOL_TEMP, VOL_YK, check)
from .distributed import SlabComm, slab_partition


def _f32c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class _SlabBackend:
    """The per-slab primitives: thin wrappers over the C ABI.  (tests substitute this class to exercise the
    distributed composition on CPU/gloo; the product always uses this one.)"""

    def __init__(self, nslice, nray, nproj, angles_rad=None, A=None, device=0):
        self.L = _lib.load()
        self.nslice, self.nray, self.nproj, self.device = nslice, nray, nproj, device
        h = ctypes.c_void_p()
        if A is not None:
            A = np.asarray(A)
            rows, cols, vals = _f32c(A[0]), _f32c(A[1]), _f32c(A[2])
            check(self.L.tomo_create_from_matrix(nslice, nray, nproj, rows.size, _ptr(rows), _ptr(cols), _ptr(vals),
                                                 device, ctypes.byref(h)))
        else:
            ang = np.ascontiguousarray(angles_rad, dtype=np.float64)
            check(self.L.tomo_create(nslice, nray, nproj, _ptr(ang), device, ctypes.byref(h)))
        self.h = h
        self._scal_t = self._halo_lo = self._halo_hi = None
        self.native = False                 # collectives by the library itself (enable_native_comm)

    def close(self):
        if getattr(self, "h", None):
            self.L.tomo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # generic call: self.c("tv_grad", eps) -> tomo_tv_grad(h, eps)
    def c(self, name, *args):
        check(getattr(self.L, "tomo_" + name)(self.h, *args))

    def scalars(self):
        out = np.zeros(S_COUNT, np.float64)
        self.c("read_scalars", _ptr(out), S_COUNT)
        return out

    def scalars_snapshot(self):
        """Enqueue the read-back of all scalar slots; ``scalars_snapshot_read`` collects it (no host wait in between)."""
        self.c("scalars_snapshot")

    def scalars_snapshot_read(self):
        out = np.zeros(S_COUNT, np.float64)
        self.c("scalars_snapshot_read", _ptr(out), S_COUNT)
        return out

    # ---- torch plumbing for the distributed path (device tensors the collectives operate on) ----------
    def enable_torch(self):
        """Device tensors the collectives operate on, bound into the engine; the engine runs on torch's current stream
        of ITS device (the caller's device selection is left alone)."""
        import torch
        dev = torch.device("cuda", self.device)
        self.tdev = dev
        self.c("set_stream", ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        self._scal_t = torch.zeros(S_COUNT, dtype=torch.float64, device=dev)
        npix = self.nray * self.nray
        z = lambda k=1: torch.zeros(k * npix, dtype=torch.float32, device=dev)  # noqa: E731
        self._halo_lo, self._halo_hi, self._send_lo, self._send_hi = z(), z(), z(), z()
        self._g_lo, self._g_hi = z(), z()          # received gradient planes (one-round TV descent)
        # planes of the fused FGP iteration, the two-slice-deep set (include/tomo_hip.h: tomo_bind_fgp_halo2): lo = [P1(-1), A(-1),
        # P2(-1), P3(-1), P1(-2)], hi = [A, P1, P2, P3](nx), [..](nx+1), and what this slab sends; the one-deep planes are the prefixes
        self._fgp_lo, self._fgp_hi, self._fgp_send_first, self._fgp_send_last = z(5), z(8), z(8), z(5)
        self.c("bind_scalar_buffer", ctypes.c_void_p(self._scal_t.data_ptr()))
        self.c("bind_halo", ctypes.c_void_p(self._halo_lo.data_ptr()), ctypes.c_void_p(self._halo_hi.data_ptr()))
        self.c("bind_fgp_halo2", *(ctypes.c_void_p(t.data_ptr()) for t in
                                   (self._fgp_lo, self._fgp_hi, self._fgp_send_first, self._fgp_send_last)))

    def enable_native_comm(self, comm):
        """A native RCCL communicator for this engine (include/tomo_hip.h: tomo_comm_*): rank 0 makes the id, the process
        group carries it to the others, every rank joins.  From then on the engine's collectives are ncclGroups on its own
        stream.  Collective over ``comm``.  If ANY rank fails to join (no librccl, an init error) every rank falls back to the
        ``torch.distributed`` collectives together -- a mixed group would deadlock -- and says so once on stderr."""
        import sys
        import torch

        def all_ok(ok):
            flag = torch.tensor([ok], dtype=torch.int32, device=self.tdev)
            comm.allreduce_min(flag)
            return int(flag.item()) == 1
        idbuf = (ctypes.c_ubyte * 128)()
        # every rank makes an id (only rank 0's is used): the call that opens librccl, so a rank without it is found BEFORE
        # anybody enters the collective ncclCommInitRank
        ok = int(self.L.tomo_comm_unique_id(idbuf) == 0)
        err = b"" if ok else self.L.tomo_last_error()
        if all_ok(ok):
            t = torch.tensor(list(idbuf), dtype=torch.uint8, device=self.tdev)
            comm.broadcast(t, 0)
            raw = bytes(t.cpu().tolist())
            ok = int(self.L.tomo_comm_init(self.h, ctypes.c_char_p(raw), comm.world, comm.rank) == 0)
            err = b"" if ok else self.L.tomo_last_error()
            if all_ok(ok):
                self.native = True
                return
        self.L.tomo_comm_destroy(self.h)
        self.native = False
        if comm.rank == 0:
            print(f"tomo_tv_amd: native RCCL communicator unavailable ({(err or b'another rank failed').decode()}); "
                  "using torch.distributed collectives", file=sys.stderr, flush=True)

    def comm_scalars(self):
        out = np.zeros(S_COUNT, np.float64)
        self.c("comm_read_scalars", _ptr(out), S_COUNT)
        return out

    def scalar_tensor(self, slot):
        return self._scal_t[slot:slot + 1]

    def scalar_gather(self, slots):
        """The listed slots of the device scalar buffer as one new tensor (one all-reduce, one read-back)."""
        return self._scal_t[list(slots)]

    def tensor(self, values, dtype=None):
        import torch
        return torch.tensor(values, dtype=dtype or torch.float64, device=self.tdev)

    def pack_planes(self, field):
        self.c("halo_pack_both", field, ctypes.c_void_p(self._send_lo.data_ptr()), ctypes.c_void_p(self._send_hi.data_ptr()))
        return self._send_lo, self._send_hi

    def tv_update_planes(self, dPOCS, clamp):
        """TV descent step that also leaves recon's new first / last slice in the send planes (no gather launch)."""
        self.c("tv_update_planes", float(dPOCS), int(clamp), ctypes.c_void_p(self._send_lo.data_ptr()),
               ctypes.c_void_p(self._send_hi.data_ptr()))
        return self._send_lo, self._send_hi

    def halo_tensors(self):
        return self._halo_lo, self._halo_hi

    def tv_grad_planes(self, eps, with_tv):
        """Norm pass of a TV descent step that also leaves the gradient's first / last slice in the send planes."""
        self.c("tv_grad_planes", float(eps), int(with_tv), ctypes.c_void_p(self._send_lo.data_ptr()),
               ctypes.c_void_p(self._send_hi.data_ptr()))
        return self._send_lo, self._send_hi

    def g_halo_tensors(self):
        return self._g_lo, self._g_hi

    def tv_halo_apply(self, dPOCS, clamp):
        """Advance the halo planes by the neighbours' update of those slices (received gradient planes, global norm)."""
        self.c("tv_halo_apply", float(dPOCS), int(clamp), ctypes.c_void_p(self._g_lo.data_ptr()),
               ctypes.c_void_p(self._g_hi.data_ptr()))

    def fgp_planes(self, deep=False):
        """(send_first, send_last, lo, hi) of the fused FGP iteration: the one-deep planes (4 / 1 / 1 / 4: views of the prefixes), or
        with ``deep`` the whole two-slice-deep set (8 / 5 / 5 / 8) a pair of iterations needs."""
        if deep:
            return self._fgp_send_first, self._fgp_send_last, self._fgp_lo, self._fgp_hi
        npix = self.nray * self.nray
        return self._fgp_send_first[:4 * npix], self._fgp_send_last[:npix], self._fgp_lo[:npix], self._fgp_hi[:4 * npix]

    def new_plane(self):
        import torch
        return torch.zeros(self.nray * self.nray, dtype=torch.float32, device=self.tdev)

    def slice_to_tensor(self, vol, s):
        import torch
        img = np.empty((self.nray, self.nray), np.float32)
        self.c("get_slice", vol, s, _ptr(img))
        return torch.from_numpy(img).to(self.tdev)

    def lipschitz(self):
        L = ctypes.c_float(0)
        check(self.L.tomo_lipschitz(self.h, ctypes.byref(L)))
        return float(L.value)

    # ---- multimodal (ChemicalTomo) element-wise steps across two engines (include/tomo_hip.h: tomo_mm_*) ----
    def share_stream_with(self, other):
        st = ctypes.c_void_p()
        check(self.L.tomo_get_stream(self.h, ctypes.byref(st)))
        other.c("set_stream", st)

    def mm_model(self, xvols, w, gamma, he, model_vol):
        check(self.L.tomo_mm_model(self.h, _ptr(xvols), len(xvols), _ptr(w), float(gamma), he.h, int(model_vol)))

    def mm_update(self, xvols, uvols, w, gamma, lamC_over_L, lamH, he, upd_vol, model_vol):
        check(self.L.tomo_mm_update(self.h, _ptr(xvols), _ptr(uvols), len(xvols), _ptr(w), float(gamma),
                                    float(lamC_over_L), float(lamH), he.h, int(upd_vol), int(model_vol)))


class _GroupBackend:
    """K sub-slab engines on ONE device behind the backend interface (``tomoengine(..., sub_slabs=K)``).

    A SART sweep is a chain of 180 dependent launches; two such chains on two streams can fill each other's launch gaps
    and kernel ramps.  Measured at 512^3 x 90 (tools/exp_two_engines_step.py, ASD-POCS step, no event logging): one engine
    27.7-28.2 ms; two UNCOUPLED 256-slice engines on two threads 25.6-26.9 ms; this group of two coupled sub-slabs
    27.3-27.6 ms; four 30.5 ms.  The gain is small and varies from box to box, so nothing selects it by default.  The
    sub-slabs must be SEPARATE allocations (sub-slabs interleaved inside one slab's rows collide in the memory system:
    "sart_streams"), so each is a complete engine: own volumes, own tables, own stream.  Slices only couple in the 3-D TV stencils and in the global sums: halo planes are taken straight from
    the neighbour's volume on the device (``tomo_halo_from``), streams are ordered with events (``tomo_wait_for``), partial
    sums are added on the device where a kernel needs the total (``tomo_scalar_sum_from``: ||grad TV||^2) and on the host
    where Python reads them.  This is the single-device form of the slab sharding of ``distributed.py`` (the reference's
    MPI ring, mpi_ctvlib.cpp:400-422,547); everything is enqueued by ONE Python thread except the sweeps themselves.
    """

    def __init__(self, nslice, nray, nproj, angles_rad=None, A=None, device=0, sub_slabs=2):
        K = int(sub_slabs)
        if not 2 <= K <= 8 or K > nslice:
            raise ValueError("sub_slabs must be 2..8 and at most the number of slices")
        self.nslice, self.nray, self.nproj, self.device = nslice, nray, nproj, device
        self.parts = [slab_partition(nslice, K, k) for k in range(K)]
        self.kids = [_SlabBackend(c, nray, nproj, angles_rad=angles_rad, A=A, device=device) for _, c in self.parts]
        self.L, self.h = self.kids[0].L, None
        self._tv_target = VOL_RECON
        for k, kid in enumerate(self.kids):
            kid.c("set_option", b"tv_gnorm_slot", S_GNORM_ALL)
            kid.c("set_slab_edges", int(k == 0), int(k == K - 1))     # FGP's non-periodic boundary (tv_fgp.cu:57,81)
        self._hs = (ctypes.c_void_p * K)(*[kid.h for kid in self.kids])

    # ---- plumbing ---------------------------------------------------------------------------------------------
    def close(self):
        for kid in getattr(self, "kids", []):
            kid.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def scalars(self):
        return np.sum([kid.scalars() for kid in self.kids], axis=0)

    def lipschitz(self):
        return self.kids[0].lipschitz()

    def enable_torch(self):
        raise _lib.TomoError("sub_slabs and a process group are alternatives: shard over ranks OR over sub-slabs of one GPU")

    def share_stream_with(self, other):
        raise _lib.TomoError("a sub-slab group has one stream per sub-slab: not usable as one side of a multimodal pair")

    def _all(self, name, *args):
        for kid in self.kids:
            kid.c(name, *args)

    def _threads(self, name, *args):
        """The same long call on every sub-slab at once: sub-slabs 1.. on helper threads (ctypes drops the GIL)."""
        import threading
        errs = []

        def run(kid):
            try:
                kid.c(name, *args)
            except BaseException as e:  # noqa: BLE001 -- re-raised on the calling thread
                errs.append(e)
        ths = [threading.Thread(target=run, args=(kid,)) for kid in self.kids[1:]]
        for t in ths:
            t.start()
        run(self.kids[0])
        for t in ths:
            t.join()
        if errs:
            raise errs[0]

    @staticmethod
    def _shift(ptr, nbytes):
        return ctypes.c_void_p((ptr.value if isinstance(ptr, ctypes.c_void_p) else int(ptr)) + nbytes)

    def _rows(self, name, which, ptr, rowbytes):
        for (first, _), kid in zip(self.parts, self.kids):
            kid.c(name, which, self._shift(ptr, first * rowbytes))

    def _owner(self, s):
        for k, (first, cnt) in enumerate(self.parts):
            if first <= s < first + cnt:
                return self.kids[k], s - first
        raise IndexError(f"slice {s} out of range")

    def _fill_halos(self, field):
        """Every sub-slab's halo planes from its ring neighbours' volumes (stream-ordered behind their last write)."""
        K = len(self.kids)
        for k, kid in enumerate(self.kids):
            lo, hi = self.kids[(k - 1) % K], self.kids[(k + 1) % K]
            for o in {id(lo): lo, id(hi): hi}.values():
                check(self.L.tomo_wait_for(kid.h, o.h))
            check(self.L.tomo_halo_from(kid.h, int(field), lo.h, hi.h))
        # ... and the other direction (write after read): a neighbour's NEXT write of its boundary slice -- the clamp of
        # tv_gd(0), the in-place SART sweep after tv() / tv_fgp -- must not overtake the copy that reads it on the reader's stream
        for k, kid in enumerate(self.kids):
            for r in {id(self.kids[(k - 1) % K]): self.kids[(k - 1) % K], id(self.kids[(k + 1) % K]): self.kids[(k + 1) % K]}.values():
                if r is not kid:
                    check(self.L.tomo_wait_for(kid.h, r.h))

    def _sum_gnorm(self):
        """||grad TV||^2 over the sub-slabs, left in S_GNORM_ALL of every sub-slab (the update kernels read it there)."""
        for kid in self.kids:
            for o in self.kids:
                if o is not kid:
                    check(self.L.tomo_wait_for(kid.h, o.h))
            check(self.L.tomo_scalar_sum_from(kid.h, S_GNORM_ALL, self._hs, len(self.kids), S_GNORM))

    # ---- the calls of the backend interface ------------------------------------------------------------------------
    _UNSUPPORTED = {"tv_partial", "tv_grad", "tv_grad_tv", "tv_update", "tv_update_planes", "tv_update_tracked", "halo_pack",
                    "halo_pack_both", "halo_local", "bind_halo", "bind_scalar_buffer", "bind_fgp_halo", "fgp_begin", "fgp_begin_vol",
                    "fgp_obj", "fgp_grad", "fgp_end", "fgp_fused_begin", "fgp_fused_step", "fgp_fused_end", "set_stream",
                    "sino_proj_max", "sino_proj_scale", "art_order", "release_geometry"}

    def c(self, name, *args):
        f = getattr(self, "c_" + name, None)
        if f is not None:
            return f(*args)
        if name in self._UNSUPPORTED:
            raise _lib.TomoError(f"tomo_{name} is a per-slab step form: not offered on a sub-slab group")
        self._all(name, *args)           # slice-independent: the same call on every sub-slab

    def c_set_tilt_series(self, ptr):
        self._rows("set_sinogram", SINO_B, ptr, self.nray * self.nproj * 4)

    def c_set_sinogram(self, which, ptr):
        self._rows("set_sinogram", which, ptr, self.nray * self.nproj * 4)

    def c_get_sinogram(self, which, ptr):
        self._rows("get_sinogram", which, ptr, self.nray * self.nproj * 4)

    def c_set_volume(self, which, ptr):
        self._rows("set_volume", which, ptr, self.nray * self.nray * 4)

    def c_get_volume(self, which, ptr):
        self._rows("get_volume", which, ptr, self.nray * self.nray * 4)

    def c_set_slice(self, vol, s, ptr):
        kid, ls = self._owner(s)
        kid.c("set_slice", vol, ls, ptr)

    def c_get_slice(self, vol, s, ptr):
        kid, ls = self._owner(s)
        kid.c("get_slice", vol, ls, ptr)

    def c_set_option(self, name, value):
        if name == b"tv_gnorm_slot":
            raise _lib.TomoError("tv_gnorm_slot is managed by the group")
        self._all("set_option", name, value)

    # the long dependent chains: all sub-slabs at once
    def c_sart(self, *a):
        self._threads("sart", *a)

    def c_sart_data(self, *a):
        self._threads("sart_data", *a)

    def c_sart_tracked(self, *a):
        self._threads("sart_tracked", *a)

    def c_art(self, *a):
        self._threads("art", *a)

    def c_read_scalars(self, ptr, count):
        v = self.scalars()
        (ctypes.c_double * count).from_address(ptr.value)[:] = list(v[:count])

    # 3-D TV: slices couple across the sub-slabs
    def c_tv_set_target(self, vol):
        self._tv_target = int(vol)
        self._all("tv_set_target", vol)

    def c_tv(self, vol, eps):
        self._fill_halos(vol)
        self._all("tv_partial", vol, eps)

    def _tv_gd(self, ng, dPOCS, eps, track, slot):
        tgt = self._tv_target
        if ng <= 0:
            self.c_tv(tgt, eps)
            self._all("positivity", tgt)
            if track >= 0:
                self._all("diff_norm_sq", tgt, track, slot)
                self._all("copy_volume", track, tgt)
            return
        for g in range(ng):
            self._fill_halos(tgt)
            self._all("tv_grad_tv" if g == 0 else "tv_grad", eps)     # sub-slab sums of g^2 (first pass: of TV too)
            self._sum_gnorm()
            if g == ng - 1 and track >= 0:
                self._all("tv_update_tracked", dPOCS, 1, track, slot)
            else:
                self._all("tv_update", dPOCS, int(g == ng - 1))

    def c_tv_gd(self, ng, dPOCS, eps):
        self._tv_gd(int(ng), dPOCS, eps, -1, 0)

    def c_tv_gd_tracked(self, ng, dPOCS, eps, track, slot):
        self._tv_gd(int(ng), dPOCS, eps, int(track), int(slot))

    def c_tv_fgp(self, iters, lam):
        self.c_tv_fgp_vol(VOL_RECON, iters, lam)

    def c_tv_fgp_vol(self, vol, iters, lam):
        """FGP-TV over the sub-slabs with the Obj / Grad pair (tv_fgp.cu:244-268) and a halo fill before each."""
        self.c_tv(vol, 1e-6)                                           # TV of the input (tv_fgp.cu:231-238)
        self._all("fgp_begin_vol", vol)
        for _ in range(int(iters)):
            self._fill_halos(FIELD_FGP_P1)
            self._all("fgp_obj", lam)
            self._fill_halos(FIELD_FGP_D)
            self._all("fgp_grad", lam)
        self._all("fgp_end", int(iters))


class _EngineBase:
    """Shared implementation; ``comm`` is None for one slab = whole volume."""

    _backend_cls = _SlabBackend

    def _setup(self, Nslice, Nray, Nproj, angles_rad=None, A=None, device=None, comm=None, sub_slabs=1):
        self.Nslice_, self.Ny, self.Nz, self.Nproj = int(Nslice), int(Nray), int(Nray), int(Nproj)
        self.Nrow, self.Ncol = self.Ny * self.Nproj, self.Ny * self.Nz
        self.comm = comm
        self.sub_slabs = int(sub_slabs)
        if self.sub_slabs > 1 and comm is not None:
            raise ValueError("sub_slabs (several slab engines on one GPU) and comm (one slab per rank) are alternatives")
        if comm is not None:
            self.first, self.nloc = slab_partition(self.Nslice_, comm.world, comm.rank)
            if self.nloc == 0:
                raise ValueError("more ranks than slices")
        else:
            self.first, self.nloc = 0, self.Nslice_
        if device is None:
            # with a process group the slab lives on the device the rank selected (torch.cuda.set_device(LOCAL_RANK))
            device = self._current_device() if comm is not None else 0
        self.gpuID = int(device)
        self._ctor_kw = dict(angles_rad=angles_rad, A=A)
        self._options = {}
        self._stream_peer = None          # the engine whose stream this one runs on (multimodal)
        self._stream_borrowers = []       # engines that run on THIS engine's stream
        self._make_backend(angles_rad=angles_rad, A=A)
        self.momentum = False
        self.tv_eps = 1e-6          # tv_gd.cu:29,54 (GPU path); ctvlib facade overrides to 1e-8
        self.projOrder = "sequential"
        self._order_rng = np.random.default_rng(0)
        self.L_A = None
        self.L_Aml = None

    @staticmethod
    def _current_device():
        try:
            import torch
            return torch.cuda.current_device() if torch.cuda.is_available() else 0
        except ImportError:
            return 0

    def _make_backend(self, angles_rad=None, A=None):
        if self.sub_slabs > 1:
            self.be = _GroupBackend(self.nloc, self.Ny, self.Nproj, angles_rad=angles_rad, A=A, device=self.gpuID,
                                    sub_slabs=self.sub_slabs)
            return
        self.be = self._backend_cls(self.nloc, self.Ny, self.Nproj, angles_rad=angles_rad, A=A, device=self.gpuID)
        if self.comm is not None:
            self.be.enable_torch()
            self.be.c("set_slab_edges", int(self.comm.rank == 0), int(self.comm.rank == self.comm.world - 1))
            if getattr(self.comm, "native_ok", lambda: False)() and hasattr(self.be, "enable_native_comm"):
                self.be.enable_native_comm(self.comm)

    # ---- helpers -------------------------------------------------------------------------------------
    def _scalar(self, slot):
        """Global value of a partial-sum slot."""
        return self._scalars((slot,))[0]

    def _scalars(self, slots):
        """Global values of several partial-sum slots: one all-reduce and one read-back for all of them."""
        if self.comm is None:
            v = self.be.scalars()
            return [float(v[k]) for k in slots]
        if self._native():
            v = self.be.comm_scalars()
            return [float(v[k]) for k in slots]
        t = self.be.scalar_gather(slots)
        self.comm.allreduce_sum(t)
        return [float(v) for v in t.tolist()]

    def _scalars_begin(self, slots):
        """``_scalars`` in two halves: this one enqueues the all-reduce / read-back and returns a token at once, so the caller
        can go on enqueueing work (the next SART sweep) before ``_scalars_end(token)`` waits for the values."""
        slots = tuple(slots)
        if self.comm is None:
            if hasattr(self.be, "scalars_snapshot"):
                self.be.scalars_snapshot()
                return ("snapshot", slots)
            return ("values", self._scalars(slots))              # sub-slab group: immediate
        if self._native():
            self.be.c("comm_scalars_snapshot")
            return ("snapshot", slots)
        t = self.be.scalar_gather(slots)
        self.comm.allreduce_sum(t)
        if t.is_cuda:
            import torch
            host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            host.copy_(t, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(t.device))
            return ("torch", host, ev)
        return ("values", [float(v) for v in t.tolist()])

    def _scalars_end(self, token):
        kind = token[0]
        if kind == "snapshot":
            v = self.be.scalars_snapshot_read()
            return [float(v[k]) for k in token[1]]
        if kind == "torch":
            token[2].synchronize()
            return [float(v) for v in token[1].tolist()]
        return list(token[1])

    def _native(self):
        """the engine runs its own collectives (RCCL from the library on its stream)"""
        return self.comm is not None and getattr(self.be, "native", False) and self.use_native_comm

    use_native_comm = True   # False: keep the torch.distributed calls although a native communicator exists (A/B, tests)

    def _exchange(self, field, planes=None):
        """Ring exchange of the field's boundary planes (``planes``: already packed by the producing kernel)."""
        if planes is None and self._native():
            self.be.c("comm_exchange_halo", int(field))
            return
        lo, hi = planes if planes is not None else self.be.pack_planes(field)
        hlo, hhi = self.be.halo_tensors()
        self.comm.exchange_planes(lo, hi, hlo, hhi)

    def _target(self):
        return VOL_YK if self.momentum else VOL_RECON

    def Nslice(self):
        return self.Nslice_

    def Nray(self):
        return self.Ny

    # ---- data ------------------------------------------------------------------------------------------
    def set_tilt_series(self, b):
        """(Nslice, Nray*Nproj), index angle*Nray+ray.  tomoengine.cpp:101."""
        b = np.asarray(b)
        if b.shape != (self.Nslice_, self.Nrow):
            raise ValueError(f"tilt series must have shape {(self.Nslice_, self.Nrow)}, got {b.shape}")
        loc = _f32c(b[self.first:self.first + self.nloc])
        self.be.c("set_tilt_series", _ptr(loc))

    def _owner(self, s):
        if s < 0 or s >= self.Nslice_:
            raise IndexError(f"slice {s} out of range [0, {self.Nslice_})")
        if self.comm is None:
            return 0
        base, rem = divmod(self.Nslice_, self.comm.world)
        return s // (base + 1) if s < rem * (base + 1) else rem + (s - rem * (base + 1)) // base

    def _set_slice(self, vol, img, s):
        img = _f32c(img)
        if img.shape != (self.Ny, self.Nz):
            raise ValueError(f"slice must have shape {(self.Ny, self.Nz)}")
        if self.first <= s < self.first + self.nloc:
            self.be.c("set_slice", vol, s - self.first, _ptr(img))
        else:
            self._owner(s)

    def _get_slice(self, vol, s):
        owner = self._owner(s)
        if self.comm is None:
            img = np.empty((self.Ny, self.Nz), np.float32)
            self.be.c("get_slice", vol, s, _ptr(img))
            return img
        if self.comm.rank == owner:
            t = self.be.slice_to_tensor(vol, s - self.first)
        else:
            t = self.be.new_plane().view(self.Ny, self.Nz)
        self.comm.broadcast(t, owner)
        return t.cpu().numpy()

    def set_recon(self, img, s):
        self._set_slice(VOL_RECON, img, s)

    def get_recon(self, s):
        return self._get_slice(VOL_RECON, s)

    def set_original_volume(self, img, s):
        self._set_slice(VOL_ORIGINAL, img, s)

    def initialize_initial_volume(self):
        pass  # volumes are allocated on first use

    initialize_original_volume = initialize_initial_volume

    def initialize_recon_copy(self):
        pass

    def initialize_tv_recon(self):
        pass

    def set_volume(self, vol, which=VOL_RECON):
        """Bulk form of set_recon / set_original_volume: (Nslice, Ny, Nz)."""
        vol = np.asarray(vol)
        if vol.shape != (self.Nslice_, self.Ny, self.Nz):
            raise ValueError("bad volume shape")
        loc = _f32c(vol[self.first:self.first + self.nloc])
        self.be.c("set_volume", which, _ptr(loc))

    def get_volume_local(self, which=VOL_RECON):
        out = np.empty((self.nloc, self.Ny, self.Nz), np.float32)
        self.be.c("get_volume", which, _ptr(out))
        return out

    def _counts(self):
        return [slab_partition(self.Nslice_, self.comm.world, r)[1] for r in range(self.comm.world)]

    def get_volume(self, which=VOL_RECON, dst=None):
        """(Nslice, Ny, Nz).  Sharded: a collective; ``dst=None`` assembles the volume on every rank, ``dst=r`` on
        rank r only (the others return None)."""
        loc = self.get_volume_local(which)
        if self.comm is None or (self.comm.world == 1 and not getattr(self.comm, "force", False)):
            return loc
        return self.comm.gather_slabs(loc, self._counts(), device=getattr(self.be, "tdev", None), dst=dst)

    def _sino(self, which, dst=None):
        out = np.empty((self.nloc, self.Nrow), np.float32)
        self.be.c("get_sinogram", which, _ptr(out))
        if self.comm is None or (self.comm.world == 1 and not getattr(self.comm, "force", False)):
            return out
        return self.comm.gather_slabs(out, self._counts(), device=getattr(self.be, "tdev", None), dst=dst)

    def _sino_local(self, which):
        out = np.empty((self.nloc, self.Nrow), np.float32)
        self.be.c("get_sinogram", which, _ptr(out))
        return out

    def _set_sino_local(self, loc):
        self.be.c("set_tilt_series", _ptr(_f32c(loc)))

    def get_projections(self):
        return self._sino(SINO_B)

    def get_model_projections(self):
        return self._sino(SINO_G)

    def restart_recon(self):
        self.be.c("restart_recon")

    def copy_recon(self):
        self.be.c("copy_volume", VOL_TEMP, VOL_RECON)

    def create_projections(self):
        """b = A * original_volume.  tomoengine.cpp:109-126 / ctvlib.cpp:101-115."""
        self.be.c("forward_projection", VOL_ORIGINAL, SINO_B)

    def forward_projection(self):
        """g = A * recon.  tomoengine.cpp:416-427."""
        self.be.c("forward_projection", VOL_RECON, SINO_G)

    def back_projection_of_tilt_series(self):
        """recon = A^T b (used by tests / Lipschitz checks)."""
        self.be.c("back_projection", SINO_B, VOL_RECON)

    def positivity(self):
        self.be.c("positivity", VOL_RECON)

    # ---- scalars -----------------------------------------------------------------------------------------
    def matrix_2norm(self):
        self.be.c("diff_norm_sq", VOL_RECON, VOL_TEMP, S_DIFF)
        return float(np.sqrt(self._scalar(S_DIFF)))

    def rmse(self):
        self.be.c("diff_norm_sq", VOL_RECON, VOL_ORIGINAL, S_RMSE)
        return float(np.sqrt(self._scalar(S_RMSE) / (self.Nslice_ * self.Ny * self.Nz)))

    def l1_norm(self):
        self.be.c("l1_norm", VOL_RECON)
        return self._scalar(S_L1)

    def _data_distance_raw(self):
        self.be.c("data_distance_sq", VOL_RECON)
        return float(np.sqrt(self._scalar(S_DD)))

    def data_distance_begin(self, vol=VOL_TEMP):
        """Start ||A vol - b|| on the engine's second stream; ``vol`` must not be written until ``data_distance_end``
        (the tracked TV step, which refreshes TEMP, orders itself behind the evaluation on the device).
        (ASD-POCS: the residual of the SART result, held in the TEMP copy, overlaps the TV descent on recon.)"""
        self.be.c("data_distance_sq_async", vol)

    def data_distance_end(self):
        self.be.c("async_wait")
        return float(np.sqrt(self._scalar(S_DD)))

    # ---- TV ----------------------------------------------------------------------------------------------------
    def _tv_of(self, vol, eps):
        if self.comm is None:
            self.be.c("tv", vol, eps)
        else:
            self._exchange(vol)
            self.be.c("tv_partial", vol, eps)
        return self._scalar(S_TV)

    def tv(self):
        return self._tv_of(VOL_RECON, self.tv_eps)

    def original_tv(self):
        return self._tv_of(VOL_ORIGINAL, self.tv_eps)

    def tv_gd(self, ng, dPOCS, vol=VOL_RECON):
        """ng steps of x -= dPOCS * g/||g|| then clamp on recon (or another volume slot); returns TV before descent
        (tv_gd.cu:141-218)."""
        ng = int(ng)
        if vol != VOL_RECON:
            self.be.c("tv_set_target", int(vol))
            try:
                return self._tv_gd(ng, dPOCS, int(vol))
            finally:
                self.be.c("tv_set_target", VOL_RECON)
        return self._tv_gd(ng, dPOCS, VOL_RECON)

    def _tv_gd(self, ng, dPOCS, vol):
        if self.comm is None:
            self.be.c("tv_gd", ng, float(dPOCS), self.tv_eps)
            return self._scalar(S_TV)
        if self._native() and self.tv_one_round:
            self.be.c("comm_tv_gd", ng, float(dPOCS), self.tv_eps, -1, 0)      # the whole sharded descent: one call
            return self._scalar(S_TV)
        if ng <= 0:
            tv0 = self._tv_of(vol, self.tv_eps)
            self.be.c("positivity", vol)
            return tv0
        if self.tv_one_round and hasattr(self.be, "tv_grad_planes"):
            self._tv_descent_one_round(ng, dPOCS, vol, lambda: self.be.c("tv_update", float(dPOCS), 1))
            return self._scalar(S_TV)
        planes = None
        for g in range(ng):
            self._exchange(vol, planes)
            # the first gradient pass also leaves the slab's share of the TV value "before descent"
            self.be.c("tv_grad_tv" if g == 0 else "tv_grad", self.tv_eps)
            self.comm.allreduce_sum(self.be.scalar_tensor(S_GNORM))   # stays on the device
            if g == ng - 1:
                self.be.c("tv_update", float(dPOCS), 1)
            else:                                                     # the step packs the planes the next exchange sends
                planes = self.be.tv_update_planes(dPOCS, 0)
        return self._scalar(S_TV)

    tv_one_round = True   # sharded TV descent: one communication round per inner iteration (False: exchange + all-reduce)

    def _tv_descent_one_round(self, ng, dPOCS, vol, last_update):
        """``ng`` descent steps on a slab with ONE communication round each (the reference needs two: the slice exchange
        mpi_ctvlib.cpp:400-422 and the norm's all-reduce :455).  The halo planes are exchanged once; after that every norm
        pass also leaves the gradient's first / last slice, those planes travel together with the all-reduce of sum g^2,
        and every rank advances its halo planes itself: halo - (dPOCS g)/||g|| is exactly the neighbour's update of that
        slice (same expression, same operands, same bits)."""
        self._exchange(vol)
        g_lo, g_hi = self.be.g_halo_tensors()
        for g in range(ng):
            first, last = self.be.tv_grad_planes(self.tv_eps, g == 0)   # + the slab's share of sum g^2 (first pass: of TV)
            self.comm.allreduce_with_planes(self.be.scalar_tensor(S_GNORM), first, last, g_lo, g_hi)
            if g == ng - 1:
                last_update()                                          # clamped; tracked form where the caller wants it
            else:
                self.be.c("tv_update", float(dPOCS), 0)                # reads the old halo planes ...
                self.be.tv_halo_apply(dPOCS, 0)                        # ... which then follow the neighbours' slices

    fgp_fused = True   # sharded FGP: one fused kernel + one ring exchange per iteration (False: Obj / Grad pair, two)
    fgp_pair = True    # ... and two iterations per pass and exchange where every slab holds two slices (False: one per pass)

    def _min_slab_slices(self):
        """The thinnest slab of the ring (every rank computes the same number: distributed.slab_partition)."""
        from .distributed import slab_partition
        return min(slab_partition(self.Nslice_, self.comm.world, r)[1] for r in range(self.comm.world))

    def tv_fgp(self, ng, lam, vol=VOL_RECON):
        """FGP-TV prox on recon (tv_fgp.cu:192-281), or on another volume slot; returns TV of the input."""
        ng, lam = int(ng), float(lam)
        if self.comm is None:
            self.be.c("tv_fgp_vol", vol, ng, lam)
            return self._scalar(S_TV)
        tv0 = self._tv_of(vol, 1e-6)
        if self.fgp_fused and ng > 1:
            # iterations 0..ng-2: one fused pass each, which also leaves the planes the next exchange sends; the last
            # iteration only needs D (tv_fgp.cu:272) and P1 of the slice below
            self.be.c("fgp_fused_begin", vol)
            first, last, lo, hi = self.be.fgp_planes()
            xchg = (lambda: self.be.c("comm_fgp_exchange")) if self._native() else (lambda: self.comm.exchange_planes(first, last, lo, hi))
            i = 0
            if self.fgp_pair and self._min_slab_slices() >= 2 and ng > 2:
                # TWO iterations per pass (k_fgp_fused2 on slabs, round 6): P stays on chip between them, the halo is two slices deep
                # and ONE exchange serves both iterations.  Every slab of the ring must hold two slices (all ranks agree: the
                # partition is a function of the global size and the world).
                first2, last2, lo2, hi2 = self.be.fgp_planes(deep=True)
                xchg2 = (lambda: self.be.c("comm_fgp_exchange2")) if self._native() else (lambda: self.comm.exchange_planes(first2, last2, lo2, hi2))
                while i + 2 < ng:
                    xchg2()
                    self.be.c("fgp_fused_step2", lam, int(i == 0))
                    i += 2
            while i + 1 < ng:
                xchg()
                self.be.c("fgp_fused_step", lam, int(i == 0))
                i += 1
            xchg()
            self.be.c("fgp_fused_end", lam)
            return tv0
        self.be.c("fgp_begin_vol", vol)
        for _ in range(ng):
            self._exchange(FIELD_FGP_P1)
            self.be.c("fgp_obj", lam)
            self._exchange(FIELD_FGP_D)
            self.be.c("fgp_grad", lam)
        self.be.c("fgp_end", ng)
        return tv0

    def soft_threshold(self, lam):
        self.be.c("soft_threshold", self._target(), float(lam))
        self.be.c("positivity", self._target())

    # ---- FISTA (tomoengine.cpp:350-384) ---------------------------------------------------------------------------
    def initialize_fista(self):
        self.momentum = True
        self.be.c("copy_volume", VOL_YK, VOL_RECON)
        self.be.c("copy_volume", VOL_RECON_OLD, VOL_RECON)
        self.L_A = self.get_lipschitz()

    def get_lipschitz(self):
        return self.be.lipschitz()

    def remove_momentum(self):
        self.momentum = False

    def fista_momentum(self, beta):
        self.be.c("fista_momentum", float(beta))

    def fista_project_yk(self):
        """After ``data_distance()`` of the iterate: the projection of the extrapolated point by linearity,
        A yk = (1 + beta) A r - beta A r_old, from the two projections the cost evaluations made anyway; the next gradient step
        starts from it instead of projecting yk (include/tomo_hip.h: tomo_fista_project_yk).  A no-op whenever the engine
        cannot prove the pieces are in place; returns whether the projection was formed."""
        done = ctypes.c_int(0)
        self.be.c("fista_project_yk", ctypes.byref(done))
        return bool(done.value)

    def synchronize(self):
        self.be.c("synchronize")

    def set_option(self, name, value):
        """Engine switches (include/tomo_hip.h: tomo_set_option), e.g. ``set_option("sart_fused", 0)``."""
        self.be.c("set_option", name.encode(), int(value))
        self._options[name] = int(value)

    def get_option(self, name):
        """A switch or a fact about the (first sub-slab) engine (include/tomo_hip.h: tomo_get_option)."""
        v = ctypes.c_int(0)
        kid = getattr(self.be, "kids", [self.be])[0]
        check(kid.L.tomo_get_option(kid.h, name.encode(), ctypes.byref(v)))
        return int(v.value)

    # ---- simulation edges shared by every engine class ---------------------------------------------------------
    def poisson_noise(self, Nc, seed=4321):
        """Poisson noise on the tilt series at a mean of ``Nc`` counts per sample, total intensity preserved
        (tomoengine.cpp:471-484, ctvlib.cpp:118-134; the reference draws from an unseeded std::default_random_engine,
        quirk Q13: here a seeded numpy generator on the host -- a simulation edge, not part of the hot path).
        Sharded: every rank draws for its own slab (seed + rank); only the total crosses ranks."""
        b = self._sino_local(SINO_B).astype(np.float64)
        total, count = float(b.sum()), float(b.size)
        if self.comm is not None and self.comm.world > 1:
            t = self.be.tensor([total, count])
            self.comm.allreduce_sum(t)
            total, count = (float(v) for v in t.tolist())
            seed = seed + self.comm.rank
        if total <= 0:
            return
        scaled = b / total * Nc * count
        noisy = np.random.default_rng(seed).poisson(scaled).astype(np.float64)
        self._set_sino_local(noisy / (Nc * count) * total)

    def _rebuild(self, Nproj, angles_rad=None, A=None):
        """New tilt geometry with the reconstruction kept (tomoengine.cpp:128-149, ctvlib.cpp:317-333).

        Failure-atomic: the new backend is built FIRST (the matrix is validated and the tables are built and uploaded while
        the old engine is still whole -- two sets of tables are in HBM for that moment, ~2 x 12 GB at 1024^2 x 120 of 288 GB),
        and only then do the volumes move over (no copy) and the old engine go.  If the build raises (bad A, wrong Nproj,
        out of memory) this engine is exactly what it was before the call.  Streams: an engine that borrows its peer's
        stream (multimodal: the HAADF engine runs on the chemical engine's) rejoins it, and an engine whose stream others
        borrow hands them the new one before the old stream is destroyed."""
        if self.sub_slabs > 1:
            raise _lib.TomoError("changing the tilt geometry of a sub-slab group is not supported: build a new engine")
        old, old_nproj, old_nrow = self.be, self.Nproj, self.Nrow
        self.Nproj = int(Nproj)
        self.Nrow = self.Ny * self.Nproj
        try:
            self._make_backend(angles_rad=angles_rad, A=A)
            for name, value in self._options.items():
                self.be.c("set_option", name.encode(), value)
            if self._stream_peer is not None:
                self._stream_peer.be.share_stream_with(self.be)
            for borrower in self._stream_borrowers:
                self.be.share_stream_with(borrower.be)
            check(self.be.L.tomo_adopt_volumes(self.be.h, old.h))
        except BaseException:
            new = self.be
            self.be, self.Nproj, self.Nrow = old, old_nproj, old_nrow
            if new is not old and new is not None:
                for borrower in self._stream_borrowers:        # whoever was already moved to the new stream goes back
                    old.share_stream_with(borrower.be)
                new.close()
            raise
        old.close()
        if self.L_A is not None:
            self.L_A = self.get_lipschitz()
        if self.L_Aml is not None:
            self.L_Aml = self.get_lipschitz()


class tomoengine(_EngineBase):
    """``tomoengine(Nslice, Nray, angles_rad)`` -- tomofusion/gpu/utils/tomoengine.cpp:48-84."""

    def __init__(self, Nslice, Nray, pyAngles=None, device=None, comm=None, sub_slabs=1):
        """``sub_slabs=K`` (an extension, default 1): run the slab as K sub-slab engines on this GPU, each on its own
        stream, so that their dependent launch chains overlap (``_GroupBackend``: -1...-3 % per ASD-POCS step at K = 2)."""
        ang = np.zeros(1) if pyAngles is None else np.ascontiguousarray(pyAngles, dtype=np.float64).ravel()
        self._setup(Nslice, Nray, ang.size, angles_rad=ang, device=device, comm=comm, sub_slabs=sub_slabs)

    # GPU selection (tomoengine.cpp:87-95): the device is fixed at construction in this build
    def set_gpu(self, gpu_id):
        if int(gpu_id) != self.gpuID:
            raise _lib.TomoError("set_gpu after construction is not supported: pass device= to the constructor")

    def get_gpu_id(self):
        return self.gpuID

    # The queries of the reference's multigpuengine (multigpuengine.cpp:385-421).  ``multigpuengine(...)`` returns THIS class where one
    # device is left (a one-GPU box, Nslice == 1, devices=[d]), so a one-device engine answers them too; the sharded classes override.
    def get_gpu_ids(self):
        return [int(self.gpuID)]

    def is_multi_gpu_enabled(self):
        return False

    def print_gpu_usage(self):
        print(f"1 GPU (device {int(self.gpuID)}): {self.Nslice_} slices")

    # initialisers of the ASTRA objects (tomoengine.cpp:151-254): state only
    def initialize_SIRT(self):
        pass

    def initialize_SART(self, order="sequential"):
        if order not in ("sequential", "random"):
            raise ValueError("SART projection order must be 'sequential' or 'random'")
        self.projOrder = order

    def initialize_FP(self):
        pass

    def initialize_BP(self):
        pass

    def initialize_CGLS(self):
        pass

    def CGLS(self, nIter=1):
        """nIter CGLS steps restarted from the current recon, then positivity (tomoengine.cpp:214-229).
        ASTRA's CGLS is not in the reference tree: this is the textbook CGLS on the parallelRay matrix."""
        self.be.c("cgls", VOL_RECON, int(nIter))

    def initialize_FBP(self, filter_name="ram-lak"):
        from .pytvlib import wbp_filters
        if filter_name not in wbp_filters():
            raise ValueError(f"unknown filter {filter_name!r}")
        self.fbpFilter = filter_name

    def FBP(self, apply_positivity=True):
        """Weighted (filtered) back-projection: recon = pi/Nproj * A^T (h * b)   (tomoengine.cpp:330-347).
        ASTRA's filter construction is not in the reference tree; taps come from ``fbp_filter_taps``."""
        taps = fbp_filter_taps(self.Ny, getattr(self, "fbpFilter", "ram-lak"))
        self.be.c("fbp", _ptr(taps), float(np.pi / self.Nproj), int(bool(apply_positivity)))

    def initialize_poisson_ML(self):
        """tomoengine.cpp:231-246: L = max(A^T A 1); normalise the tilt series by its maximum if > 1."""
        self.L_Aml = self.get_lipschitz()
        b = self._sino_local(SINO_B)
        m = float(b.max())
        if self.comm is not None and self.comm.world > 1:
            t = self.be.tensor([m])
            self.comm.allreduce_max(t)
            m = float(t.item())
        if m > 1:
            self._set_sino_local(b / np.float32(m))

    def SIRT(self, nIter=1):
        """ASTRA SIRT with min-constraint 0 on recon, or on yk under momentum (tomoengine.cpp:189-205)."""
        self.be.c("sirt", self._target(), int(nIter))

    def SART(self, beta=1.0, nIter=1):
        """Nproj*nIter single-angle updates, relaxation beta, min-constraint 0 (tomoengine.cpp:162-179)."""
        order = self._sart_order()
        self.be.c("sart", VOL_RECON, float(beta), int(nIter), _ptr(order) if order is not None else None)

    def _sart_order(self):
        if self.projOrder != "random":
            return None
        order = np.ascontiguousarray(self._order_rng.permutation(self.Nproj), dtype=np.int32)
        if self.comm is not None and self.comm.world > 1:   # every rank must sweep the same order
            import torch
            t = self.be.tensor(order.astype(np.int64), dtype=torch.int64)
            self.comm.broadcast(t, 0)
            order = np.ascontiguousarray(t.cpu().numpy(), dtype=np.int32)
        return order

    def SART_tracked(self, beta=1.0, nIter=1, defer=False):
        """``SART(beta, nIter)`` followed by ``matrix_2norm()`` and ``copy_recon()`` (the three calls after which the
        ASD-POCS loop continues, examples/sim_ASD.py:70-78) with the norm and the snapshot produced by the sweep's last
        back-projection pass.  The snapshot (TEMP) must equal recon's state before the sweep.  Returns the step norm;
        ``defer=True`` leaves its square in scalar slot ``S_DIFF2`` instead (no host synchronisation: read it later
        together with the other scalars of the iteration, ``tv_gd_tracked(..., extra=(S_DIFF2,))``)."""
        order = self._sart_order()
        self.be.c("sart_tracked", VOL_RECON, SINO_B, float(beta), int(nIter), _ptr(order) if order is not None else None,
                  VOL_TEMP, S_DIFF2 if defer else S_DIFF)
        return None if defer else float(np.sqrt(self._scalar(S_DIFF)))

    def tv_gd_tracked(self, ng, dPOCS, extra=(), defer=False):
        """``tv_gd(ng, dPOCS)`` followed by ``matrix_2norm()`` and ``copy_recon()`` in one call (the last descent step
        also forms the norm and refreshes the snapshot).  Returns (TV before descent, step norm) + the raw values of
        the ``extra`` scalar slots, all from ONE all-reduce / read-back (``S_DD`` waits for the asynchronous data
        distance first).  ``defer=True`` returns a zero-argument callable instead that delivers that tuple: the read-back
        is enqueued, the caller may enqueue more work (the next iteration's SART sweep, which these values do not steer)
        and call it afterwards -- no idle device between two iterations."""
        ng = int(ng)
        extra = tuple(extra)

        def finish(v):
            return (v[0], float(np.sqrt(v[1]))) + tuple(v[2:])

        def read():
            if S_DD in extra:
                self.be.c("async_wait")
            if defer:
                token = self._scalars_begin((S_TV, S_DIFF) + extra)
                return lambda: finish(self._scalars_end(token))
            return finish(self._scalars((S_TV, S_DIFF) + extra))
        if self.comm is None:
            self.be.c("tv_gd_tracked", ng, float(dPOCS), self.tv_eps, VOL_TEMP, S_DIFF)
            return read()
        if self._native() and self.tv_one_round:
            self.be.c("comm_tv_gd", ng, float(dPOCS), self.tv_eps, VOL_TEMP, S_DIFF)
            return read()
        if ng <= 0:
            tv0 = self.tv_gd(ng, dPOCS)
            nrm = self.matrix_2norm()
            self.copy_recon()
            out = (tv0, nrm) + tuple(self._scalars(extra)) if extra else (tv0, nrm)
            return (lambda: out) if defer else out
        if self.tv_one_round and hasattr(self.be, "tv_grad_planes"):
            self._tv_descent_one_round(ng, dPOCS, VOL_RECON,
                                       lambda: self.be.c("tv_update_tracked", float(dPOCS), 1, VOL_TEMP, S_DIFF))
            return read()
        planes = None
        for g in range(ng):
            self._exchange(VOL_RECON, planes)
            self.be.c("tv_grad_tv" if g == 0 else "tv_grad", self.tv_eps)
            self.comm.allreduce_sum(self.be.scalar_tensor(S_GNORM))   # stays on the device
            if g == ng - 1:
                self.be.c("tv_update_tracked", float(dPOCS), 1, VOL_TEMP, S_DIFF)
            else:
                planes = self.be.tv_update_planes(dPOCS, 0)
        return read()

    def poisson_ML(self, lam):
        self.be.c("poisson_ml", float(lam))
        return self._scalar(S_COST)

    def update_projection_angles(self, pyAngles):
        """New tilt geometry, reconstruction kept (tomoengine.cpp:128-149).  The tilt series must be set again (its
        shape changed)."""
        ang = np.ascontiguousarray(pyAngles, dtype=np.float64).ravel()
        self._rebuild(ang.size, angles_rad=ang)

    def data_distance(self):
        """||A recon - b||_2, un-normalised (tomoengine.cpp:410-413)."""
        return self._data_distance_raw()


class multigpuengine(tomoengine):
    """Slab-sharded engine with the GLOBAL sizes.

    * In a plain process (no ``torch.distributed`` job) it spreads the slices over every visible GPU by itself, like the
      reference's class does (tomofusion/gpu/utils/multigpuengine.cpp:140-193: an OpenMP team, one thread per GPU): one slab
      engine per device, each driven by its own host thread, composed like the ranks of a job (``inprocess.py``).  ``devices=``
      names the devices (default: all visible).
    * Inside a ``torchrun`` job construct it in every rank: one slab per rank, RCCL over the process group.

    Replaces the host-resident volume, dynamic per-slice scheduling and single-GPU TV of the reference's class by static
    device-resident slabs + RCCL."""

    def __new__(cls, Nslice, Nray, pyAngles=None, group=None, force_collectives=False, devices=None):
        from . import inprocess
        if group is None and not force_collectives and inprocess.process_group_world() <= 1:
            devs = list(devices) if devices is not None else inprocess.visible_devices()
            # never more slabs than slices (the reference's per-slice scheduler simply leaves the surplus devices idle,
            # multigpuengine.cpp:163-193); one device left = the plain single-GPU engine
            devs = devs[:max(1, min(len(devs), int(Nslice)))]
            if len(devs) == 1 and not inprocess.process_group_initialized():
                return tomoengine(Nslice, Nray, pyAngles, device=devs[0])
            return inprocess.InProcessMultiGPU(lambda comm, dev: tomoengine(Nslice, Nray, pyAngles, device=dev, comm=comm), devs)
        return super().__new__(cls)

    def __init__(self, Nslice, Nray, pyAngles=None, group=None, force_collectives=False, devices=None):
        super().__init__(Nslice, Nray, pyAngles, device=None, comm=SlabComm(group, force=force_collectives))

    def get_gpu_ids(self):
        return self.comm.all_gather_ints(self.gpuID)

    def is_multi_gpu_enabled(self):
        return self.comm.world > 1

    def print_gpu_usage(self):
        if self.comm.rank == 0:
            print(f"{self.comm.world} ranks, one GPU each; slab of rank 0: {self.nloc} of {self.Nslice_} slices")


class ctvlib(_EngineBase):
    """``ctvlib(Nslice, Nray, Nproj)`` + ``load_A`` -- tomofusion/cpu/utils/ctvlib.cpp:28-48, 309-315, on the GPU."""

    def __init__(self, Nslice, Nray, Nproj, device=None, comm=None):
        self._ctor = (int(Nslice), int(Nray), int(Nproj), device, comm)
        self.be = None

    def load_A(self, A):
        Nslice, Nray, Nproj, device, comm = self._ctor
        if self.be is not None:
            self.be.close()
        self._setup(Nslice, Nray, Nproj, A=A, device=device, comm=comm)
        self.tv_eps = 1e-8  # ctvlib.cpp:339,408

    def update_proj_angles(self, A, Nproj):
        """New measurement matrix for ``Nproj`` projections, reconstruction kept (ctvlib.cpp:317-333; the dynamic
        harness appends tilts this way, cpu/utils/pytvlib.py:183-184).  The tilt series must be set again."""
        if self.be is None:
            raise _lib.TomoError("update_proj_angles before load_A")
        self._ctor = (self._ctor[0], self._ctor[1], int(Nproj)) + self._ctor[3:]
        self._rebuild(int(Nproj), A=A)

    def row_inner_product(self):
        self.be.c("row_inner_product")

    def cimminos_method(self):
        """Cimmino weights M = diag(|A_i|^2) (ctvlib.cpp:245-251): from now on ``SIRT`` and ``lipschits`` take the
        Cimmino branch.  The reference multiplies by the row norms where Cimmino's method divides (quirk Q10); the
        branch is reproduced as written."""
        self._cimmino = True

    def lipschits(self):
        if getattr(self, "_cimmino", False):
            L = ctypes.c_float(0)
            check(self.be.L.tomo_lipschitz_cimmino(self.be.h, ctypes.byref(L)))
            return float(L.value)
        return self.get_lipschitz()

    def SIRT(self, beta):
        """Landweber (or, after ``cimminos_method``, Cimmino) step + positivity (ctvlib.cpp:205-221)."""
        if getattr(self, "_cimmino", False):
            self.be.c("sirt_cimmino", VOL_RECON, float(beta), 1)
        else:
            self.be.c("sirt_landweber", VOL_RECON, float(beta), 1)

    def ART(self, beta):
        self.be.c("art", float(beta))

    def randART(self, beta, seed=None):
        """Kaczmarz sweep over a seeded random permutation of the rows (the reference's randART, ctvlib.cpp:158-179,
        walks a random sequence with an unseedable generator: quirk Q9).  Returns the permutation used."""
        rng = np.random.default_rng(seed) if seed is not None else self._order_rng
        order = np.ascontiguousarray(rng.permutation(self.Nrow), dtype=np.int32)
        self.be.c("art_order", float(beta), _ptr(order))
        return order

    def data_distance(self):
        """||A recon - b||_2 / (Nslice*Nrow)  (ctvlib.cpp:272-276)."""
        return self._data_distance_raw() / (self.Nslice_ * self.Nrow)

    def tv_gd(self, ng, dPOCS):  # void in the reference (ctvlib.cpp:406); the TV value is returned as a courtesy
        return super().tv_gd(ng, dPOCS)


def fbp_filter_taps(n, name="ram-lak"):
    """Real-space taps h[0..n-1] of the symmetric FBP filter |f| * W(f) (names: tomofusion/pytvlib.py:33-36).

    The band-limited ramp is the Kak-Slaney kernel (h[0] = 1/4, h[odd k] = -1/(pi k)^2, h[even] = 0); a window is
    applied in the frequency domain of a 4n-point periodic extension, f = frequency / Nyquist in [0, 1]."""
    L = 4 * int(2 ** np.ceil(np.log2(max(n, 2))))
    k = np.arange(L)
    k = np.minimum(k, L - k).astype(np.float64)
    h = np.zeros(L)
    h[0] = 0.25
    odd = (k % 2) == 1
    h[odd] = -1.0 / (np.pi * k[odd]) ** 2
    H = np.fft.rfft(h).real
    f = np.arange(H.size) / (H.size - 1.0)
    c = lambda a: sum(ai * np.cos(i * np.pi * f) for i, ai in enumerate(a))  # noqa: E731
    windows = {
        "ram-lak": lambda: np.ones_like(f),
        "shepp-logan": lambda: np.sinc(f / 2),
        "cosine": lambda: np.cos(np.pi * f / 2),
        "hamming": lambda: c([0.54, 0.46]),
        "lanczos": lambda: np.sinc(f),
        "triangular": lambda: 1 - f,
        "gaussian": lambda: np.exp(-0.5 * (f / 0.4) ** 2),
        "blackman": lambda: c([0.42, 0.5, 0.08]),
        "nuttall": lambda: c([0.355768, 0.487396, 0.144232, 0.012604]),
        "blackman-harris": lambda: c([0.35875, 0.48829, 0.14128, 0.01168]),
        "kaiser": lambda: np.i0(8.6 * np.sqrt(np.clip(1 - f * f, 0, 1))) / np.i0(8.6),
        "parzen": lambda: np.where(f <= 0.5, 1 - 6 * f ** 2 * (1 - f), 2 * (1 - f) ** 3),
    }
    taps = np.fft.irfft(H * windows[name](), L)[:n]
    return np.ascontiguousarray(taps, dtype=np.float32)


def system_matrix(Nside, angles_deg):
    """Drop-in for ``parallelRay(Nside, angles)`` (tomofusion/cpu/utils/pytvlib.py:8-121): float32 (3, nnz)."""
    L = _lib.load()
    ang = np.ascontiguousarray(np.asarray(angles_deg, dtype=np.float64) * np.pi / 180)
    nnz = ctypes.c_int64(0)
    check(L.tomo_system_matrix(int(Nside), ang.size, _ptr(ang), 0, None, None, None, ctypes.byref(nnz)))
    A = np.empty((3, nnz.value), np.float32)
    check(L.tomo_system_matrix(int(Nside), ang.size, _ptr(ang), nnz.value, _ptr(A[0]), _ptr(A[1]), _ptr(A[2]),
                               ctypes.byref(nnz)))
    return A
