"""Fused multi-modal (chemical) tomography on the HIP engine.

* ``multimodal``   -- method table of tomofusion/chemistry/utils/multimodal.cpp:520-564.
* ``ChemicalTomo`` -- tomofusion/chemistry/reconstructor.py:20-249 without the Tk viewer.
* ``create_weighted_summation_weights`` -- the per-element weights of
  tomofusion/chemistry/utils/fusion_helper.py:5-32 (incl. its float16 storage, quirk Q11).

Two C-ABI engines share one device and stream: ``ce`` holds the chemical-map geometry and one tomogram per element
(volume slots ``VOL_USER0 + e``), ``he`` the HAADF geometry, the model volume ``Sigma x^gamma`` and its SIRT/SART
refinement.  Sigma is pixel-diagonal with one weight per element, so ``Sigma x`` is a weighted sum over elements and
no sparse matrix is stored (the reference sizes Sigma with Nslice where Ny is meant, quirk Q15; here it is by pixel).
Everything is slice-independent except the 3-D TV prox, so it shards over ranks exactly like ``tomoengine``.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import (S_COST, S_DD, S_RMSE, S_TV, SINO_B, SINO_G, SINO_R, SINO_USER0, VOL_ORIGINAL, VOL_RECON, VOL_TEMP,
                   VOL_USER0, check)
from .engine import _EngineBase, _f32c, _ptr, tomoengine

PERIODIC_TABLE = ("h he li be b c n o f ne na mg al si p s cl ar k ca sc ti v cr mn fe co ni cu zn ga ge as se br kr rb sr y "
                  "zr nb mo tc ru rh pd ag cd in sn sb te i xe cs ba la ce pr nd pm sm eu gd tb dy ho er tm yb lu hf ta w re "
                  "os ir pt au hg tl pb bi po at rn fr ra ac th pa u np pu am cm bk cf es fm md no lr rf").split()


def get_periodic_table():
    """fusion_helper.py:34-48."""
    return {el: i + 1 for i, el in enumerate(PERIODIC_TABLE)}


def create_weighted_summation_weights(zNums, gamma, method=0):
    """One weight per element = the values ``create_weighted_summation_matrix`` repeats for every pixel
    (fusion_helper.py:5-32); stored as float16 there, hence the rounding."""
    z = np.asarray(zNums, dtype=np.float64)
    if method == 0:
        w = np.ones_like(z)
    elif method == 1:
        w = z / np.mean(z)
    elif method == 2:
        w = z ** gamma / np.mean(z ** gamma)
    elif method == 3:
        w = z / np.sum(z)
    elif method == 4:
        w = z ** gamma / np.sum(z ** gamma)
    else:
        raise ValueError("sigmaMethod must be 0..4")
    return w.astype(np.float16).astype(np.float32)


class multimodal:
    """``multimodal(Nslice, Nray, Nelements, haadfAngles_rad, chemAngles_rad)`` -- multimodal.cpp:48-108."""

    eps = 1e-1  # multimodal.hpp:67
    _engine_cls = tomoengine

    def __init__(self, Nslice, Nray, Nelements, haadfAngles, chemAngles, device=None, comm=None):
        self.Nslice_, self.Ny, self.Nz, self.Nel = int(Nslice), int(Nray), int(Nray), int(Nelements)
        if not 1 <= self.Nel <= 8:
            raise ValueError("1..8 elements supported")
        self.he = self._engine_cls(Nslice, Nray, haadfAngles, device=device, comm=comm)
        self.ce = self._engine_cls(Nslice, Nray, chemAngles, device=device, comm=comm)
        self.ce.be.share_stream_with(self.he.be)
        self.he._stream_peer = self.ce        # a rebuilt HAADF engine (update_projection_angles) rejoins this stream ...
        self.ce._stream_borrowers.append(self.he)   # ... and a rebuilt chemical engine hands the HAADF engine its new one
        self.comm = comm
        self.NprojHaadf, self.NprojChem = self.he.Nproj, self.ce.Nproj
        self.NrowHaadf, self.NrowChem = self.he.Nrow, self.ce.Nrow
        self.gamma_ = 1.0
        self.w = np.ones(self.Nel, np.float32)
        self.measureHaadf_ = self.measureChem_ = False
        self.L_Aps = self.ce.get_lipschitz()          # multimodal.cpp:261: max(BP4D(FP4D(1))) = max(A_c^T A_c 1)
        self.L_ASig = None
        self.projOrder = "sequential"
        # slot maps
        self._x = np.arange(self.Nel, dtype=np.int32) + VOL_USER0                 # tomograms (ce)
        self._u = np.arange(self.Nel, dtype=np.int32) + VOL_USER0 + self.Nel      # BP_C results (ce)
        self._gt = np.arange(self.Nel, dtype=np.int32) + VOL_USER0 + 2 * self.Nel  # ground truth (ce)
        self._b = np.arange(self.Nel, dtype=np.int32) + SINO_USER0                # bChem per element (ce)
        self.MODEL, self.UPD = VOL_USER0, VOL_USER0 + 1                           # he volumes
        for v in self._x:
            self.ce.be.c("copy_volume", int(v), int(v))                           # allocate (zero)

    # ---- flags / parameters (multimodal.cpp:118-148) -------------------------------------------------
    def set_gpu(self, gpu_id):
        self.ce.set_gpu(gpu_id)

    def get_gpu_id(self):
        return self.ce.get_gpu_id()

    get_gpu = get_gpu_id

    def set_measureHaadf(self, flag):
        self.measureHaadf_ = bool(flag)

    def set_measureChem(self, flag):
        self.measureChem_ = bool(flag)

    def measureHaadf(self):
        return self.measureHaadf_

    def measureChem(self):
        return self.measureChem_

    def set_gamma(self, gamma):
        self.gamma_ = float(gamma)

    def gamma(self):
        return self.gamma_

    def initialize_FP(self):
        pass

    def initialize_BP(self):
        pass

    def initialize_SIRT(self):
        pass

    def initialize_SART(self, order="sequential"):
        self.projOrder = order
        self.he.initialize_SART(order)
        self.ce.initialize_SART(order)

    def initialize_initial_volume(self):
        pass

    # ---- data ------------------------------------------------------------------------------------------
    def set_haadf_tilt_series(self, bh):
        self.he.set_tilt_series(bh)

    def set_chem_tilt_series(self, bChem):
        """(Nslice, NrowChem*Nel): per slice the elements' sinograms concatenated (chemistry/reconstructor.py:124-135)."""
        bChem = np.asarray(bChem)
        if bChem.shape != (self.Nslice_, self.NrowChem * self.Nel):
            raise ValueError(f"chemical tilt series must have shape {(self.Nslice_, self.NrowChem * self.Nel)}")
        ce = self.ce
        for e in range(self.Nel):
            loc = _f32c(bChem[ce.first:ce.first + ce.nloc, e * self.NrowChem:(e + 1) * self.NrowChem])
            ce.be.c("set_sinogram", int(self._b[e]), _ptr(loc))

    def load_sigma(self, sigma):
        """(3, nnz) [row, col, val] of create_weighted_summation_matrix: the element of an entry is col // nPix."""
        sigma = np.asarray(sigma)
        npix = int(sigma[0].max()) + 1
        el = (sigma[1].astype(np.int64) // npix)
        w = np.zeros(self.Nel, np.float32)
        for e in range(self.Nel):
            vals = np.unique(sigma[2][el == e])
            if vals.size != 1:
                raise ValueError("summation matrix is not one-weight-per-element")
            w[e] = vals[0]
        self.w = w

    def set_weights(self, w):
        self.w = _f32c(w)

    def set_recon(self, img, element, s):
        self.ce._set_slice(int(self._x[element]), img, s)

    def get_recon(self, element, s):
        return self.ce._get_slice(int(self._x[element]), s)

    def set_original_volume(self, img, element, s):
        self.ce._set_slice(int(self._gt[element]), img, s)

    def get_gt(self, element, s):
        return self.ce._get_slice(int(self._gt[element]), s)

    def set_volume(self, vol4d, which="recon"):
        slots = self._x if which == "recon" else self._gt
        for e in range(self.Nel):
            self.ce.set_volume(vol4d[e], int(slots[e]))

    def get_volume(self, which="recon"):
        slots = self._x if which == "recon" else self._gt
        return np.stack([self.ce.get_volume(int(slots[e])) for e in range(self.Nel)])

    def get_haadf_projections(self):
        return self.he.get_projections()

    def get_model_projections(self):
        return self.he.get_model_projections()

    def get_chem_projections(self):
        return np.concatenate([self.ce._sino(int(self._b[e])) for e in range(self.Nel)], axis=1)

    def restart_recon(self):
        for v in self._x:
            self.ce.be.c("scale_volume", int(v), 0.0)

    # ---- helpers -------------------------------------------------------------------------------------------
    def _mm_model(self):
        self.ce.be.mm_model(self._x, self.w, self.gamma_, self.he.be, self.MODEL)

    def _chem_gradient(self, measure):
        """u_e = BP_C((A x_e - b_e)/(A x_e + eps)); returns the Poisson cost when asked (multimodal.cpp:284-292)."""
        cost = 0.0
        ce = self.ce
        for e in range(self.Nel):
            ce.be.c("poisson_residual", int(self._x[e]), int(self._b[e]), SINO_R)
            if measure:
                cost += ce._scalar(S_COST)
            ce.be.c("back_projection", SINO_R, int(self._u[e]))
        return cost

    # ---- reconstruction ------------------------------------------------------------------------------------
    def estimate_lipschitz(self):
        """multimodal.cpp:259-265.  L_ASig = max(Sigma^T BP_H(FP_H(Sigma 1))) = max(w) * sum(w) * max(A_h^T A_h 1)."""
        self.L_Aps = self.ce.get_lipschitz()
        self.L_ASig = float(self.w.max() * self.w.sum(dtype=np.float32) * np.float32(self.he.get_lipschitz()))

    def poisson_ml(self, lambdaCHEM):
        """multimodal.cpp:277-304."""
        cost = self._chem_gradient(self.measureChem_)
        self.ce.be.mm_update(self._x, self._u, self.w, self.gamma_, float(lambdaCHEM) / self.L_Aps, 0.0, self.he.be,
                             self.UPD, self.MODEL)
        return cost

    def data_fusion(self, lambdaHAADF, lambdaCHEM, nIter=1, method="SIRT"):
        """multimodal.cpp:452-491: returns (costHAADF, costCHEM)."""
        he = self.he
        self._mm_model()
        he.be.c("forward_projection", self.MODEL, SINO_G)                      # g = FP_H(Sigma x^gamma)
        he.be.c("copy_volume", self.UPD, self.MODEL)                           # fuse(): nIter SIRT/SART steps on the model
        if method == "SIRT":
            he.be.c("sirt_data", self.UPD, SINO_B, int(nIter))
        else:
            he.be.c("sart_data", self.UPD, SINO_B, 1.0, int(nIter), None)
        costCHEM = self._chem_gradient(self.measureChem_)
        self.ce.be.mm_update(self._x, self._u, self.w, self.gamma_, float(lambdaCHEM) / self.L_Aps, float(lambdaHAADF),
                             he.be, self.UPD, self.MODEL)
        costHAADF = 0.0
        if self.measureHaadf_:
            he.be.c("sino_diff_norm_sq", SINO_G, SINO_B, S_DD)
            costHAADF = float(np.sqrt(he._scalar(S_DD)))
        return costHAADF, costCHEM

    def sirt_data_fusion(self, lambdaHAADF, lambdaCHEM, nIter):
        return self.data_fusion(lambdaHAADF, lambdaCHEM, nIter, "SIRT")

    def sart_data_fusion(self, lambdaHAADF, lambdaCHEM):
        return self.data_fusion(lambdaHAADF, lambdaCHEM, 1, "SART")

    def chemical_SIRT(self, nIter):
        for e in range(self.Nel):
            self.ce.be.c("sirt_data", int(self._x[e]), int(self._b[e]), int(nIter))

    def chemical_SART(self, nIter):
        for e in range(self.Nel):
            self.ce.be.c("sart_data", int(self._x[e]), int(self._b[e]), 1.0, int(nIter), None)

    def rescale_tomograms(self, scale):
        for v in self._x:
            self.ce.be.c("scale_volume", int(v), float(scale))

    def rescale_projections(self):
        """multimodal.cpp:312-328: per projection, bh <- bh / max(bh) * max(FP_H(Sigma x^gamma))."""
        he = self.he
        self._mm_model()
        he.be.c("forward_projection", self.MODEL, SINO_G)
        mb = np.empty(self.NprojHaadf, np.float32)
        mg = np.empty(self.NprojHaadf, np.float32)
        he.be.c("sino_proj_max", SINO_B, _ptr(mb))
        he.be.c("sino_proj_max", SINO_G, _ptr(mg))
        if self.comm is not None and self.comm.world > 1:
            import torch
            t = torch.from_numpy(np.stack([mb, mg])).to(he.be.halo_tensors()[0].device)
            self.comm.allreduce_max(t)
            mb, mg = t.cpu().numpy()
        mb, mg = _f32c(mb), _f32c(mg)
        he.be.c("sino_proj_scale", SINO_B, _ptr(mb), _ptr(mg))

    def data_distance(self):
        """multimodal.cpp:218-224: ||FP_C(x) - bChem||_F over all elements."""
        tot = 0.0
        for e in range(self.Nel):
            self.ce.be.c("forward_projection", int(self._x[e]), SINO_G)
            self.ce.be.c("sino_diff_norm_sq", SINO_G, int(self._b[e]), S_DD)
            tot += self.ce._scalar(S_DD)
        return float(np.sqrt(tot))

    def tv_fgp_4D(self, ng, lambdaTV):
        """Per-element 3-D FGP prox; returns the summed TV of the inputs (chemistry/.../tv_fgp.cu:192-...)."""
        return sum(self.ce.tv_fgp(int(ng), float(lambdaTV), vol=int(v)) for v in self._x)

    def tv_gd(self, ng, lambdaTV):
        """``tv_gd`` of the reference's table = ``tv_gd_4D`` (multimodal.cpp:494,548): per element ``ng`` normalised TV
        descent steps of length ``lambdaTV`` + positivity; returns the summed TV before descent
        (chemistry/utils/regularizers/tv_gd.cu:208-296)."""
        return sum(self.ce.tv_gd(int(ng), float(lambdaTV), vol=int(v)) for v in self._x)

    def forward_projection(self, inVol):
        """HAADF projection of ONE slice image (Ny*Nz values) -> NrowHaadf values (multimodal.cpp:180-192).  Every
        slice shares the system matrix, so the rank-local slice 0 of a scratch volume carries it; no collective."""
        img = _f32c(np.asarray(inVol).reshape(self.Ny, self.Nz))
        self.he.be.c("set_slice", self.UPD, 0, _ptr(img))
        self.he.be.c("forward_projection", self.UPD, SINO_G)
        return self.he._sino_local(SINO_G)[0].copy()

    def rmse(self):
        out = np.zeros(self.Nel, np.float32)
        for e in range(self.Nel):
            self.ce.be.c("diff_norm_sq", int(self._x[e]), int(self._gt[e]), S_RMSE)
            out[e] = np.sqrt(self.ce._scalar(S_RMSE) / (self.Nslice_ * self.Ny * self.Nz))
        return out


def _one_device_queries(cls):
    """The multi-GPU queries of multigpufusion (multigpufusion.cpp:463-474), answered by the one-device class too:
    ``multigpufusion(...)`` returns ``multimodal`` where one device is left."""
    def get_gpu_ids(self):
        return [int(self.ce.gpuID)]

    def is_multi_gpu_enabled(self):
        return False

    def print_gpu_usage(self):
        print(f"1 GPU (device {int(self.ce.gpuID)}): {self.Nslice_} slices")
    for f in (get_gpu_ids, is_multi_gpu_enabled, print_gpu_usage):
        if f.__name__ not in cls.__dict__:
            setattr(cls, f.__name__, f)
    return cls


_one_device_queries(multimodal)


class multigpufusion(multimodal):
    """``multigpufusion`` (chemistry/utils/multigpufusion.cpp:463-474): the slab-sharded ``multimodal`` -- construct it
    in every rank of a ``torchrun`` job with the GLOBAL sizes.  Unlike the reference (where only ``poisson_ml`` is really
    multi-GPU, quirk Q14, and the fusion step drops a factor, quirk Q12) every method runs on the slabs."""

    def __new__(cls, Nslice, Nray, Nelements, haadfAngles, chemAngles, group=None, devices=None):
        # a plain process spreads the slabs over the visible GPUs by itself, one host thread per device (inprocess.py), like the
        # reference's class (multigpufusion.cpp:140-193); inside a torchrun job every rank constructs its own slab
        from . import inprocess
        if group is None and inprocess.process_group_world() <= 1:
            devs = list(devices) if devices is not None else inprocess.visible_devices()
            devs = devs[:max(1, min(len(devs), int(Nslice)))]        # never more slabs than slices; one device = the plain class
            if len(devs) == 1 and not inprocess.process_group_initialized():
                return multimodal(Nslice, Nray, Nelements, haadfAngles, chemAngles, device=devs[0])
            return inprocess.InProcessMultiGPU(
                lambda comm, dev: multimodal(Nslice, Nray, Nelements, haadfAngles, chemAngles, device=dev, comm=comm), devs)
        return super().__new__(cls)

    def __init__(self, Nslice, Nray, Nelements, haadfAngles, chemAngles, group=None, devices=None):
        from .distributed import SlabComm
        super().__init__(Nslice, Nray, Nelements, haadfAngles, chemAngles, device=None, comm=SlabComm(group))

    def get_gpu_ids(self):
        return self.comm.all_gather_ints(self.ce.gpuID)

    def is_multi_gpu_enabled(self):
        return self.comm.world > 1

    def print_gpu_usage(self):
        if self.comm.rank == 0:
            print(f"{self.comm.world} ranks, one GPU each; slab of rank 0: {self.ce.nloc} of {self.Nslice_} slices")


class ChemicalTomo:
    """tomofusion/chemistry/reconstructor.py:20-249 (drivers only)."""

    def __init__(self, haadf, haadfTiltAngles, chem, chemTiltAngles, gamma=1.6, sigmaMethod=3, gpu_id=-1, comm=None):
        self.nx, self.ny, _ = haadf.shape
        self.elements = list(chem)
        self.nz = len(chem)
        from .reconstructor import determine_gpu_config
        if comm is None and determine_gpu_config(gpu_id) == "multigpu":
            # more than one GPU and no explicit choice: the sharded class, as chemistry/reconstructor.py:44-88 picks it
            self.tomo = multigpufusion(self.nx, self.ny, self.nz, np.deg2rad(haadfTiltAngles), np.deg2rad(chemTiltAngles))
        else:
            self.tomo = multimodal(self.nx, self.ny, self.nz, np.deg2rad(haadfTiltAngles), np.deg2rad(chemTiltAngles),
                                   device=(None if gpu_id < 0 else gpu_id), comm=comm)   # None: the rank's current device
        self.NprojHAADF, self.NprojCHEM = len(haadfTiltAngles), len(chemTiltAngles)
        self.set_haadf_projections(haadf)
        self.set_chemical_projections(chem)
        self.set_summation_matrix(gamma, sigmaMethod)
        self.gamma, self.sigmaMethod, self.reduceLambda = gamma, sigmaMethod, True
        self.tomo.estimate_lipschitz()
        self.tomo.set_measureChem(True)
        self.tomo.set_measureHaadf(True)
        self.reconTotal = None
        self.chemistry_reconstructed = False

    def set_haadf_projections(self, haadf):
        haadf = np.array(haadf, dtype=np.float64)
        haadf[haadf < 0] = 0
        haadf /= np.max(haadf)
        bh = np.ascontiguousarray(haadf.transpose(0, 2, 1)).reshape(self.nx, -1)
        self.tomo.set_haadf_tilt_series(bh)

    def set_chemical_projections(self, chem):
        parts = []
        for el in self.elements:
            c = np.array(chem[el], dtype=np.float64)
            c[c < 0] = 0
            c /= np.max(c)
            parts.append(np.ascontiguousarray(c.transpose(0, 2, 1)).reshape(self.nx, -1))
        self.tomo.set_chem_tilt_series(np.concatenate(parts, axis=1).astype(np.float32))

    def set_summation_matrix(self, gamma=1.6, sigmaMethod=3):
        self.tomo.set_gamma(gamma)
        pt = get_periodic_table()
        zNums = [pt[el.lower()] for el in self.elements]
        self.tomo.set_weights(create_weighted_summation_weights(zNums, 1.6, sigmaMethod))  # 1.6 hard-wired: reconstructor.py:152

    def chemical_tomography(self, Niter=100, lambdaCHEM=0.05):
        self.tomo.restart_recon()
        cost = np.zeros(Niter)
        for i in range(Niter):
            cost[i] = self.tomo.poisson_ml(lambdaCHEM)
        self.chemistry_reconstructed = True
        return cost

    def _rescale_data(self, scale=10):
        self.tomo.rescale_tomograms(scale)
        self.tomo.rescale_projections()

    def data_fusion(self, Niter=50, lambdaCHEM=5e-2, lambdaHAADF=10, lambdaTV=1e-4, iterSIRT=5, tvIter=5,
                    chem_iters=100):
        if not self.chemistry_reconstructed:
            self.chemical_tomography(Niter=chem_iters, lambdaCHEM=lambdaCHEM)
        self._rescale_data()
        costCHEM = np.zeros(Niter, np.float32)
        costHAADF, costTV = costCHEM.copy(), costCHEM.copy()
        for i in range(Niter):
            costHAADF[i], costCHEM[i] = self.tomo.sirt_data_fusion(lambdaHAADF, lambdaCHEM, iterSIRT)
            costTV[i] = self.tomo.tv_fgp_4D(tvIter, lambdaTV)
            if i > 0 and costHAADF[i] > costHAADF[i - 1]:
                lambdaCHEM *= 0.95
        return costHAADF, costCHEM, costTV

    def get_recon(self):
        self.reconTotal = self.tomo.get_volume()
        return self.reconTotal
