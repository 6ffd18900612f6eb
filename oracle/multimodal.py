"""CPU restatement of the reference's fused multi-modal engine -- TEST INFRASTRUCTURE ONLY.

Follows tomofusion/chemistry/utils/multimodal.cpp (poisson_ml :277-304, rescale :307-328, SIRT / SART :339-412, fuse :425-441,
data_fusion :452-491, tv_gd_4D :494, data_distance :218-224) with Sigma from tomofusion/chemistry/utils/fusion_helper.py:5-32.
The reference's FP / BP / SIRT are ASTRA calls (absent, un-pinned): here they are the parallelRay matrix products
and the normalised SIRT of oracle/tomo_oracle.c (orc_forward, orc_back, orc_sirt_norm), so everything that passes
through them is parity-unpinned; the element-wise fusion maths is restated line by line in float32.
"""
import ctypes

import numpy as np

from . import _p, ctvlib, lib, parallel_ray

F = np.float32


class multimodal:
    eps = F(1e-1)  # multimodal.hpp:67

    def __init__(self, Nslice, Nray, Nel, haadf_deg, chem_deg):
        self.Ns, self.N, self.Nel = Nslice, Nray, Nel
        self.H = ctvlib(Nslice, Nray, len(haadf_deg))
        self.H.load_A(parallel_ray(Nray, np.asarray(haadf_deg, np.float64)))
        self.C = ctvlib(Nslice, Nray, len(chem_deg))
        self.C.load_A(parallel_ray(Nray, np.asarray(chem_deg, np.float64)))
        self.recon = np.zeros((Nel, Nslice, Nray, Nray), F)
        self.bh = np.zeros((Nslice, self.H.Nrow), F)
        self.bChem = np.zeros((Nel, Nslice, self.C.Nrow), F)
        self.g = np.zeros_like(self.bh)
        self.w = np.ones(Nel, F)
        self.gamma = F(1.0)
        self.L_Aps = F(self.C.lipschits())

    # Sigma * v = sum_e w_e v_e (Eigen row dot, ascending element)   fusion_helper.py:5-32
    def sigma_apply(self, v4):
        acc = np.zeros(v4.shape[1:], F)
        for e in range(self.Nel):
            acc = acc + self.w[e] * v4[e]
        return acc

    def model(self):
        x = self.recon
        return self.sigma_apply(x if self.gamma == 1 else np.power(x, self.gamma, dtype=F))

    def _fp(self, t, vol):
        out = np.empty((self.Ns, t.Nrow), F)
        v = np.ascontiguousarray(vol, F)
        lib().orc_forward(self.Ns, t.Nrow, t.Ncol, *t._a(), _p(v), _p(out))
        return out

    def _chem_update(self, measure):
        """updateCHEM_e = BP_C((Ax - b)/(Ax + eps)), cost = sum(Ax - b log(Ax + eps))   multimodal.cpp:284-292"""
        upd = np.empty_like(self.recon)
        cost = 0.0
        for e in range(self.Nel):
            Ax = self._fp(self.C, self.recon[e])
            t = ((Ax - self.bChem[e]) / (Ax + self.eps)).astype(F)
            upd[e] = self.C.back_projection(t)
            if measure:
                cost += float((Ax - self.bChem[e] * np.log(Ax + self.eps, dtype=F)).astype(np.float64).sum())
        return upd, cost

    def poisson_ml(self, lamC):
        upd, cost = self._chem_update(True)
        self.recon = np.maximum(self.recon - (F(lamC) / self.L_Aps) * upd, 0).astype(F)
        return cost

    def data_fusion(self, lamH, lamC, nIter, method="SIRT"):
        m = self.model()
        self.g = self._fp(self.H, m)
        # fuse(): nIter normalised SIRT steps (min-constraint 0) started from the model volume, data bh   multimodal.cpp:339-358;
        # sart_data_fusion: ONE SART sweep (run(Nproj * 1), relaxation 1) from the same start            multimodal.cpp:377-396
        self.H.set_tilt_series(self.bh)
        self.H.recon[:] = m
        if method == "SIRT":
            self.H.SIRT_norm(nIter)
        else:
            self.H.SART(1.0, nIter)
        d = (self.H.recon - m).astype(F)
        x = self.recon
        upd, costC = self._chem_update(True)
        new = np.empty_like(x)
        for e in range(self.Nel):
            uh = self.w[e] * d
            if self.gamma != 1:
                uh = (self.gamma * np.power(x[e], self.gamma - F(1.0), dtype=F)) * uh
            new[e] = x[e] - ((F(lamC) / self.L_Aps) * upd[e] - F(lamH) * uh)
        self.recon = np.maximum(new, 0).astype(F)
        costH = float(np.sqrt(((self.g.astype(np.float64) - self.bh) ** 2).sum()))
        return costH, costC

    def rescale_tomograms(self, s):
        self.recon = (self.recon * F(s)).astype(F)

    def rescale_projections(self):
        self.g = self._fp(self.H, self.model())
        N = self.N
        for p in range(self.H.Nproj):
            blk = slice(N * p, N * (p + 1))
            self.bh[:, blk] = (self.bh[:, blk] / self.bh[:, blk].max()) * self.g[:, blk].max()

    def data_distance(self):
        tot = 0.0
        for e in range(self.Nel):
            tot += float(((self._fp(self.C, self.recon[e]).astype(np.float64) - self.bChem[e]) ** 2).sum())
        return float(np.sqrt(tot))

    def chemical_SIRT(self, n):
        for e in range(self.Nel):
            self.C.set_tilt_series(self.bChem[e])
            self.C.recon[:] = self.recon[e]
            self.C.SIRT_norm(n)
            self.recon[e] = self.C.recon

    def chemical_SART(self, n):
        """multimodal.cpp:399-412: per element and slice, run(Nproj * n) single-angle SART updates on bChem_e."""
        for e in range(self.Nel):
            self.C.set_tilt_series(self.bChem[e])
            self.C.recon[:] = self.recon[e]
            self.C.SART(1.0, n)
            self.recon[e] = self.C.recon

    def tv_gd_4D(self, ng, lam, eps=1e-6):
        """multimodal.cpp:494,548 -> chemistry/utils/regularizers/tv_gd.cu:208-296: per element ng normalised descent steps
        of length lam + positivity; the summed TV before descent."""
        tv = 0.0
        self.C.tv_eps = eps
        for e in range(self.Nel):
            self.C.recon[:] = self.recon[e]
            tv += self.C.tv_gd(ng, lam)
            self.recon[e] = self.C.recon
        return tv

    def tv_fgp_4D(self, ng, lam):
        tv = 0.0
        for e in range(self.Nel):
            self.C.recon[:] = self.recon[e]
            tv += self.C.tv_fgp(ng, lam)
            self.recon[e] = self.C.recon
        return tv
