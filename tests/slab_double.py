"""CPU stand-in for the per-slab C-ABI backend, used ONLY by the gloo tests.

The product's distributed composition (tomo_tv_amd/engine.py: which partial sums are all-reduced, when halo planes
are exchanged, how slabs are partitioned) is backend-agnostic.  This double implements the per-slab primitives with
numpy + the oracle so that composition can be exercised with world_size 2 on a CPU-only box.  It is test
infrastructure: nothing in tomo_tv_amd imports it.
"""
import ctypes

import numpy as np
import torch

import oracle
from tomo_tv_amd._lib import (FIELD_FGP_D, FIELD_FGP_P1, S_COST, S_COUNT, S_DD, S_GNORM, S_TV, SINO_B, SINO_G, VOL_ORIGINAL,
                              VOL_RECON, VOL_TEMP)


def _arr(ptr, shape):
    n = int(np.prod(shape))
    addr = ptr.value if isinstance(ptr, ctypes.c_void_p) else ptr
    buf = (ctypes.c_float * n).from_address(addr)
    return np.frombuffer(buf, dtype=np.float32).reshape(shape)


class OracleSlabBackend:
    def __init__(self, nslice, nray, nproj, angles_rad=None, A=None, device=0):
        self.nslice, self.nray, self.nproj, self.device = nslice, nray, nproj, device
        if A is None:
            A = oracle.parallel_ray(nray, np.asarray(angles_rad) * 180 / np.pi)
        self.t = oracle.ctvlib(nslice, nray, nproj)
        self.t.load_A(A)
        self.vol = {VOL_RECON: self.t.recon}
        self.sino = {}
        self.fgp_target = VOL_RECON
        self.tv_target = VOL_RECON
        self.scal = torch.zeros(S_COUNT, dtype=torch.float64)
        npix = nray * nray
        self.halo_lo, self.halo_hi = torch.zeros(npix), torch.zeros(npix)
        self.first_edge, self.last_edge = 1, 1
        self.fields = {}
        self.L = None
        self.h = None

    def _v(self, vid):
        if vid not in self.vol:
            self.vol[vid] = np.zeros_like(self.t.recon)
        return self.vol[vid]

    def _field(self, f):
        if f in (FIELD_FGP_D, FIELD_FGP_P1):
            return self.fields[f]
        return self._v(f)

    def close(self):
        pass

    def c(self, name, *args):
        return getattr(self, "c_" + name)(*args)

    # ---- torch plumbing ------------------------------------------------------------------------------
    def enable_torch(self):
        self.tdev = "cpu"
        npix = self.nray * self.nray
        # the two-slice-deep set (include/tomo_hip.h: tomo_bind_fgp_halo2); the one-deep planes are the prefixes
        self.fgp_lo, self.fgp_hi = torch.zeros(5 * npix), torch.zeros(8 * npix)
        self.fgp_send_first, self.fgp_send_last = torch.zeros(8 * npix), torch.zeros(5 * npix)

    def scalar_tensor(self, slot):
        return self.scal[slot:slot + 1]

    def scalar_gather(self, slots):
        return self.scal[list(slots)]

    def tensor(self, values, dtype=None):
        return torch.tensor(values, dtype=dtype or torch.float64)

    def fgp_planes(self, deep=False):
        if deep:
            return self.fgp_send_first, self.fgp_send_last, self.fgp_lo, self.fgp_hi
        npix = self.nray * self.nray
        return self.fgp_send_first[:4 * npix], self.fgp_send_last[:npix], self.fgp_lo[:npix], self.fgp_hi[:4 * npix]

    def c_async_wait(self):
        pass

    def c_tv_set_target(self, vid):
        self.tv_target = vid

    def scalars(self):
        return self.scal.numpy().copy()

    def pack_planes(self, field):
        x = self._field(field)
        return torch.from_numpy(x[0].ravel().copy()), torch.from_numpy(x[-1].ravel().copy())

    def tv_update_planes(self, dPOCS, clamp):
        self.c_tv_update(dPOCS, clamp)
        return self.pack_planes(self.tv_target)

    def halo_tensors(self):
        return self.halo_lo, self.halo_hi

    # one-round TV descent (tomo_tv_grad_planes / tomo_tv_halo_apply)
    def tv_grad_planes(self, eps, with_tv):
        if with_tv:
            self.c_tv_grad_tv(eps)
        else:
            self.c_tv_grad(eps)
        return torch.from_numpy(self.g[0].ravel().copy()), torch.from_numpy(self.g[-1].ravel().copy())

    def g_halo_tensors(self):
        if not hasattr(self, "g_lo"):
            self.g_lo, self.g_hi = self.new_plane(), self.new_plane()
        return self.g_lo, self.g_hi

    def tv_halo_apply(self, dPOCS, clamp):
        nrm = np.float32(np.sqrt(float(self.scal[S_GNORM])))
        for h, g in ((self.halo_lo, self.g_lo), (self.halo_hi, self.g_hi)):
            v = h.numpy() - (np.float32(dPOCS) * g.numpy()) / nrm
            if clamp:
                np.maximum(v, 0, out=v)
            h.copy_(torch.from_numpy(v.astype(np.float32)))

    def new_plane(self):
        return torch.zeros(self.nray * self.nray)

    def slice_to_tensor(self, vol, s):
        return torch.from_numpy(self._v(vol)[s].copy())

    # ---- primitives ------------------------------------------------------------------------------------
    def c_set_slab_edges(self, first, last):
        self.first_edge, self.last_edge = first, last

    def c_set_tilt_series(self, ptr):
        self.t.set_tilt_series(_arr(ptr, (self.nslice, self.nray * self.nproj)))

    def c_set_volume(self, vid, ptr):
        self._v(vid)[:] = _arr(ptr, self.t.recon.shape)

    def c_get_volume(self, vid, ptr):
        _arr(ptr, self.t.recon.shape)[:] = self._v(vid)

    def c_copy_volume(self, dst, src):
        self._v(dst)[:] = self._v(src)

    def c_restart_recon(self):
        self.t.recon[:] = 0

    def c_positivity(self, vid):
        np.maximum(self._v(vid), 0, out=self._v(vid))

    def c_sart(self, vid, beta, niter, order):
        assert vid == VOL_RECON and order is None
        self.t.SART(beta, niter)

    def c_sart_tracked(self, vid, sino, beta, niter, order, track, slot):
        assert vid == VOL_RECON and sino == 0 and order is None
        self.t.SART(beta, niter)
        self.c_diff_norm_sq(vid, track, slot)
        self._v(track)[:] = self._v(vid)

    def c_tv_update_tracked(self, dPOCS, clamp, track, slot):
        self.c_tv_update(dPOCS, clamp)
        self.c_diff_norm_sq(VOL_RECON, track, slot)
        self._v(track)[:] = self._v(VOL_RECON)

    def c_data_distance_sq(self, vid):
        self.t.forward_projection()
        d = (self.t.g.astype(np.float64) - self.t.b) if False else (self.t.g - self.t.b)
        self.scal[S_DD] = float((d.astype(np.float64) ** 2).sum())

    def c_data_distance_sq_async(self, vid):
        v = np.ascontiguousarray(self._v(vid))
        g = np.empty_like(self.t.b)
        oracle.lib().orc_forward(self.nslice, self.t.Nrow, self.t.Ncol, *self.t._a(), oracle._p(v), oracle._p(g))
        self.scal[S_DD] = float(((g - self.t.b).astype(np.float64) ** 2).sum())

    def c_diff_norm_sq(self, a, b, slot):
        d = self._v(a) - self._v(b)
        self.scal[slot] = float((d.astype(np.float64) ** 2).sum())

    # ---- stencils on [halo_lo | slab | halo_hi] ---------------------------------------------------------
    def _ext(self, x):
        n = self.nray
        lo = self.halo_lo.numpy().reshape(1, n, n)
        hi = self.halo_hi.numpy().reshape(1, n, n)
        return np.concatenate([lo, x, hi], axis=0).astype(np.float32)

    def c_tv_partial(self, vid, eps):
        e = self._ext(self._v(vid))
        c, ip = e[1:-1], e[2:]
        eps = np.float32(eps)
        t = np.sqrt(eps + (c - ip) ** 2 + (c - np.roll(c, -1, 1)) ** 2 + (c - np.roll(c, -1, 2)) ** 2)
        self.scal[S_TV] = float(t.astype(np.float64).sum())

    def c_tv_grad(self, eps):
        e = self._ext(self._v(self.tv_target))
        eps = np.float32(eps)
        c, ip, im = e[1:-1], e[2:], e[:-2]
        jp = lambda v: np.roll(v, -1, 1)  # noqa: E731
        jm = lambda v: np.roll(v, 1, 1)   # noqa: E731
        kp = lambda v: np.roll(v, -1, 2)  # noqa: E731
        km = lambda v: np.roll(v, 1, 2)   # noqa: E731
        three = np.float32(3.0)
        v1n = three * c - ip - jp(c) - kp(c)
        v1d = np.sqrt(eps + (c - ip) ** 2 + (c - jp(c)) ** 2 + (c - kp(c)) ** 2)
        v2n = c - im
        v2d = np.sqrt(eps + (im - c) ** 2 + (im - jp(im)) ** 2 + (im - kp(im)) ** 2)
        b = jm(c)
        v3n = c - b
        v3d = np.sqrt(eps + (b - jm(ip)) ** 2 + (b - c) ** 2 + (b - kp(b)) ** 2)
        d = km(c)
        v4n = c - d
        v4d = np.sqrt(eps + (d - km(ip)) ** 2 + (d - jp(d)) ** 2 + (d - c) ** 2)
        self.g = (v1n / v1d + v2n / v2d + v3n / v3d + v4n / v4d).astype(np.float32)
        self.scal[S_GNORM] = float((self.g.astype(np.float64) ** 2).sum())

    def c_tv_grad_tv(self, eps):
        self.c_tv_partial(self.tv_target, eps)
        self.c_tv_grad(eps)

    def c_tv_update(self, dPOCS, clamp):
        nrm = np.float32(np.sqrt(float(self.scal[S_GNORM])))
        x = self._v(self.tv_target)
        x -= (np.float32(dPOCS) * self.g) / nrm
        if clamp:
            np.maximum(x, 0, out=x)

    def c_fgp_begin(self):
        self.fgp_target = VOL_RECON
        z = lambda: np.zeros_like(self.t.recon)  # noqa: E731
        self.fields = {FIELD_FGP_D: z(), FIELD_FGP_P1: z(), "P2": z(), "P3": z()}

    def c_fgp_obj(self, lam):
        n = self.nray
        P1, P2, P3 = self.fields[FIELD_FGP_P1], self.fields["P2"], self.fields["P3"]
        lo = np.zeros((1, n, n), np.float32) if self.first_edge else self.halo_lo.numpy().reshape(1, n, n)
        v1 = np.concatenate([lo, P1[:-1]], axis=0)
        v2 = np.zeros_like(P2)
        v2[:, 1:, :] = P2[:, :-1, :]
        v3 = np.zeros_like(P3)
        v3[:, :, 1:] = P3[:, :, :-1]
        d = self._v(self.fgp_target) - np.float32(lam) * (P1 + P2 + P3 - v1 - v2 - v3)
        self.fields[FIELD_FGP_D] = np.maximum(d, 0).astype(np.float32)

    def c_fgp_grad(self, lam):
        n = self.nray
        D = self.fields[FIELD_FGP_D]
        P1, P2, P3 = self.fields[FIELD_FGP_P1], self.fields["P2"], self.fields["P3"]
        multip = np.float32(1.0) / (np.float32(26.0) * np.float32(lam))
        if self.last_edge:
            nxt = np.concatenate([D[1:], D[-1:]], axis=0)      # D - D = 0 on the last global slice
        else:
            nxt = np.concatenate([D[1:], self.halo_hi.numpy().reshape(1, n, n)], axis=0)
        v1 = D - nxt
        v2 = np.zeros_like(D)
        v2[:, :-1, :] = D[:, :-1, :] - D[:, 1:, :]
        v3 = np.zeros_like(D)
        v3[:, :, :-1] = D[:, :, :-1] - D[:, :, 1:]
        a, b, c = P1 + multip * v1, P2 + multip * v2, P3 + multip * v3
        den = a * a + b * b + c * c
        sq = np.where(den > 1, np.float32(1.0) / np.sqrt(np.maximum(den, np.float32(1e-30))), np.float32(1.0)).astype(np.float32)
        self.fields[FIELD_FGP_P1], self.fields["P2"], self.fields["P3"] = a * sq, b * sq, c * sq

    def c_fgp_end(self, iters):
        self._v(self.fgp_target)[:] = self.fields[FIELD_FGP_D]

    # fused step form (include/tomo_hip.h: tomo_fgp_fused_*): D of the slice above is rebuilt from its A, P planes
    def c_fgp_fused_begin(self, vid):
        self.c_fgp_begin_vol(vid)
        n = self.nray
        A = self._v(vid)
        self.fgp_send_first[:n * n] = torch.from_numpy(A[0].ravel().copy())
        if A.shape[0] >= 2:                                  # the deep set: A of slice 1 and of the last slice
            self.fgp_send_first[4 * n * n:5 * n * n] = torch.from_numpy(A[1].ravel().copy())
            self.fgp_send_last[n * n:2 * n * n] = torch.from_numpy(A[-1].ravel().copy())

    def _fgp_obj_with(self, lo_plane):
        keep = self.halo_lo
        self.halo_lo = lo_plane
        self.c_fgp_obj(self._lam)
        self.halo_lo = keep

    def c_fgp_fused_step(self, lam, first_iteration):
        n, F = self.nray, np.float32
        self._lam = lam
        P1, P2, P3 = self.fields[FIELD_FGP_P1], self.fields["P2"], self.fields["P3"]
        # D of the slab (P1 of the slice below from the lo plane; the first iteration knows P = 0 and reads no plane)
        self._fgp_obj_with(torch.zeros(n * n) if first_iteration else self.fgp_lo[:n * n])
        hi = self.fgp_hi[:4 * n * n].numpy().reshape(4, n, n)
        a, p1, p2, p3 = hi[0], hi[1], hi[2], hi[3]
        if first_iteration:
            p1 = p2 = p3 = np.zeros((n, n), F)
        v2 = np.zeros((n, n), F); v2[1:, :] = p2[:-1, :]
        v3 = np.zeros((n, n), F); v3[:, 1:] = p3[:, :-1]
        d_hi = np.maximum(a - F(lam) * (p1 + p2 + p3 - P1[-1] - v2 - v3), 0).astype(F)
        keep = self.halo_hi
        self.halo_hi = torch.from_numpy(d_hi.ravel().copy())
        self.c_fgp_grad(lam)
        self.halo_hi = keep
        P1, P2, P3 = self.fields[FIELD_FGP_P1], self.fields["P2"], self.fields["P3"]
        sf = self.fgp_send_first.numpy().reshape(8, n, n)
        sf[1], sf[2], sf[3] = P1[0], P2[0], P3[0]
        self.fgp_send_last[:n * n] = torch.from_numpy(P1[-1].ravel().copy())

    def c_fgp_fused_step2(self, lam, first_iteration):
        """TWO iterations on the slab extended by the two-slice-deep planes (tomo_fgp_fused_step2 on slabs): the same Obj / Grad
        expressions on the extended block, whose outermost cells are wrong after each iteration and never used."""
        n, F = self.nray, np.float32
        A = self._v(self.fgp_target)
        nx = A.shape[0]
        P = [self.fields[FIELD_FGP_P1], self.fields["P2"], self.fields["P3"]]
        lo = self.fgp_lo.numpy().reshape(5, n, n)
        hi = self.fgp_hi.numpy().reshape(8, n, n)
        zero = np.zeros((1, n, n), F)
        below = 0 if self.first_edge else 2
        above = 0 if self.last_edge else 2
        parts_a, parts_p = [], [[], [], []]
        if below:
            parts_a += [zero, lo[1:2]]                                   # A(-2) is never used; A(-1)
            for k, (m2, m1) in enumerate(((lo[4:5], lo[0:1]), (zero, lo[2:3]), (zero, lo[3:4]))):
                parts_p[k] += [m2, m1] if not first_iteration else [zero, zero]
        parts_a.append(A)
        for k in range(3):
            parts_p[k].append(P[k])
        if above:
            parts_a += [hi[0:1], hi[4:5]]
            for k in range(3):
                parts_p[k] += [hi[1 + k:2 + k], hi[5 + k:6 + k]] if not first_iteration else [zero, zero]
        Ae = np.concatenate(parts_a, axis=0).astype(F)
        Pe = [np.concatenate(parts_p[k], axis=0).astype(F) for k in range(3)]
        keep = (self.vol, self.fields, self.first_edge, self.last_edge, self.fgp_target)
        try:
            # run the block as if it were a whole volume (its own ends are edges: that is what makes the outermost cells wrong)
            self.vol = {VOL_RECON: Ae}
            self.fgp_target = VOL_RECON
            self.fields = {FIELD_FGP_D: np.zeros_like(Ae), FIELD_FGP_P1: Pe[0], "P2": Pe[1], "P3": Pe[2]}
            self.first_edge = self.last_edge = True
            for _ in range(2):
                self.c_fgp_obj(lam)
                self.c_fgp_grad(lam)
            out = [self.fields[FIELD_FGP_P1][below:below + nx], self.fields["P2"][below:below + nx], self.fields["P3"][below:below + nx]]
        finally:
            self.vol, self.fields, self.first_edge, self.last_edge, self.fgp_target = keep
        self.fields[FIELD_FGP_P1], self.fields["P2"], self.fields["P3"] = (np.ascontiguousarray(o) for o in out)
        P1, P2, P3 = out
        sf = self.fgp_send_first.numpy().reshape(8, n, n)
        sf[1], sf[2], sf[3] = P1[0], P2[0], P3[0]
        sf[5], sf[6], sf[7] = P1[1], P2[1], P3[1]
        sl = self.fgp_send_last.numpy().reshape(5, n, n)
        sl[0], sl[2], sl[3], sl[4] = P1[-1], P2[-1], P3[-1], P1[-2]

    def c_fgp_fused_end(self, lam):
        self._lam = lam
        n = self.nray
        self._fgp_obj_with(self.fgp_lo[:n * n])
        self._v(self.fgp_target)[:] = self.fields[FIELD_FGP_D]

    # ---- generic slots + multimodal primitives (tests/test_distributed_gloo.py::test_sharded_multimodal) -----
    def _s(self, sid):
        if sid == SINO_B:
            return self.t.b
        if sid == SINO_G:
            return self.t.g
        if sid not in self.sino:
            self.sino[sid] = np.zeros_like(self.t.b)
        return self.sino[sid]

    def c_set_sinogram(self, sid, ptr):
        self._s(sid)[:] = _arr(ptr, self.t.b.shape)

    def c_get_sinogram(self, sid, ptr):
        _arr(ptr, self.t.b.shape)[:] = self._s(sid)

    def c_forward_projection(self, vid, sid):
        v = np.ascontiguousarray(self._v(vid))
        out = np.empty_like(self.t.b)
        oracle.lib().orc_forward(self.nslice, self.t.Nrow, self.t.Ncol, *self.t._a(), oracle._p(v), oracle._p(out))
        self._s(sid)[:] = out

    def c_back_projection(self, sid, vid):
        self._v(vid)[:] = self.t.back_projection(self._s(sid))

    def c_sirt_data(self, vid, sid, niter):
        keep_b, keep_r = self.t.b, self.t.recon
        self.t.b, self.t.recon = np.ascontiguousarray(self._s(sid)), np.ascontiguousarray(self._v(vid).copy())
        self.t.SIRT_norm(niter)
        self._v(vid)[:] = self.t.recon
        self.t.b, self.t.recon = keep_b, keep_r

    def c_poisson_residual(self, vid, sid_b, sid_out):
        self.c_forward_projection(vid, 9999)
        ax, b = self.sino[9999], self._s(sid_b)
        eps = np.float32(0.1)
        self._s(sid_out)[:] = (ax - b) / (ax + eps)
        self.scal[S_COST] = float((ax - b * np.log(ax + eps, dtype=np.float32)).astype(np.float64).sum())

    def c_scale_volume(self, vid, f):
        self._v(vid)[:] = self._v(vid) * np.float32(f)

    def c_sino_diff_norm_sq(self, a, b, slot):
        d = self._s(a).astype(np.float64) - self._s(b)
        self.scal[slot] = float((d ** 2).sum())

    def c_sino_proj_max(self, sid, ptr):
        g = self._s(sid).reshape(self.nslice, self.nproj, self.nray)
        _arr(ptr, (self.nproj,))[:] = g.max(axis=(0, 2))

    def c_sino_proj_scale(self, sid, pdiv, pmul):
        g = self._s(sid).reshape(self.nslice, self.nproj, self.nray)
        g[:] = (g / _arr(pdiv, (self.nproj,))[None, :, None]) * _arr(pmul, (self.nproj,))[None, :, None]

    def c_fgp_begin_vol(self, vid):
        self.fgp_target = vid
        z = lambda: np.zeros_like(self.t.recon)  # noqa: E731
        self.fields = {FIELD_FGP_D: z(), FIELD_FGP_P1: z(), "P2": z(), "P3": z()}

    def share_stream_with(self, other):
        pass

    def lipschitz(self):
        return self.t.lipschits()

    def mm_model(self, xvols, w, gamma, he, model_vol):
        acc = np.zeros_like(self.t.recon)
        for e, v in enumerate(xvols):
            x = self._v(int(v))
            acc = acc + np.float32(w[e]) * (x if gamma == 1 else np.power(x, np.float32(gamma), dtype=np.float32))
        he._v(model_vol)[:] = acc

    def mm_update(self, xvols, uvols, w, gamma, lamC_over_L, lamH, he, upd_vol, model_vol):
        F = np.float32
        d = (he._v(upd_vol) - he._v(model_vol)) if lamH != 0 else 0
        for e, (v, u) in enumerate(zip(xvols, uvols)):
            x, uc = self._v(int(v)), self._v(int(u))
            uh = F(w[e]) * d
            if gamma != 1:
                uh = (F(gamma) * np.power(x, F(gamma) - F(1), dtype=F)) * uh
            x[:] = np.maximum(x - (F(lamC_over_L) * uc - F(lamH) * uh), 0)

