"""Fused multi-modal engine (config 5 path) against the oracle restatement of multimodal.cpp, through the C ABI."""
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, rel_l2
from oracle.multimodal import multimodal as ref_multimodal
from tomo_tv_amd.chemistry import ChemicalTomo, create_weighted_summation_weights, multimodal
from tomo_tv_amd.phantom import ellipsoids

pytestmark = pytest.mark.gpu
TOL = 1e-5


def make_case(N=32, Nx=6, Ph=9, Pc=7, Nel=2, gamma=1.6, seed=0):
    ha, ca = np.linspace(-70, 70, Ph), np.linspace(-64, 68, Pc)
    gt = np.stack([ellipsoids(Nx, N, seed=seed + e, k=8) * (0.6 + 0.3 * e) for e in range(Nel)])
    w = create_weighted_summation_weights([31, 8, 22][:Nel], 1.6, 3)
    ref = ref_multimodal(Nx, N, Nel, ha, ca)
    ref.w, ref.gamma = w.copy(), np.float32(gamma)
    # synthetic measurements from the oracle's own operators
    for e in range(Nel):
        ref.bChem[e] = ref._fp(ref.C, gt[e])
    ref.recon = gt.copy()
    ref.bh = ref._fp(ref.H, ref.model())
    ref.bh /= ref.bh.max()
    ref.bChem /= ref.bChem.max()
    dev = multimodal(Nx, N, Nel, np.deg2rad(ha), np.deg2rad(ca))
    dev.set_gamma(gamma)
    dev.set_weights(w)
    dev.set_haadf_tilt_series(ref.bh)
    dev.set_chem_tilt_series(np.concatenate([ref.bChem[e] for e in range(Nel)], axis=1))
    dev.set_measureChem(True)
    dev.set_measureHaadf(True)
    dev.estimate_lipschitz()
    ref.recon = np.zeros_like(gt)
    return dev, ref, gt


@pytest.mark.parametrize("gamma", [1.0, 1.6])
def test_poisson_ml_then_data_fusion(gpu, gamma):
    dev, ref, gt = make_case(gamma=gamma)
    assert abs(dev.L_Aps - float(ref.L_Aps)) <= 1e-6 * dev.L_Aps
    for it in range(5):
        c_dev, c_ref = dev.poisson_ml(0.05), ref.poisson_ml(0.05)
        assert abs(c_dev - c_ref) <= 2e-5 * abs(c_ref), it
    assert rel_l2(dev.get_volume(), ref.recon) < TOL
    assert abs(dev.data_distance() - ref.data_distance()) <= 1e-5 * ref.data_distance()
    dev.rescale_tomograms(10)
    ref.rescale_tomograms(10)
    dev.rescale_projections()
    ref.rescale_projections()
    assert rel_l2(dev.get_haadf_projections(), ref.bh) < TOL
    for it in range(3):
        (h_dev, c_dev), (h_ref, c_ref) = dev.sirt_data_fusion(10, 0.05, 3), ref.data_fusion(10, 0.05, 3)
        assert abs(h_dev - h_ref) <= 2e-5 * h_ref and abs(c_dev - c_ref) <= 2e-5 * abs(c_ref), it
        tv_dev, tv_ref = dev.tv_fgp_4D(3, 1e-4), ref.tv_fgp_4D(3, 1e-4)
        assert abs(tv_dev - tv_ref) <= 2e-5 * tv_ref
        assert rel_l2(dev.get_volume(), ref.recon) < 2e-5, it
    assert rel_l2(dev.get_model_projections(), ref.g) < TOL


def test_chemical_sirt_and_accessors(gpu):
    dev, ref, gt = make_case(Nel=3, Pc=9, Ph=9)
    dev.chemical_SIRT(4)
    ref.chemical_SIRT(4)
    assert rel_l2(dev.get_volume(), ref.recon) < TOL
    dev.set_volume(gt, which="gt")
    want = np.sqrt(((ref.recon - gt).astype(np.float64) ** 2).mean(axis=(1, 2, 3)))
    assert np.allclose(dev.rmse(), want, rtol=1e-5)
    assert np.array_equal(dev.get_recon(1, 2), dev.get_volume()[1, 2])
    assert rel_l2(dev.get_chem_projections(), np.concatenate(list(ref.bChem), axis=1)) < 1e-7
    dev.restart_recon()
    assert dev.get_volume().max() == 0
    with pytest.raises(ValueError):
        dev.set_chem_tilt_series(np.zeros((2, 3)))


def test_sart_data_fusion_chemical_sart_and_tv_gd_4d(gpu):
    """The SART branches of the fused engine and its 4-D TV descent against the oracle (multimodal.cpp:377-412 SART /
    chemical_SART, :452-491 data_fusion with the SART fuse step, :494,548 tv_gd_4D)."""
    dev, ref, gt = make_case(N=32, Nx=6, Ph=9, Pc=7, Nel=2, gamma=1.6)
    dev.chemical_SART(2)
    ref.chemical_SART(2)
    assert rel_l2(dev.get_volume(), ref.recon) < TOL
    dev.rescale_projections()
    ref.rescale_projections()
    for it in range(3):
        (h_dev, c_dev), (h_ref, c_ref) = dev.sart_data_fusion(10, 0.05), ref.data_fusion(10, 0.05, 1, method="SART")
        assert abs(h_dev - h_ref) <= 2e-5 * h_ref and abs(c_dev - c_ref) <= 2e-5 * abs(c_ref), it
        assert rel_l2(dev.get_volume(), ref.recon) < 2e-5, it
    tv_dev, tv_ref = dev.tv_gd(3, 0.05), ref.tv_gd_4D(3, 0.05, eps=dev.ce.tv_eps)
    assert abs(tv_dev - tv_ref) <= 2e-5 * tv_ref
    assert rel_l2(dev.get_volume(), ref.recon) < 2e-5
    assert dev.get_volume().min() >= 0


def test_model_of_a_start_volume_with_negative_and_zero_voxels(gpu):
    """Sigma x^gamma for a caller-supplied volume with negative voxels and an integer gamma is what numpy's ** gives
    (multimodal.cpp:425-427 uses Eigen's pow), not NaN; 0^gamma = 0."""
    dev, ref, gt = make_case(gamma=2.0)
    x = (gt - np.float32(0.2)).astype(np.float32)
    x[:, :, :4] = 0
    dev.set_volume(x)
    dev._mm_model()
    got = dev.he.get_volume(dev.MODEL)
    want = sum(np.float32(dev.w[e]) * x[e] ** 2 for e in range(x.shape[0]))
    assert np.isfinite(got).all() and (x < 0).any()
    assert rel_l2(got, want) < 1e-6


def test_data_fusion_with_and_without_projection_reuse(gpu):
    """multimodal::data_fusion projects the model volume for its cost and then runs SIRT from a copy of it (multimodal.cpp:452-470);
    the engine starts that SIRT run from the projection it already has ("fp_reuse").  Bit-identical to projecting twice."""
    outs = []
    for reuse in (1, 0):
        dev, ref, gt = make_case(gamma=1.6)
        dev.he.set_option("fp_reuse", reuse)
        dev.ce.set_option("fp_reuse", reuse)
        for _ in range(2):
            dev.poisson_ml(0.05)
        dev.rescale_tomograms(10)
        dev.rescale_projections()
        costs = [dev.sirt_data_fusion(10, 0.05, 3) for _ in range(2)]
        outs.append((np.array(costs), dev.get_volume()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_chemicaltomo_driver_runs_and_reduces_costs(gpu):
    """ChemicalTomo.data_fusion end to end (chemistry/reconstructor.py:182-225) on a small synthetic sample."""
    N, Nx, P = 32, 6, 11
    ang = np.linspace(-70, 70, P)
    gt = np.stack([ellipsoids(Nx, N, seed=3 + e, k=8) for e in range(2)])
    ref = ref_multimodal(Nx, N, 2, ang, ang)
    chem = {}
    for e, el in enumerate(["Zn", "O"]):
        chem[el] = ref._fp(ref.C, gt[e]).reshape(Nx, P, N).transpose(0, 2, 1).copy()
    ref.w = create_weighted_summation_weights([30, 8], 1.6, 3)
    ref.gamma = np.float32(1.6)
    ref.recon = gt
    haadf = ref._fp(ref.H, ref.model()).reshape(Nx, P, N).transpose(0, 2, 1).copy()
    ct = ChemicalTomo(haadf, ang, chem, ang)
    cost0 = ct.chemical_tomography(Niter=20)
    assert cost0[-1] < cost0[0]
    h, c, tv = ct.data_fusion(Niter=6, chem_iters=20)
    assert np.all(np.isfinite(h)) and np.all(np.isfinite(c)) and h[-1] < h[0]
    rec = ct.get_recon()
    assert rec.shape == (2, Nx, N, N) and rec.min() >= 0


def test_config5_full_size_properties(gpu):
    """BASELINE config 5 on one GPU: ADF + 2 spectral channels, 512^3, 70 tilts.  The oracle is far too slow here, so:
    (a) the Poisson-ML steps are independent per slice -- two half-slab engines on the corresponding halves of the data
    reproduce the whole engine voxel for voxel (what tilt-axis sharding of ChemicalTomo relies on); (b) the Poisson-ML
    cost falls; (c) a data-fusion iteration + the 4-D FGP prox stay finite, non-negative and lower the
    HAADF misfit; (d) the model projections are what the engine's own forward projector makes of sum_e w_e x_e^gamma."""
    from tomo_tv_amd.phantom import tilt_angles
    nx, n, p, nel = 512, 512, 70, 2
    ang = np.deg2rad(tilt_angles(p))
    w = create_weighted_summation_weights([30, 8], 1.6, 3)
    gt = np.stack([ellipsoids(nx, n, seed=5 + e) * np.float32(0.5 + 0.3 * e) for e in range(nel)])

    def build(s0, ns, bh=None, bchem=None):
        mm = multimodal(ns, n, nel, ang, ang)
        mm.set_gamma(1.6)
        mm.set_weights(w)
        if bh is None:                      # synthetic measurements through the engine's own operators
            mm.set_volume(gt[:, s0:s0 + ns])
            mm._mm_model()
            mm.he.be.c("forward_projection", mm.MODEL, 0)
            bh = mm.he.get_projections()
            bh = bh / bh.max()
            for e in range(nel):
                mm.ce.be.c("forward_projection", int(mm._x[e]), int(mm._b[e]))
            bchem = mm.get_chem_projections()
            bchem = bchem / bchem.max()
            mm.restart_recon()
        mm.set_haadf_tilt_series(bh[s0:s0 + ns] if bh.shape[0] != ns else bh)
        mm.set_chem_tilt_series(bchem[s0:s0 + ns] if bchem.shape[0] != ns else bchem)
        mm.set_measureChem(True)
        mm.set_measureHaadf(True)
        mm.estimate_lipschitz()
        return mm, bh, bchem

    whole, bh, bchem = build(0, nx)
    costs = [whole.poisson_ml(0.05) for _ in range(3)]
    assert np.all(np.isfinite(costs)) and costs[2] < costs[0]
    vol = whole.get_volume()
    assert vol.shape == (nel, nx, n, n) and vol.min() >= 0
    parts = []
    for s0 in (0, nx // 2):
        half, _, _ = build(s0, nx // 2, bh, bchem)
        for _ in range(3):
            half.poisson_ml(0.05)
        parts.append(half.get_volume())
        del half
    assert rel_l2(np.concatenate(parts, axis=1), vol) < 2e-6
    del parts, vol
    whole.rescale_tomograms(10)
    whole.rescale_projections()
    h0, c0 = whole.sirt_data_fusion(10, 0.05, 5)
    tv0 = whole.tv_fgp_4D(5, 1e-4)
    h1, c1 = whole.sirt_data_fusion(10, 0.05, 5)
    tv1 = whole.tv_fgp_4D(5, 1e-4)
    assert np.all(np.isfinite([h0, c0, tv0, h1, c1, tv1])) and h1 < h0 and tv0 > 0 and tv1 > 0
    assert whole.get_volume().min() >= 0
