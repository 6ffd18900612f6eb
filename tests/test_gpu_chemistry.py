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
