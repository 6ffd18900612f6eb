"""Host-only: the file helpers either side of the path (tomo_tv_amd/io.py, mirroring tomofusion/pytvlib.py:57-162)."""
import os

import numpy as np
import pytest

from tomo_tv_amd import io as tio


class _FakeTomo:
    def __init__(self, vol, nproj):
        self.vol, self.Nslice_, self.Nproj, self.comm = vol, vol.shape[0], nproj, None

    def get_recon(self, s):
        return self.vol[s]


def _read(path):
    if path.endswith(".npz"):
        return dict(np.load(path))
    import h5py
    out = {}
    with h5py.File(path, "r") as f:
        def visit(name, obj):
            if isinstance(obj, h5py.Dataset):
                out[name] = obj[()]
            for k, v in obj.attrs.items():
                out[f"{name}/{k}"] = np.asarray(v)
        f.visititems(visit)
    return out


def test_npy_tilt_series_roundtrip(tmp_path):
    d = tmp_path / "Tilt_Series"
    d.mkdir()
    ts = np.random.default_rng(0).random((6, 8, 5), dtype=np.float32)
    np.save(d / "256_Co2P_tiltser.npy", ts)
    name, got = tio.load_data("256", "Co2P_tiltser.npy", dir=str(d))
    assert name == "Co2P" and np.array_equal(got, ts)
    with pytest.raises(ValueError):
        tio.load_data("256", "Co2P_tiltser.raw", dir=str(d))


def test_results_tree_and_recon(tmp_path):
    root = str(tmp_path / "results")
    vol = np.random.default_rng(1).random((3, 4, 4), dtype=np.float32)
    p = tio.save_results(("run1", "asd"), meta={"Niter": 20, "eps": 0.025}, results={"dd": np.arange(5.0), "tv": np.ones(5)}, root=root)
    p2 = tio.save_recon(("run1", "asd"), (3, 4, 7), _FakeTomo(vol, 7), root=root)
    assert p == p2 and os.path.exists(p)
    t = _read(p)
    assert np.allclose(t["results/dd"], np.arange(5.0)) and t["results/dd"].dtype == np.float32
    assert int(t["parameters/Niter"]) == 20 and abs(float(t["parameters/eps"]) - 0.025) < 1e-12
    assert np.array_equal(t["Reconstruction/recon"], vol)
    assert int(t["Reconstruction/Nproj"]) == 7


def test_h5_style_input_from_npz(tmp_path):
    d = tmp_path / "Tilt_Series"
    d.mkdir()
    ts, ang = np.zeros((2, 3, 4), np.float32), np.linspace(-70, 70, 4)
    np.savez(d / "au_sto.npz", tiltSeries=ts, tiltAngles=ang)
    name, a, v = tio.load_h5_data("", "au_sto.npz", dir=str(d))
    assert name == "au_sto" and np.array_equal(np.asarray(a), ang) and np.asarray(v).shape == ts.shape


def test_mpi_save_results_single_rank(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    vol = np.arange(2 * 3 * 3, dtype=np.float32).reshape(2, 3, 3)
    p = tio.mpi_save_results(("out", "rec"), _FakeTomo(vol, 5), True, meta={"alg": 1}, results={"dd": [1.0, 0.5]})
    t = _read(p)
    assert np.array_equal(t["Reconstruction/recon"], vol) and np.allclose(t["results/dd"], [1.0, 0.5])
