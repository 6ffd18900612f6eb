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


def _write_plain_tiff(path, stack, bo="<"):
    """A minimal multi-page uncompressed TIFF (one strip per page), the form the built-in reader of tomo_tv_amd/io.py handles."""
    import struct
    stack = np.asarray(stack)
    kind = {"u": 1, "i": 2, "f": 3}[stack.dtype.kind]
    bits = stack.dtype.itemsize * 8
    out = bytearray(b"II" if bo == "<" else b"MM") + struct.pack(bo + "HI", 42, 8)
    for k, page in enumerate(stack):
        data = page.astype(stack.dtype.newbyteorder(bo)).tobytes()
        ifd = len(out)
        tags = [(256, 4, page.shape[1]), (257, 4, page.shape[0]), (258, 3, bits), (259, 3, 1), (262, 3, 1),
                (273, 4, ifd + 2 + 12 * 10 + 4), (277, 3, 1), (278, 4, page.shape[0]), (279, 4, len(data)), (339, 3, kind)]
        out += struct.pack(bo + "H", len(tags))
        for tag, typ, val in tags:
            out += struct.pack(bo + "HHI", tag, typ, 1) + (struct.pack(bo + "I", val) if typ == 4 else struct.pack(bo + "HH", val, 0))
        nxt = ifd + 2 + 12 * len(tags) + 4 + len(data) if k + 1 < len(stack) else 0
        out += struct.pack(bo + "I", nxt) + data
    open(path, "wb").write(bytes(out))


@pytest.mark.parametrize("dtype,bo", [(np.float32, "<"), (np.uint16, ">"), (np.uint8, "<")])
def test_load_data_reads_tiff_tilt_series(tmp_path, monkeypatch, dtype, bo):
    """pytvlib.py:57-79: a .tif / .tiff stack loads as (z, y, x) and is swapped to (x, y, z).  Checked through whatever reader
    the image offers AND through the built-in reader of uncompressed stacks (scikit-image, tifffile and Pillow hidden)."""
    from tomo_tv_amd import io as tio
    rng = np.random.default_rng(3)
    stack = (rng.random((5, 7, 9)) * 200).astype(dtype)                  # (angles, y, x)
    d = tmp_path / "Tilt_Series"
    d.mkdir()
    for ext in (".tif", ".tiff"):
        _write_plain_tiff(str(d / f"256_phantom_tiltser{ext}"), stack, bo)
        name, ts = tio.load_data("256", f"phantom_tiltser{ext}", dir=str(d) + "/")
        assert name == "phantom" and ts.dtype == np.float32 and ts.shape == (9, 7, 5)
        assert np.array_equal(ts, np.swapaxes(stack.astype(np.float32), 0, 2))
    plain = tio._read_plain_tiff(str(d / "256_phantom_tiltser.tif"))
    assert plain.shape == stack.shape and np.array_equal(plain.astype(dtype), stack)
    with pytest.raises(ValueError):
        open(str(d / "bad.tif"), "wb").write(b"not a tiff file at all")
        tio._read_plain_tiff(str(d / "bad.tif"))
