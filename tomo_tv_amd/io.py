"""Data in / results out: the file helpers either side of the reconstruction path (tomofusion/pytvlib.py:57-162).

Same function names, argument meaning and on-disk naming as the reference: tilt series under ``Tilt_Series/`` as
``<vol_size>_<file_name>`` (``.npy``; ``.tif/.tiff`` through scikit-image, tifffile or Pillow when importable, else a built-in
reader of uncompressed stacks; ``.h5`` with datasets
``tiltSeries`` / ``tiltAngles`` when h5py is importable), results under ``results/<dir>/<name>.h5`` with the groups
``parameters`` (attributes), ``results`` (float32 datasets) and ``Reconstruction/recon`` (+ ``Nslice, Nray, Nproj``
attributes).  Where h5py is absent (this image) the same tree is written to ``<name>.npz`` with ``/``-joined keys
(``parameters/<key>``, ``results/<key>``, ``Reconstruction/recon`` ...), and ``load_h5_data`` reads such a file.
Pure host code: nothing here touches the device."""
import os

import numpy as np

try:  # optional, exactly as the reference uses them
    import h5py
except ImportError:  # pragma: no cover - this image
    h5py = None
try:
    from skimage import io as _skio
except ImportError:  # pragma: no cover - this image
    _skio = None

TILT_DIR = "Tilt_Series/"
RESULT_DIR = "results/"


def _read_tiff_stack(path):
    """A (pages, rows, columns) array from a TIFF stack -- what ``skimage.io.imread`` returns for the reference's tilt series
    (pytvlib.py:64-71).  scikit-image if present, else tifffile, else Pillow, else a reader for plain uncompressed strips (the form
    microscope exports and numpy-side tools write): 8 / 16 / 32-bit unsigned, signed or IEEE samples, either byte order."""
    if _skio is not None:
        return np.asarray(_skio.imread(path))
    try:
        import tifffile
        return np.asarray(tifffile.imread(path))
    except ImportError:
        pass
    try:
        from PIL import Image, ImageSequence
        with Image.open(path) as im:
            return np.stack([np.array(page) for page in ImageSequence.Iterator(im)])
    except ImportError:
        pass
    return _read_plain_tiff(path)


def _read_plain_tiff(path):
    import struct
    raw = open(path, "rb").read()
    bo = {b"II": "<", b"MM": ">"}.get(raw[:2])
    if bo is None or struct.unpack(bo + "H", raw[2:4])[0] != 42:
        raise ValueError(f"{path}: not a (classic) TIFF file")
    size = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 6: 1, 7: 1, 8: 2, 9: 4, 10: 8, 11: 4, 12: 8, 16: 8}
    fmt = {1: "B", 3: "H", 4: "I", 6: "b", 8: "h", 9: "i", 16: "Q"}
    pages, off = [], struct.unpack(bo + "I", raw[4:8])[0]
    while off:
        n = struct.unpack(bo + "H", raw[off:off + 2])[0]
        tags = {}
        for k in range(n):
            tag, typ, cnt = struct.unpack(bo + "HHI", raw[off + 2 + 12 * k:off + 10 + 12 * k])
            nbytes = size.get(typ, 1) * cnt
            pos = off + 10 + 12 * k if nbytes <= 4 else struct.unpack(bo + "I", raw[off + 10 + 12 * k:off + 14 + 12 * k])[0]
            if typ in fmt:
                tags[tag] = struct.unpack(bo + fmt[typ] * cnt, raw[pos:pos + nbytes])
        w, h = tags[256][0], tags[257][0]
        bits, comp, spp = tags.get(258, (1,))[0], tags.get(259, (1,))[0], tags.get(277, (1,))[0]
        kind = tags.get(339, (1,))[0]                        # SampleFormat: 1 unsigned, 2 signed, 3 IEEE
        if comp != 1 or spp != 1 or bits not in (8, 16, 32, 64):
            raise ValueError(f"{path}: only uncompressed single-sample TIFF pages are read without scikit-image / tifffile / Pillow")
        dt = np.dtype({1: "u", 2: "i", 3: "f"}[kind] + str(bits // 8)).newbyteorder(bo)
        data = b"".join(raw[o:o + c] for o, c in zip(tags[273], tags[279]))
        pages.append(np.frombuffer(data, dtype=dt, count=w * h).reshape(h, w))
        off = struct.unpack(bo + "I", raw[off + 2 + 12 * n:off + 6 + 12 * n])[0]
    return np.stack(pages)


def load_data(vol_size, file_name, dir=TILT_DIR):
    """pytvlib.py:57-79.  Returns (name without the ``_tiltser<ext>`` suffix, tilt series (x, y, angles))."""
    full_name = f"{vol_size}_{file_name}" if vol_size != "" else file_name
    path = os.path.join(dir, full_name)
    for ftype in (".tiff", ".tif"):
        if full_name.endswith(ftype):
            # the stack loads as (z, y, x): swap back to (x, y, z) like the reference
            ts = np.swapaxes(np.array(_read_tiff_stack(path), dtype=np.float32), 0, 2)
            return file_name.replace("_tiltser" + ftype, ""), ts
    if full_name.endswith(".npy"):
        return file_name.replace("_tiltser.npy", ""), np.load(path)
    raise ValueError(f"unsupported tilt-series file type: {full_name}")


def load_h5_data(vol_size, file_name, dir=TILT_DIR):
    """pytvlib.py:82-95.  Returns (name, tiltAngles, tiltSeries)."""
    full_name = f"{vol_size}_{file_name}" if vol_size != "" else file_name
    path = os.path.join(dir, full_name)
    if path.endswith(".npz") or (h5py is None and os.path.exists(os.path.splitext(path)[0] + ".npz")):
        z = np.load(os.path.splitext(path)[0] + ".npz")
        return os.path.splitext(file_name)[0], z["tiltAngles"], z["tiltSeries"]
    if h5py is None:
        raise ImportError("reading .h5 needs h5py (not in this image); an .npz with tiltSeries/tiltAngles is accepted")
    f = h5py.File(path, "r")
    return file_name.replace(".h5", ""), f["tiltAngles"], f["tiltSeries"]


class _Tree:
    """The h5 layout the reference writes, kept in one of two backends."""

    def __init__(self, base, mode):
        self.base, self.mode = base, mode
        self.h5 = h5py.File(base + ".h5", mode) if h5py is not None else None
        self.flat = {}
        if self.h5 is None and mode == "a" and os.path.exists(base + ".npz"):
            self.flat = dict(np.load(base + ".npz", allow_pickle=False))

    def group(self, name, attrs=None, datasets=None):
        if self.h5 is not None:
            g = self.h5.create_group(name)
            for k, v in (attrs or {}).items():
                g.attrs[k] = v
            for k, v in (datasets or {}).items():
                g.create_dataset(k, dtype=np.float32, data=v)
            return
        for k, v in (attrs or {}).items():
            self.flat[f"{name}/{k}"] = np.asarray(v)
        for k, v in (datasets or {}).items():
            self.flat[f"{name}/{k}"] = np.asarray(v, dtype=np.float32)

    def close(self):
        if self.h5 is not None:
            self.h5.close()
        else:
            np.savez(self.base + ".npz", **self.flat)
        return self.base + (".h5" if self.h5 is not None else ".npz")


def _base(fname, root):
    d = os.path.join(root, fname[0])
    os.makedirs(d, exist_ok=True)
    return os.path.join(d, fname[1])


def save_results(fname, meta=None, results=None, root=RESULT_DIR):
    """pytvlib.py:120-139: ``fname = (directory, name)``; meta -> attributes of "parameters", results -> float32
    datasets of "results".  Returns the path written."""
    t = _Tree(_base(fname, root), "w")
    if meta is not None:
        t.group("parameters", attrs=meta)
    if results is not None:
        t.group("results", datasets=results)
    return t.close()


def save_gif(fname, meta, gif, root=RESULT_DIR):
    """pytvlib.py:141-145 (the reference shadows its own argument there; this stores the frames and the slice index)."""
    t = _Tree(_base(fname, root), "a")
    t.group("gif", attrs={"img_slice": meta}, datasets={"gif": gif})
    return t.close()


def save_recon(fname, meta, tomo, root=RESULT_DIR):
    """pytvlib.py:147-162: gathers the volume slice by slice (``tomo.get_recon(s)``) and appends "Reconstruction"."""
    Nslice, Nray, Nproj = meta
    recon = np.zeros([Nslice, Nray, Nray], dtype=np.float32)
    for s in range(Nslice):
        recon[s, :, :] = tomo.get_recon(s)
    t = _Tree(_base(fname, root), "a")
    t.group("Reconstruction", attrs={"Nslice": Nslice, "Nray": Nray, "Nproj": Nproj}, datasets={"recon": recon})
    return t.close()


def mpi_save_results(fname, tomo, saveRecon, meta=None, results=None):
    """pytvlib.py:97-118 for the slab-sharded engine: every rank takes part in the gather (``get_recon`` broadcasts from
    the owner), rank 0 writes.  ``fname = (directory, name)``, written below the current directory like the reference."""
    rank = tomo.comm.rank if getattr(tomo, "comm", None) is not None else 0
    recon = None
    if saveRecon:
        recon = np.stack([tomo.get_recon(s) for s in range(tomo.Nslice_)]).astype(np.float32)
    if rank != 0:
        return None
    t = _Tree(_base(fname, ""), "w")
    if recon is not None:
        t.group("Reconstruction", attrs={"Nslice": recon.shape[0], "Nray": recon.shape[1], "Nproj": int(tomo.Nproj)},
                datasets={"recon": recon})
    if meta is not None:
        t.group("parameters", attrs=meta)
    if results is not None:
        t.group("results", datasets=results)
    return t.close()
