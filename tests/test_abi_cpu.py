"""The C-ABI library loads on a CPU-only box and exports exactly what include/tomo_hip.h declares."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "tomo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tomo_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from tomo_tv_amd import _lib
    L = _lib.load()
    names = declared_symbols()
    assert len(names) >= 45
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/tomo_hip.h but not exported"
    # and the ctypes table binds every one of them (except the error getter, bound separately)
    assert set(names) - {"tomo_last_error"} == set(_lib.SIGNATURES)


def test_header_cites_reference_for_every_entry_point():
    src = open(os.path.join(ROOT, "include", "tomo_hip.h")).read()
    import re as _re
    assert len(_re.findall(r"(?:\.cpp|\.cu|\.py)?:\d+(?:-\d+)?", src)) >= 40  # file:line citations


def test_device_count_without_gpu_is_zero_or_more():
    from tomo_tv_amd import _lib
    assert _lib.device_count() >= 0


def test_missing_library_fails_loudly(monkeypatch):
    from tomo_tv_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libtomo_hip.so")
    with pytest.raises(_lib.TomoError, match="no CPU fallback"):
        _lib.load()


def test_communicator_entry_points_reject_bad_arguments_without_a_gpu():
    """tomo_comm_* (the native RCCL path of the slab sharding) on null engines / null buffers: error codes and a message, no
    crash, nothing initialised (librccl is only opened by tomo_comm_unique_id / tomo_comm_init on a real engine)."""
    from tomo_tv_amd import _lib
    L = _lib.load()
    w, r = ctypes.c_int(-1), ctypes.c_int(-1)
    assert L.tomo_comm_info(None, ctypes.byref(w), ctypes.byref(r)) != 0 and L.tomo_last_error()
    assert L.tomo_comm_unique_id(None) != 0
    for fn, args in (("tomo_comm_init", (None, None, 1, 0)), ("tomo_comm_share", (None, None)), ("tomo_comm_destroy", (None,)),
                     ("tomo_comm_exchange_halo", (None, 0)), ("tomo_comm_read_scalars", (None, None, 1)),
                     ("tomo_comm_scalars_snapshot", (None,)), ("tomo_comm_tv_gd", (None, 1, 0.1, 1e-6, -1, 0)),
                     ("tomo_comm_fgp_exchange", (None,))):
        assert getattr(L, fn)(*args) != 0, fn


def test_error_reporting_host_side():
    from tomo_tv_amd import _lib
    L = _lib.load()
    nnz = ctypes.c_int64(0)
    rc = L.tomo_system_matrix(0, 1, None, 0, None, None, None, ctypes.byref(nnz))
    assert rc != 0 and L.tomo_last_error()
    with pytest.raises(_lib.TomoError):
        _lib.check(rc)


@pytest.mark.parametrize("name", ["A_N16_P5.npz", "A_N32_P9.npz", "A_N64_P16.npz", "A_axis_N8.npz", "A_odd_N9.npz"])
def test_system_matrix_equals_reference_parallelRay(name):
    """Host-only entry point tomo_system_matrix vs the imported reference's output, bit for bit."""
    from tomo_tv_amd.engine import system_matrix
    g = np.load(os.path.join(GOLDEN, name))
    A = system_matrix(int(g["N"]), g["angles_deg"])
    assert A.shape == g["A"].shape
    assert np.array_equal(A, g["A"])


def test_system_matrix_digest_config1():
    import hashlib
    import json
    from tomo_tv_amd.engine import system_matrix
    dig = json.load(open(os.path.join(GOLDEN, "A_digest.json")))
    for key, (N, P) in {"N128_P31_lin70": (128, 31), "N256_P50_lin70": (256, 50)}.items():
        A = system_matrix(N, np.linspace(-70, 70, P))
        assert A.shape[1] == dig[key]["nnz"]
        assert hashlib.sha256(np.ascontiguousarray(A).tobytes()).hexdigest() == dig[key]["sha256"]


@pytest.mark.parametrize("N,P", [(256, 60), (512, 90), (512, 70), (1024, 120)])
def test_system_matrix_digest_baseline_geometries(N, P):
    """The product's builder at the BASELINE.json geometries (configs 2, 3, 5, 4) against sha256 digests of the IMPORTED
    reference's parallelRay output (tools/gen_golden.py --only-baseline-digests; cpu/utils/pytvlib.py:8-121): bit for bit."""
    import hashlib
    import json
    from tomo_tv_amd.engine import system_matrix
    d = json.load(open(os.path.join(GOLDEN, "A_digest.json")))[f"N{N}_P{P}_lin70"]
    A = system_matrix(N, np.linspace(-70, 70, P))
    assert A.shape[1] == d["nnz"]
    assert hashlib.sha256(np.ascontiguousarray(A).tobytes()).hexdigest() == d["sha256"]


def test_engine_creation_without_gpu_fails_loudly():
    """No CPU fallback anywhere: on a box without a HIP device the engine refuses to exist."""
    from tomo_tv_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    from tomo_tv_amd.engine import tomoengine
    with pytest.raises(_lib.TomoError, match="hip"):
        tomoengine(4, 16, np.deg2rad([0.0, 30.0]))
    from tomo_tv_amd.reconstructor import TomoGPU
    with pytest.raises(_lib.TomoError):
        TomoGPU(np.array([0.0, 30.0]), np.zeros((4, 16, 2), np.float32))


def test_product_never_imports_the_oracle():
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import tomo_tv_amd.engine, tomo_tv_amd.reconstructor, tomo_tv_amd.chemistry, "
            "tomo_tv_amd.pytvlib; assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle imported'" % ROOT)
    subprocess.check_call([sys.executable, "-c", code])
    for dirpath, _, files in os.walk(os.path.join(ROOT, "tomo_tv_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".inc")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "tomo_oracle" not in src, f


@pytest.mark.parametrize("N,P,amax", [(50, 9, 89.0), (96, 17, 70.0), (33, 4, 45.0), (7, 3, 60.0), (64, 1, 0.0)])
def test_tile_tables_replay_the_matrix(tmp_path, N, P, amax):
    """Host tables of the tile-stationary projectors (sysmat.cpp: build_tiles, build_bp_tiles), replayed on the CPU by
    tests/native/sysmat_tables_check.cpp: every matrix entry in exactly one stream, every tile segment used by exactly
    one row, forward = CSR product, backward cells = the cell table, ray windows within the kernel's LDS budget."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "stc")
    src = [os.path.join(root, "tests", "native", "sysmat_tables_check.cpp"), os.path.join(root, "tomo_tv_amd", "csrc", "sysmat.cpp")]
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(root, "tomo_tv_amd", "csrc"), *src, "-lpthread", "-o", exe],
                   check=True)
    r = subprocess.run([exe, str(N), str(P), str(amax)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr


def test_strip_tables_replay_the_matrix(tmp_path):
    """Host tables of the sheared-strip forward projector (sysmat.cpp: build_fp_strips), replayed on the CPU by
    tests/native/fp_strips_check.cpp exactly as k_fp_strip walks them (entry streams, per-(tile, wave, slot) batch counts, K
    accumulators, flush flags, row lists): every matrix entry in exactly one stream, every partial sum written once and
    used by exactly one row, forward = CSR product; odd sizes, one angle, axis-aligned and steep angles."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "fsc")
    src = [os.path.join(root, "tests", "native", "fp_strips_check.cpp"), os.path.join(root, "tomo_tv_amd", "csrc", "sysmat.cpp")]
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(root, "tomo_tv_amd", "csrc"), *src, "-lpthread", "-o", exe],
                   check=True)
    for N, P, amax, nchunk in [(50, 9, 89.0, 1), (96, 17, 70.0, 2), (33, 4, 45.0, 8), (7, 3, 60.0, 1), (64, 1, 0.0, 4), (16, 5, 70.0, 1),
                               (129, 12, 80.0, 3), (256, 60, 70.0, 4), (256, 60, 70.0, 16)]:
        r = subprocess.run([exe, str(N), str(P), str(amax), str(nchunk), "q"], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.rstrip().endswith("ok"), (N, P, amax, nchunk, r.stdout + r.stderr)


def test_forward_projector_lists_replay_the_matrix(tmp_path):
    """Host tables of the list form of the sheared-strip forward projector (sysmat.cpp: build_fp_lists), replayed on the CPU by
    tests/native/fp_lists_check.cpp as k_fp_list walks them (tiles staged alternately into two buffers, per (tile, wave) entry
    batches, then the flush records): every matrix entry in exactly one list, every accumulator clean when an item ends, every
    partial sum written once and owned by one ray, forward = CSR product."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "flc")
    src = [os.path.join(root, "tests", "native", "fp_lists_check.cpp"), os.path.join(root, "tomo_tv_amd", "csrc", "sysmat.cpp")]
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(root, "tomo_tv_amd", "csrc"), *src, "-lpthread", "-o", exe],
                   check=True)
    for N, P, amax in [(50, 9, 89.0), (96, 17, 70.0), (33, 4, 45.0), (7, 3, 60.0), (64, 1, 0.0), (16, 5, 70.0), (129, 12, 80.0), (256, 60, 70.0),
                       (200, 40, 70.0)]:
        r = subprocess.run([exe, str(N), str(P), str(amax), "q"], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.rstrip().endswith("ok"), (N, P, amax, r.stdout + r.stderr)


def test_back_projector_lists_replay_the_matrix(tmp_path):
    """Host tables of the entry-list back-projector (sysmat.cpp: build_bp_lists), replayed on the CPU by tests/native/bp_lists_check.cpp
    as k_bp_list walks them: every nonzero weight of the cell table exactly once, in the pixel's order of k_bp_all, its row inside the
    staged window of its angle and stage, padding only with weight 0; back projection = CSR transpose product.  Odd sizes, one
    angle, axis-aligned and steep angles, a number of angles that is no multiple of a stage."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "blc")
    src = [os.path.join(root, "tests", "native", "bp_lists_check.cpp"), os.path.join(root, "tomo_tv_amd", "csrc", "sysmat.cpp")]
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(root, "tomo_tv_amd", "csrc"), *src, "-lpthread", "-o", exe],
                   check=True)
    for N, P, amax in [(50, 9, 89.0), (96, 17, 70.0), (33, 4, 45.0), (7, 3, 60.0), (64, 1, 0.0), (16, 5, 70.0), (129, 13, 80.0), (256, 60, 70.0)]:
        r = subprocess.run([exe, str(N), str(P), str(amax), "q"], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.rstrip().endswith("ok"), (N, P, amax, r.stdout + r.stderr)


def test_resident_sweep_tables_replay_the_matrix(tmp_path):
    """Host tables of the volume-resident SART sweep (tomo_tv_amd/csrc/resident.cpp: build_sart_resident), replayed on the CPU by
    tests/native/resident_check.cpp as k_sart_resident uses them: block sums per wave through the fpc cells, tile sums over the
    waves by window slot, ray sums over the reducer lists = CSR product; the bpc cells = the cell table (rays through the wave's
    window, weights, the single-precision divisor); windows within the kernel's limits; and the geometries it must refuse."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "rsc")
    src = [os.path.join(root, "tests", "native", "resident_check.cpp"), os.path.join(root, "tomo_tv_amd", "csrc", "sysmat.cpp"),
           os.path.join(root, "tomo_tv_amd", "csrc", "resident.cpp")]
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(root, "tomo_tv_amd", "csrc"), *src, "-lpthread", "-o", exe], check=True)
    for N, P, amax in [(32, 7, 60.0), (64, 16, 70.0), (40, 9, 45.0), (96, 20, 89.0), (128, 31, 70.0), (8, 3, 60.0), (64, 1, 0.0), (256, 60, 70.0)]:
        r = subprocess.run([exe, str(N), str(P), str(amax)], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.startswith("OK"), (N, P, amax, r.stdout + r.stderr)
    r = subprocess.run([exe, "512", "9", "70", "64"], capture_output=True, text=True)         # 256 tiles on a 64-CU device
    assert r.returncode == 2 and "more tiles than resident workgroups" in r.stdout, r.stdout + r.stderr


def test_facades_carry_every_method_of_the_reference_tables(golden):
    """tests/golden/method_tables.json holds the method NAMES of the reference's five pybind11 classes; every one of
    them must exist on the class of the same name here (the drop-in boundary of SURVEY.md section 8b)."""
    import json
    import os
    from conftest import GOLDEN
    from tomo_tv_amd import chemistry, engine
    tables = json.load(open(os.path.join(GOLDEN, "method_tables.json")))["tables"]
    cls = {"tomoengine": engine.tomoengine, "multigpuengine": engine.multigpuengine, "ctvlib": engine.ctvlib,
           "multimodal": chemistry.multimodal, "multigpufusion": chemistry.multigpufusion}
    for name, methods in tables.items():
        missing = [m for m in methods if not callable(getattr(cls[name], m, None))]
        assert not missing, (name, missing)


def test_cpu_harness_signatures_match_the_reference():
    """cpu/utils/pytvlib.py:171-206: run(tomo, alg, beta=1), initialize_algorithm(tomo, alg, Nray, tiltAngles, angleStart=0),
    create_projections(tomo, original_volume, SNR=0), load_exp_tilt_series(tomo, tiltSeries)."""
    import inspect
    from tomo_tv_amd import cpu_harness as H
    sig = lambda f: [(p.name, p.default) for p in inspect.signature(f).parameters.values()]  # noqa: E731
    E = inspect.Parameter.empty
    assert sig(H.run) == [("tomo", E), ("alg", E), ("beta", 1)]
    assert sig(H.initialize_algorithm) == [("tomo", E), ("alg", E), ("Nray", E), ("tiltAngles", E), ("angleStart", 0)]
    assert sig(H.create_projections) == [("tomo", E), ("original_volume", E), ("SNR", 0)]
    assert sig(H.load_exp_tilt_series) == [("tomo", E), ("tiltSeries", E)]
    assert sig(H.parallelRay)[0][0] == "Nside"


def test_table_builders_are_clean_under_asan_and_ubsan():
    """sysmat.cpp and resident.cpp are index arithmetic end to end and ship inside libtomo_hip.so: the five table replays of
    tests/native, built with -fsanitize=address,undefined (no recovery), on two geometries each (tests/native/Makefile: san)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["make", "-C", os.path.join(root, "tests", "native"), "-s", "san"], capture_output=True, text=True)
    assert r.returncode == 0 and "san: clean" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
