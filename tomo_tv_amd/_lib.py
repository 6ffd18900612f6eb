"""ctypes binding of ``libtomo_hip.so`` (C ABI: ``include/tomo_hip.h``).

There is no CPU fallback: if the HIP library is missing or fails to load, importing an engine raises.
"""
import ctypes
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtomo_hip.so")

# enum mirrors (include/tomo_hip.h)
VOL_RECON, VOL_TEMP, VOL_ORIGINAL, VOL_YK, VOL_RECON_OLD = 0, 1, 2, 3, 4
SINO_B, SINO_G, SINO_R, SINO_USER0 = 0, 1, 2, 3
SINO_YK_MODEL = 1000     # read-only: A * yk as tomo_fista_project_yk formed it
VOL_USER0 = 5
S_DD, S_DIFF, S_TV, S_GNORM, S_RMSE, S_COST, S_L1, S_DIFF2, S_GNORM_ALL, S_COUNT = 0, 1, 2, 3, 4, 5, 6, 7, 8, 16
FIELD_FGP_D, FIELD_FGP_P1 = 100, 101
K_BP_ANGLE, K_FP_ANGLE, K_TV_GRAD, K_TV_UPDATE, K_FGP_OBJ, K_FGP_GRAD, K_SART_FUSED, K_FP_TILE, K_BP_TILE, K_FP_REDUCE, K_SART_RESIDENT = range(11)
# tomo_form (tomo_get_option "form_fp" / "form_bp" / "form_sart"): the kernel family an operation of an engine runs as
FORM_FP = ("rows", "tile", "strip", "list")
FORM_BP = ("all", "tile", "list")
FORM_SART = ("angle", "tile", "resident")

_i, _i64, _f, _p = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p
_pp = ctypes.POINTER(ctypes.c_void_p)

# name -> argtypes; every function returns int status except tomo_last_error
SIGNATURES = {
    "tomo_device_count": [ctypes.POINTER(_i)],
    "tomo_system_matrix": [_i, _i, _p, _i64, _p, _p, _p, ctypes.POINTER(_i64)],
    "tomo_create": [_i, _i, _i, _p, _i, _pp],
    "tomo_create_from_matrix": [_i, _i, _i, _i64, _p, _p, _p, _i, _pp],
    "tomo_destroy": [_p],
    "tomo_set_stream": [_p, _p],
    "tomo_synchronize": [_p],
    "tomo_get_device": [_p, ctypes.POINTER(_i)],
    "tomo_get_dims": [_p, ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_i64)],
    "tomo_set_tilt_series": [_p, _p],
    "tomo_get_sinogram": [_p, _i, _p],
    "tomo_set_volume": [_p, _i, _p],
    "tomo_get_volume": [_p, _i, _p],
    "tomo_set_slice": [_p, _i, _i, _p],
    "tomo_get_slice": [_p, _i, _i, _p],
    "tomo_restart_recon": [_p],
    "tomo_copy_volume": [_p, _i, _i],
    "tomo_copy_volume_from": [_p, _i, _p, _i],
    "tomo_forward_projection": [_p, _i, _i],
    "tomo_back_projection": [_p, _i, _i],
    "tomo_lipschitz": [_p, ctypes.POINTER(_f)],
    "tomo_row_inner_product": [_p],
    "tomo_sirt_landweber": [_p, _i, _f, _i],
    "tomo_sirt": [_p, _i, _i],
    "tomo_sirt_cimmino": [_p, _i, _f, _i],
    "tomo_lipschitz_cimmino": [_p, ctypes.POINTER(_f)],
    "tomo_release_geometry": [_p],
    "tomo_adopt_volumes": [_p, _p],
    "tomo_sart": [_p, _i, _f, _i, _p],
    "tomo_art": [_p, _f],
    "tomo_art_order": [_p, _f, _p],
    "tomo_poisson_ml": [_p, _f],
    "tomo_positivity": [_p, _i],
    "tomo_soft_threshold": [_p, _i, _f],
    "tomo_fista_momentum": [_p, _f],
    "tomo_fista_project_yk": [_p, ctypes.POINTER(_i)],
    "tomo_data_distance_sq": [_p, _i],
    "tomo_diff_norm_sq": [_p, _i, _i, _i],
    "tomo_data_distance_sq_async": [_p, _i],
    "tomo_async_wait": [_p],
    "tomo_l1_norm": [_p, _i],
    "tomo_read_scalars": [_p, _p, _i],
    "tomo_scalars_snapshot": [_p],
    "tomo_scalars_snapshot_read": [_p, _p, _i],
    "tomo_bind_scalar_buffer": [_p, _p],
    "tomo_bind_halo": [_p, _p, _p],
    "tomo_halo_pack": [_p, _i, _i, _p],
    "tomo_halo_pack_both": [_p, _i, _p, _p],
    "tomo_halo_local": [_p, _i],
    "tomo_set_slab_edges": [_p, _i, _i],
    "tomo_wait_for": [_p, _p],
    "tomo_halo_from": [_p, _i, _p, _p],
    "tomo_scalar_sum_from": [_p, _i, _p, _i, _i],
    "tomo_profile_intervals": [_p, _i, _p, _p, _p, _i, ctypes.POINTER(_i)],
    "tomo_tv_partial": [_p, _i, _f],
    "tomo_tv_set_target": [_p, _i],
    "tomo_tv_grad": [_p, _f],
    "tomo_tv_grad_tv": [_p, _f],
    "tomo_tv_update": [_p, _f, _i],
    "tomo_fgp_begin": [_p],
    "tomo_fgp_obj": [_p, _f],
    "tomo_fgp_grad": [_p, _f],
    "tomo_fgp_end": [_p, _i],
    "tomo_tv": [_p, _i, _f],
    "tomo_tv_gd": [_p, _i, _f, _f],
    "tomo_tv_fgp": [_p, _i, _f],
    "tomo_set_option": [_p, ctypes.c_char_p, _i],
    "tomo_get_option": [_p, ctypes.c_char_p, ctypes.POINTER(_i)],
    "tomo_set_sinogram": [_p, _i, _p],
    "tomo_cgls": [_p, _i, _i],
    "tomo_fbp": [_p, _p, _f, _i],
    "tomo_sirt_data": [_p, _i, _i, _i],
    "tomo_sart_data": [_p, _i, _i, _f, _i, _p],
    "tomo_sart_tracked": [_p, _i, _i, _f, _i, _p, _i, _i],
    "tomo_tv_update_tracked": [_p, _f, _i, _i, _i],
    "tomo_tv_update_planes": [_p, _f, _i, _p, _p],
    "tomo_tv_grad_planes": [_p, _f, _i, _p, _p],
    "tomo_tv_halo_apply": [_p, _f, _i, _p, _p],
    "tomo_tv_gd_tracked": [_p, _i, _f, _f, _i, _i],
    "tomo_poisson_residual": [_p, _i, _i, _i],
    "tomo_scale_volume": [_p, _i, _f],
    "tomo_sino_diff_norm_sq": [_p, _i, _i, _i],
    "tomo_sino_proj_max": [_p, _i, _p],
    "tomo_sino_proj_scale": [_p, _i, _p, _p],
    "tomo_fgp_begin_vol": [_p, _i],
    "tomo_tv_fgp_vol": [_p, _i, _i, _f],
    "tomo_bind_fgp_halo": [_p, _p, _p, _p, _p],
    "tomo_bind_fgp_halo2": [_p, _p, _p, _p, _p],
    "tomo_fgp_fused_begin": [_p, _i],
    "tomo_fgp_fused_step": [_p, _f, _i],
    "tomo_fgp_fused_step2": [_p, _f, _i],
    "tomo_fgp_fused_last": [_p, _f, _i],
    "tomo_fgp_fused_end": [_p, _f],
    "tomo_get_stream": [_p, _pp],
    "tomo_mm_model": [_p, _p, _i, _p, _f, _p, _i],
    "tomo_mm_update": [_p, _p, _p, _i, _p, _f, _f, _f, _p, _i, _i],
    "tomo_sart_chain_count": [_p, ctypes.POINTER(_i)],
    "tomo_comm_unique_id": [_p],
    "tomo_comm_init": [_p, _p, _i, _i],
    "tomo_comm_share": [_p, _p],
    "tomo_comm_destroy": [_p],
    "tomo_comm_info": [_p, ctypes.POINTER(_i), ctypes.POINTER(_i)],
    "tomo_comm_exchange_halo": [_p, _i],
    "tomo_comm_read_scalars": [_p, _p, _i],
    "tomo_comm_scalars_snapshot": [_p],
    "tomo_comm_tv_gd": [_p, _i, _f, _f, _i, _i],
    "tomo_comm_fgp_exchange": [_p],
    "tomo_comm_fgp_exchange2": [_p],
    "tomo_profile_enable": [_p, _i, _i],
    "tomo_profile_read": [_p, _i, ctypes.POINTER(_i64), ctypes.POINTER(ctypes.c_double)],
    "tomo_profile_read2": [_p, _i, ctypes.POINTER(_i64), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)],
}

_lib = None


class TomoError(RuntimeError):
    pass


def load():
    """Load libtomo_hip.so; raises if it is absent (the product has no CPU path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TomoError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C tomo_tv_amd/csrc` (hipcc, --offload-arch=gfx950). There is no CPU fallback.")
    # PyTorch-ROCm wheels bundle their own HIP runtime (torch/lib/libamdhip64.so) beside the system's, which this library links.  Both
    # can live in one process only when torch's initialises FIRST: after this library has opened the device, a later `import torch`
    # (the sharded classes import it lazily) finds "No HIP GPUs are available" (measured, round 6, ROCm 7.2 + torch 2.10+rocm7.0).
    # So torch, where it is installed, is imported and asked for its device count before the library is opened; TOMO_TORCH_FIRST=0
    # skips that (a host that never shards and wants to save the import).
    if os.environ.get("TOMO_TORCH_FIRST", "1") != "0" and "torch" not in sys.modules:
        try:
            import torch
            torch.cuda.device_count()
        except Exception:  # noqa: BLE001 -- torch is optional: single-GPU use needs none of it
            pass
    L = ctypes.CDLL(LIB_PATH)
    L.tomo_last_error.restype = ctypes.c_char_p
    L.tomo_last_error.argtypes = []
    for name, args in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError here = header/library mismatch
        fn.restype = _i
        fn.argtypes = args
    _lib = L
    return L


def check(rc):
    if rc != 0:
        msg = load().tomo_last_error()
        raise TomoError(f"libtomo_hip error {rc}: {msg.decode() if msg else '?'}")


def device_count():
    n = _i(0)
    check(load().tomo_device_count(ctypes.byref(n)))
    return n.value
