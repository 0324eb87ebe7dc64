"""ctypes binding of libbhcore.so — the only door from Python into the HIP kernels.

There is NO CPU fallback: if the shared library is missing, or no MI355X is visible, the
operators raise.  (The CPU oracle lives under ``oracle/`` and is test infrastructure only.)
"""

from __future__ import annotations

import ctypes as C
import os
import threading
from pathlib import Path

PKG = Path(__file__).resolve().parent
LIB_PATH = PKG / "libbhcore.so"

BH_OK, BH_ERR_INVALID, BH_ERR_HIP, BH_ERR_NOMEM, BH_ERR_UNSUPPORTED = range(5)
DT_U8, DT_U16, DT_F32, DT_I16 = 0, 1, 2, 3
FILL_NONE, FILL_CONSTANT, FILL_MEAN = 0, 1, 2
INTERP_NEAREST, INTERP_LINEAR, INTERP_CUBIC = 0, 1, 3
BOUNDARY_ITK, BOUNDARY_SCIPY_CONSTANT, BOUNDARY_ZEROS = 0, 1, 2
PCC_NORM = {None: 0, "magnitude": 1, "classic": 2}
FILTER_F32, FILTER_BF16 = 0, 1
(T_DESKEW, T_FILL, T_RL_TOTAL, T_TIKHONOV, T_AFFINE, T_CROPFLIP, T_RL_ITER, T_TF, T_FLATFIELD) = range(9)

_i64, _f64, _f32, _int, _vp = C.c_int64, C.c_double, C.c_float, C.c_int, C.c_void_p

# name -> (restype, argtypes); mirrors include/bhcore.h one to one
SIGNATURES = {
    "bh_abi_version": (_int, []),
    "bh_last_error": (C.c_char_p, []),
    "bh_device_count": (_int, [C.POINTER(_int)]),
    "bh_ctx_create": (_int, [_int, _vp, C.POINTER(_vp)]),
    "bh_ctx_destroy": (_int, [_vp]),
    "bh_ctx_set_stream": (_int, [_vp, _vp]),
    "bh_ctx_synchronize": (_int, [_vp]),
    "bh_ctx_release_workspace": (_int, [_vp]),
    "bh_ctx_workspace_bytes": (_int, [_vp, C.POINTER(C.c_uint64)]),
    "bh_malloc": (_int, [C.POINTER(_vp), C.c_uint64]),
    "bh_free": (_int, [_vp]),
    "bh_memcpy_h2d": (_int, [_vp, _vp, _vp, C.c_uint64]),
    "bh_memcpy_d2h": (_int, [_vp, _vp, _vp, C.c_uint64]),
    "bh_median_z": (_int, [_vp, _vp, _int, _i64, _i64, _i64, _vp]),
    "bh_flat_field": (_int, [_vp, _vp, _int, _i64, _i64, _i64, _vp, _vp, C.POINTER(_f64)]),
    "bh_bin_reduce": (_int, [_vp, _vp, _int, _i64, _i64, _i64, C.POINTER(_int), _int, _vp, C.POINTER(_f32)]),
    "bh_bin_finish": (_int, [_vp, _vp, _i64, _int, _f32, _f32, _f32, _int, _vp]),
    "bh_valid_mask": (_int, [_vp, _vp, _int, _i64, _vp, C.POINTER(C.c_uint64)]),
    "bh_bits_and": (_int, [_vp, _vp, _vp, _i64]),
    "bh_bits_unpack": (_int, [_vp, _vp, _i64, _vp]),
    "bh_blosc_unfilter": (_int, [_vp, _vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, _int]),
    "bh_blosc_filter": (_int, [_vp, _vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, _int]),
    "bh_blosc_lz4_bound": (C.c_uint64, [C.c_uint32, C.c_uint32, C.c_uint32]),
    "bh_blosc_lz4_compress": (_int, [_vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _int, _vp, _vp]),
    "bh_lz4_decompress_streams": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_uint32, _vp]),
    "bh_host_blosc_unfilter": (_int, [_vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, _int]),
    "bh_host_blosc_filter": (_int, [_vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, _int]),
    "bh_deskew_shape": (_int, [_i64, _i64, _i64, _f64, _f64, _int, _int, _f64, C.POINTER(_i64), C.POINTER(_f64)]),
    "bh_deskew": (_int, [_vp, _vp, _int, _i64, _i64, _i64, _f64, _f64, _int, _int, _int, _f32, _vp,
                         C.POINTER(_f32)]),
    "bh_deskew_rows": (_int, [_vp, _vp, _int, _i64, _i64, _i64, _f64, _f64, _int, _int, _int, _f32, _vp,
                              C.POINTER(_f32), _vp]),
    "bh_deskew_fill_path": (_int, [_vp, C.POINTER(_int)]),
    "bh_overhang_fill": (_int, [_vp, _vp, _i64, _i64, _i64, _int, _f32, _int, C.POINTER(_f32)]),
    "bh_overhang_fill_connectivity": (_int, [_vp, _vp, _i64, _i64, _i64, _int, _f32, _int, _int, C.POINTER(_f32)]),
    "bh_transfer_function": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp]),
    "bh_ctx_fft_plans_replaced": (_int, [_vp, C.POINTER(_int)]),
    "bh_richardson_lucy_plan": (_int, [_i64, _i64, _i64, _i64, _i64, _i64, C.POINTER(_i64), C.POINTER(_int)]),
    "bh_tikhonov": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _f64, _vp]),
    "bh_inverse_filter": (_int, [_vp, _vp, _vp, _int, _i64, _i64, _i64, _i64, _f64, _int, _int, _vp]),
    "bh_inverse_filter_create": (_int, [_vp, _vp, _int, _i64, _i64, _i64, _i64, _f64, _int, C.POINTER(_vp)]),
    "bh_inverse_filter_apply": (_int, [_vp, _vp, _vp, _int, _vp]),
    "bh_inverse_filter_destroy": (_int, [_vp]),
    "bh_inverse_filter_trim": (_int, []),
    "bh_phase_transfer_function_3d": (_int, [_vp, _i64, _i64, _i64, _f64, _f64, _f64, _i64, _f64, _f64, _f64, _int, _vp, _vp]),
    "bh_fluorescence_transfer_function_3d": (_int, [_vp, _i64, _i64, _i64, _f64, _f64, _f64, _i64, _f64, _f64, _vp]),
    "bh_fourier_central_cuboid": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64]),
    "bh_host_deskew": (_int, [_vp, _int, _i64, _i64, _i64, _f64, _f64, _int, _int, _int, _f32, _vp, C.POINTER(_f32), _int]),
    "bh_richardson_lucy": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _int, _f32, _vp]),
    "bh_richardson_lucy_create": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, C.POINTER(_vp)]),
    "bh_richardson_lucy_apply": (_int, [_vp, _vp, _vp, _int, _f32, _vp]),
    "bh_richardson_lucy_apply_rows": (_int, [_vp, _vp, _vp, _int, _f32, _vp, _vp, C.POINTER(_int)]),
    "bh_richardson_lucy_destroy": (_int, [_vp]),
    "bh_richardson_lucy_info": (_int, [_vp, C.POINTER(_i64), C.POINTER(_int), C.POINTER(_int), C.POINTER(C.c_uint64)]),
    "bh_alloc_layout": (_int, [C.POINTER(_int), C.POINTER(_int), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "bh_alloc_retained": (_int, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "bh_torch_alloc": (_vp, [C.c_size_t, _int, _vp]),
    "bh_torch_free": (None, [_vp, C.c_size_t, _int, _vp]),
    "bh_phase_cross_corr": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _int, C.POINTER(_f32), _vp]),
    "bh_phase_cross_corr_create": (_int, [_vp, _vp, _i64, _i64, _i64, _int, C.POINTER(_vp)]),
    "bh_phase_cross_corr_apply": (_int, [_vp, _vp, _vp, _int, _int, C.POINTER(_f32), _vp]),
    "bh_phase_cross_corr_destroy": (_int, [_vp]),
    "bh_image_stats": (_int, [_vp, _vp, _i64, _i64, _i64, C.POINTER(_f64)]),
    "bh_smooth_shrink": (_int, [_vp, _vp, _i64, _i64, _i64, C.POINTER(_f64), C.POINTER(_int), _vp, C.POINTER(_i64),
                                C.POINTER(_i64)]),
    "bh_mattes_mi": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, C.POINTER(_f64), C.POINTER(_f64), _int,
                            _i64, _i64, C.POINTER(_f64), C.POINTER(_f64), C.POINTER(_f64)]),
    "bh_sobel": (_int, [_vp, _vp, _i64, _i64, _i64, _vp]),
    "bh_block_peaks": (_int, [_vp, _vp, _i64, _i64, _i64, _int, C.POINTER(_int), _vp, _vp, C.POINTER(_i64)]),
    "bh_patch_peaks": (_int, [_vp, _vp, _i64, _i64, _i64, C.POINTER(_int), _int, C.POINTER(_int), _f64, C.POINTER(_i64)]),
    "bh_average_patches": (_int, [_vp, _vp, _i64, _i64, _i64, C.POINTER(_int), _int, C.POINTER(_int), _int, _vp]),
    "bh_affine": (_int, [_vp, _vp, _int, _i64, _i64, _i64, C.POINTER(_f64), _int, _int, _f32, _vp, _i64, _i64,
                         _i64, C.POINTER(_i64)]),
    "bh_spline_prefilter": (_int, [_vp, _vp, _int, _i64, _i64, _i64, _vp]),
    "bh_crop_flip": (_int, [_vp, _vp, _int, _i64, _i64, _i64, _i64, C.POINTER(_i64), _i64, _i64, _i64, _int,
                            _int, _int, _vp]),
    "bh_last_elapsed_ms": (_int, [_vp, _int, C.POINTER(_f32)]),
    "bh_ctx_set_timing": (_int, [_vp, _int]),
}

_lib = None
_lock = threading.Lock()


class BhError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libbhcore.so (built in-tree by ``python -m biahub_amd.build``)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = Path(os.environ.get("BHCORE_LIB", LIB_PATH))
        if not path.exists():
            raise ImportError(
                f"{path} not found: the HIP extension is not built. Run `python -m biahub_amd.build` "
                "(needs hipcc). biahub_amd has no CPU fallback."
            )
        lib = C.CDLL(str(path))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError here == header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def last_error() -> str:
    return load().bh_last_error().decode("utf-8", "replace")


def check(status: int) -> None:
    if status == BH_OK:
        return
    msg = last_error()
    if status == BH_ERR_INVALID:
        raise ValueError(msg)
    if status == BH_ERR_NOMEM:
        raise MemoryError(msg)
    if status == BH_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise BhError(msg)


def deskew_shape(shape, ls_angle_deg, px_to_scan_ratio, keep_overhang, average_n_slices=1, pixel_size_um=1.0):
    """Host-only geometry through the C-ABI (no GPU needed)."""
    out = (_i64 * 3)()
    vox = (_f64 * 3)()
    Z, Y, X = (int(s) for s in shape)
    check(load().bh_deskew_shape(Z, Y, X, float(ls_angle_deg), float(px_to_scan_ratio), int(bool(keep_overhang)),
                                 int(average_n_slices), float(pixel_size_um), out, vox))
    return tuple(int(v) for v in out), tuple(float(v) for v in vox)
