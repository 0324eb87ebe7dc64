"""Thin device wrappers of the three registration C-ABI calls (include/bhcore.h, csrc/regmetric.hip)."""

from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib
from ..device import get_context, ptr


def _check_vol(t: torch.Tensor):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.ndim == 3 and t.is_contiguous()):
        raise ValueError("expected a contiguous float32 CUDA tensor of shape (Z, Y, X)")


def image_stats(vol: torch.Tensor) -> dict:
    """min, max, sum and intensity centre of mass (ZYX index units) of a device volume."""
    _check_vol(vol)
    out = (C.c_double * 6)()
    ctx = get_context(vol.device)
    Z, Y, X = vol.shape
    _lib.check(ctx.lib.bh_image_stats(ctx.handle, ptr(vol), Z, Y, X, out))
    mn, mx, s, sz, sy, sx = (float(v) for v in out)
    com = np.array([sz, sy, sx]) / s if s != 0 else (np.array(vol.shape, dtype=np.float64) - 1) / 2
    return {"min": mn, "max": mx, "sum": s, "center_of_mass": com}


def smooth_shrink_geometry(shape, factor):
    """(out_shape, offset) of a pyramid level: level index i <-> input index factor * i + offset."""
    os_, off = (C.c_int64 * 3)(), (C.c_int64 * 3)()
    f = (C.c_int * 3)(*[int(v) for v in factor])
    sg = (C.c_double * 3)(0.0, 0.0, 0.0)
    Z, Y, X = (int(s) for s in shape)
    _lib.check(_lib.load().bh_smooth_shrink(None, None, Z, Y, X, sg, f, None, os_, off))
    return tuple(int(v) for v in os_), tuple(int(v) for v in off)


def smooth_shrink(vol: torch.Tensor, sigma, factor):
    """Gaussian-smooth (sigma in voxels per axis) and subsample by integer factors; returns (level, offset)."""
    _check_vol(vol)
    sigma = [float(s) for s in (sigma if np.ndim(sigma) else (sigma,) * 3)]
    factor = [int(f) for f in (factor if np.ndim(factor) else (factor,) * 3)]
    out_shape, offset = smooth_shrink_geometry(vol.shape, factor)
    ctx = get_context(vol.device)
    with torch.cuda.device(vol.device):
        out = torch.empty(out_shape, dtype=torch.float32, device=vol.device)
    os_, off = (C.c_int64 * 3)(), (C.c_int64 * 3)()
    Z, Y, X = vol.shape
    _lib.check(ctx.lib.bh_smooth_shrink(ctx.handle, ptr(vol), Z, Y, X, (C.c_double * 3)(*sigma), (C.c_int * 3)(*factor),
                                        ptr(out), os_, off))
    return out, offset


def mattes_mi(fixed: torch.Tensor, moving: torch.Tensor, pull_3x4, intensity_range, bins: int = 32, stride: int = 1,
              offset: int = 0):
    """(MI, dMI/dP as 3x4, samples used) for fixed(p) vs moving(P p); see ``bh_mattes_mi``."""
    _check_vol(fixed)
    _check_vol(moving)
    P = np.ascontiguousarray(np.asarray(pull_3x4, dtype=np.float64).reshape(3, 4))
    ctx = get_context(fixed.device)
    val, nv = C.c_double(), C.c_double()
    grad = (C.c_double * 12)()
    _lib.check(ctx.lib.bh_mattes_mi(ctx.handle, ptr(fixed), *fixed.shape, ptr(moving), *moving.shape,
                                    (C.c_double * 12)(*P.ravel()), (C.c_double * 4)(*[float(v) for v in intensity_range]),
                                    int(bins), int(stride), int(offset), C.byref(val), grad, C.byref(nv)))
    return float(val.value), np.array(grad, dtype=np.float64).reshape(3, 4), int(nv.value)


def sobel(vol: torch.Tensor) -> torch.Tensor:
    """``skimage.filters.sobel`` of a 3-D device volume (gradient magnitude, reflected edges)."""
    _check_vol(vol)
    ctx = get_context(vol.device)
    with torch.cuda.device(vol.device):
        out = torch.empty_like(vol)
    _lib.check(ctx.lib.bh_sobel(ctx.handle, ptr(vol), *vol.shape, ptr(out)))
    return out
