"""Bit-exact crop / flip on MI355X — mirror of ``biahub/utils/array_ops.py`` and ``biahub/flip.py``."""

from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .device import get_context, ptr, resolve_device, to_host


def _bounds(sl, n):
    if not isinstance(sl, slice):
        raise TypeError("slicing parameters must be slice objects")
    start, stop, step = sl.indices(n)
    if step != 1:
        raise ValueError("only unit-step slices are supported")
    return start, max(stop - start, 0)


def crop_flip_device(t: torch.Tensor, lo, shape, flip_y=False, flip_x=False, nan_to_zero=False) -> torch.Tensor:
    """(C,Z,Y,X) device tensor -> cropped / flipped copy, dtype preserved, bit-exact."""
    if t.ndim != 4:
        raise ValueError("expected a (C, Z, Y, X) tensor")
    t = t.contiguous()
    if nan_to_zero and not t.dtype.is_floating_point:
        nan_to_zero = False  # np.nan_to_num is the identity on integer arrays
    if nan_to_zero and t.element_size() not in (4, 8):
        raise NotImplementedError("nan_to_zero supports float32/float64 only")
    return _crop_raw(t, lo, shape, flip_y, flip_x, nan_to_zero)


def _to_device_raw(a: np.ndarray, dev) -> torch.Tensor:
    a = np.ascontiguousarray(a)
    if a.dtype.itemsize not in (1, 2, 4, 8) or a.dtype.kind not in "uifb":
        raise TypeError(f"unsupported dtype {a.dtype}")
    kind = {1: np.uint8, 2: np.int16, 4: np.int32, 8: np.int64}[a.dtype.itemsize]
    return torch.from_numpy(a.view(kind)).to(dev)


def _crop_numpy(a4: np.ndarray, slicing, nan_to_zero, device, flip_y=False, flip_x=False) -> np.ndarray:
    dev = resolve_device(device)
    b = [_bounds(s, n) for s, n in zip(slicing, a4.shape[1:])]
    lo, shape = [v[0] for v in b], [v[1] for v in b]
    is_float = a4.dtype.kind == "f"
    if nan_to_zero and is_float and a4.dtype.itemsize == 2:
        a4 = a4.astype(np.float32)  # float16: widen, clean, (reference returns float16) narrow below
        narrow = True
    else:
        narrow = False
    t = _to_device_raw(a4, dev)
    if nan_to_zero and is_float:
        # the kernel works on raw bits; tell it which float width this is through the itemsize
        out = _crop_raw(t, lo, shape, flip_y, flip_x, True)
    else:
        out = _crop_raw(t, lo, shape, flip_y, flip_x, False)
    res = to_host(out).view(a4.dtype)
    return res.astype(np.float16) if narrow else res


def _crop_raw(t, lo, shape, flip_y, flip_x, nan_bits):
    Cn, Zi, Yi, Xi = (int(s) for s in t.shape)
    out = torch.empty((Cn,) + tuple(int(s) for s in shape), dtype=t.dtype, device=t.device)
    if out.numel() == 0:
        return out
    ctx = get_context(t.device)
    lo_c = (C.c_int64 * 3)(*[int(v) for v in lo])
    with torch.cuda.device(t.device):
        _lib.check(ctx.lib.bh_crop_flip(ctx.handle, ptr(t), t.element_size(), Cn, Zi, Yi, Xi, lo_c, int(shape[0]),
                                        int(shape[1]), int(shape[2]), int(flip_y), int(flip_x), int(nan_bits),
                                        ptr(out)))
    return out


def copy_n_paste(zyx_data: np.ndarray, zyx_slicing_params: list, device="cuda") -> np.ndarray:
    """NaN -> 0 then crop a ZYX array (biahub/utils/array_ops.py:9-33)."""
    a = np.asarray(zyx_data)
    return _crop_numpy(a[None], zyx_slicing_params, True, device)[0]


def copy_n_paste_czyx(czyx_data: np.ndarray, czyx_slicing_params: list, device="cuda") -> np.ndarray:
    """Crop a CZYX array, dtype preserved, no NaN handling (biahub/utils/array_ops.py:36-59)."""
    return _crop_numpy(np.asarray(czyx_data), czyx_slicing_params, False, device)


def flip_zyx(zyx_data: np.ndarray, x: bool = False, y: bool = False, device="cuda") -> np.ndarray:
    """The body of ``biahub flip`` for one (t, c) volume: ``[:, :, ::-1]`` and/or ``[:, ::-1, :]`` (flip.py:26-32)."""
    a = np.asarray(zyx_data)
    full = [slice(0, n) for n in a.shape]
    return _crop_numpy(a[None], full, False, device, flip_y=y, flip_x=x)[0]


def _check_nan_n_zeros(input_array: np.ndarray) -> bool:
    """True when the array is all zeros or all NaN (biahub/utils/array_ops.py:62-76); host-side."""
    return bool(np.all(np.isnan(input_array)) or np.all(input_array == 0))
