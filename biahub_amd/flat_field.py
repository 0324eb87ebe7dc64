"""Flat-field correction — mirror of ``biahub/flat_field.py`` (the step before deskew in the mantis pipeline).

``flat_field_zyx(zyx) = zyx / median(zyx, axis=0) * median(zyx, axis=0).mean()`` (flat_field.py:101-120).  The median and
the division run in ``csrc/flatfield.hip`` (``bh_flat_field``): uint16 camera stacks cross PCIe at 2 B/voxel, the
pattern never leaves the device.
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .device import as_device_volume, get_context, ptr, to_host
from .device import empty as device_empty, empty_like as device_empty_like


def median_z_device(vol, device=None) -> torch.Tensor:
    """``np.median(vol, axis=0)`` on the device: float64 (Y, X) tensor (float32-valued for float32 input)."""
    t, code, dev = as_device_volume(vol, device)
    if t.ndim != 3:
        raise ValueError(f"expected a 3-D volume, got shape {tuple(t.shape)}")
    ctx = get_context(dev)
    Z, Y, X = (int(s) for s in t.shape)
    with torch.cuda.device(dev):
        pattern = torch.empty((Y, X), dtype=torch.float64, device=dev)
    _lib.check(ctx.lib.bh_median_z(ctx.handle, ptr(t), code, Z, Y, X, ptr(pattern)))
    return pattern


def _median_tiled(data: np.ndarray, axis: int, tile_bytes: int = 0) -> np.ndarray:
    """``np.median(data, axis=axis)`` (flat_field.py:56-99; ``tile_bytes`` was a CPU cache knob and is ignored).

    The kernel reduces along axis 0 of a 3-D volume; other axes / ranks are brought to that form by a view.
    """
    a = np.asarray(data)
    if a.ndim == 1:
        a, axis = a[:, None], 0
    moved = np.moveaxis(a, axis, 0)
    lead = moved.shape[1:]
    vol = np.ascontiguousarray(moved.reshape(moved.shape[0], 1, -1))
    out = median_z_device(vol).cpu().numpy().reshape(lead)
    if np.asarray(data).dtype == np.float32:
        out = out.astype(np.float32)  # np.median keeps float32; integer input gives float64
    return out if out.ndim else out[()]


def flat_field_device(vol, device=None, return_pattern: bool = False):
    """Device-level flat field: (Z, Y, X) volume in, float32 tensor out."""
    t, code, dev = as_device_volume(vol, device)
    if t.ndim != 3:
        raise ValueError(f"expected a 3-D volume, got shape {tuple(t.shape)}")
    ctx = get_context(dev)
    Z, Y, X = (int(s) for s in t.shape)
    with torch.cuda.device(dev):
        out = device_empty((Z, Y, X), torch.float32, dev)
        pattern = torch.empty((Y, X), dtype=torch.float64, device=dev) if return_pattern else None
    _lib.check(ctx.lib.bh_flat_field(ctx.handle, ptr(t), code, Z, Y, X, ptr(out),
                                     ptr(pattern) if return_pattern else None, None))
    return (out, pattern) if return_pattern else out


def flat_field_zyx(zyx_data: np.ndarray, axis: int = 0, device="cuda") -> np.ndarray:
    """Divide out the median pattern along ``axis`` (flat_field.py:101-120).  Returns float32 — the dtype the reference's
    CZYX adapter writes (:144-155); the arithmetic is float64 for integer input and float32 for float32 input, as numpy
    evaluates the reference expression."""
    a = np.asarray(zyx_data)
    if a.ndim != 3:
        raise ValueError(f"expected a 3-D volume, got shape {a.shape}")
    if axis % 3 != 0:  # the reference expression only broadcasts for axis 0 (zyx / pattern, flat_field.py:119-120)
        raise ValueError(f"operands could not be broadcast together: the median pattern along axis {axis} does not "
                         f"divide a {a.shape} volume; flat_field_zyx supports axis=0")
    return to_host(flat_field_device(a, device))


def flat_field_correction(zyx_data: np.ndarray, axis: int = 0, device="cuda") -> np.ndarray:
    """Deprecated alias (flat_field.py:123-141)."""
    import warnings

    warnings.warn("flat_field_correction is deprecated; use flat_field_zyx instead.", DeprecationWarning, stacklevel=2)
    return flat_field_zyx(zyx_data, axis=axis, device=device)


def _flat_field_czyx(czyx_data: np.ndarray, target_indices: list[int], device="cuda") -> np.ndarray:
    """Correct the channels in ``target_indices``; pass the others through as float32 (flat_field.py:144-155)."""
    czyx_data = np.asarray(czyx_data)
    out = np.empty(czyx_data.shape, dtype=np.float32)
    target = set(target_indices)
    for c in range(czyx_data.shape[0]):
        if c in target:
            out[c] = flat_field_zyx(czyx_data[c], device=device)
        else:
            out[c] = czyx_data[c].astype(np.float32)
    return out
