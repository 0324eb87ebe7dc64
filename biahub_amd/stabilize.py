"""Per-timepoint stabilisation warp on MI355X — mirror of ``biahub/stabilize.py:32-90``."""

from __future__ import annotations

import numpy as np

from . import _lib
from .device import resolve_device, to_host
from .register import affine_device


def apply_stabilization_transform(
    zyx_data: np.ndarray,
    list_of_shifts: list[np.ndarray],
    input_time_index: int,
    output_shape: tuple[int, int, int] = None,
    device="cuda",
):
    """Warp one (Z,Y,X) or (C,Z,Y,X) volume with ``list_of_shifts[input_time_index]``.

    Always linear interpolation, no crop, NaN -> 0, float32 result (stabilize.py:63-90).  An
    out-of-range ``input_time_index`` raises ``IndexError`` like the reference's list lookup (:68).
    """
    zyx_data = np.asarray(zyx_data)
    if output_shape is None:
        output_shape = zyx_data.shape[-3:]
    output_shape = tuple(int(s) for s in output_shape)
    matrix = np.asarray(list_of_shifts[input_time_index], dtype=np.float64)
    if zyx_data.ndim == 4:
        out = np.zeros((zyx_data.shape[0],) + output_shape, dtype=np.float32)
        for c in range(zyx_data.shape[0]):
            out[c] = apply_stabilization_transform(zyx_data[c], list_of_shifts, input_time_index, output_shape,
                                                   device=device)
        return out
    dev = resolve_device(device)
    return to_host(affine_device(zyx_data, matrix, output_shape, "linear", _lib.BOUNDARY_ITK, 0.0, device=dev))
