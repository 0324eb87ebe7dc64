"""Oblique light-sheet deskew on MI355X — host-side mirror of ``biahub/deskew.py``.

Same names, argument meaning and error behaviour as the reference's L1/L1b operators
(reference file:line in each docstring); the arithmetic runs in ``libbhcore.so``
(``csrc/deskew.hip``, ``csrc/fill.hip``) through the C-ABI in ``include/bhcore.h``.
"""

from __future__ import annotations

import ctypes as C
from typing import Literal

import numpy as np
import torch

from . import _lib
from .device import as_device_volume, get_context, ptr, resolve_device, to_host
from .device import empty as device_empty, empty_like as device_empty_like


def _get_averaged_shape(deskewed_data_shape: tuple, average_window_width: int) -> tuple:
    """Shape after N-slice averaging of axis 0 (biahub/deskew.py:157-177)."""
    return (-(-int(deskewed_data_shape[0]) // int(average_window_width)),) + tuple(deskewed_data_shape[1:])


def _average_n_slices(data, average_window_width=1):
    """Edge-pad axis 0 to a multiple of the window, then mean (biahub/deskew.py:43-68).

    Host helper kept for API parity; the device path fuses this into the deskew kernel.
    """
    data = np.asarray(data)
    w = int(average_window_width)
    rem = data.shape[0] % w
    if rem:
        data = np.concatenate([data, np.repeat(data[-1:], w - rem, axis=0)], axis=0)
    return data.reshape((data.shape[0] // w, w) + data.shape[1:]).mean(axis=1)


def _get_transform_matrix(ls_angle_deg: float, px_to_scan_ratio: float):
    """Pull matrix of the deskew geometry (biahub/deskew.py:180-210)."""
    ct = np.cos(ls_angle_deg * np.pi / 180)
    m = np.zeros((4, 4))
    m[0, 0], m[0, 2] = -px_to_scan_ratio * ct, px_to_scan_ratio
    m[1, 0] = -1
    m[2, 1] = -1
    m[3, 3] = 1
    return m


def get_deskewed_data_shape(
    raw_data_shape: tuple,
    ls_angle_deg: float,
    px_to_scan_ratio: float,
    keep_overhang: bool,
    average_n_slices: int = 1,
    pixel_size_um: float = 1,
):
    """Deskewed ZYX shape and voxel size (biahub/deskew.py:213-274).

    Raises ``ValueError("Dataset contains only overhang ...")`` like the reference when
    ``keep_overhang=False`` leaves nothing (deskew.py:262-267).  Evaluated by
    ``bh_deskew_shape`` (host-only C-ABI call; no GPU needed).
    """
    if len(raw_data_shape) != 3:
        raise ValueError(f"raw_data_shape must have 3 entries, got {raw_data_shape}")
    return _lib.deskew_shape(raw_data_shape, ls_angle_deg, px_to_scan_ratio, keep_overhang, average_n_slices,
                             pixel_size_um)


def _fill_args(overhang_fill):
    if isinstance(overhang_fill, str):
        if overhang_fill != "mean":
            raise ValueError(f'overhang_fill must be "mean" or a number, got {overhang_fill!r}')
        return _lib.FILL_MEAN, 0.0
    v = float(overhang_fill)
    return (_lib.FILL_NONE, 0.0) if v == 0 else (_lib.FILL_CONSTANT, v)


def fast_deskew_zyx(
    raw_data,
    ls_angle_deg: float,
    px_to_scan_ratio: float,
    keep_overhang: bool,
    average_n_slices: int = 1,
    overhang_fill: Literal["mean"] | float = 0,
    row_sums: torch.Tensor | None = None,
) -> torch.Tensor:
    """Fused deskew of a (Z_scan, Y_tilt, X_coverslip) volume (biahub/deskew.py:456-542).

    ``raw_data`` is a tensor already on the GPU (float32 like the reference, or
    uint8/uint16/int16 which the kernel widens on load).  Returns a float32 tensor
    ``(ceil(Y/N), X, Xp)`` on the same device.  One kernel replaces the reference's
    permute/flip copy, edge pad, grid build, ``grid_sample`` and mean.  ``overhang_fill`` other than 0: float32 volumes
    take the one-pass fill of ``csrc/deskew_rows.inc`` (fill value from row sums of the input, whole rows written once; the
    bit-mask pipeline of ``csrc/fill.hip`` re-runs the volume only when the data held exact zeros), other inputs the mask
    pipeline.  ``row_sums`` (float64 ``(Z, Y)`` on the device: sums over x of ``raw_data``) may be handed in by whoever
    produced the volume (``PreparedRichardsonLucy.apply(..., row_sums=...)``); it saves the one read that reduces them.
    """
    if not isinstance(raw_data, torch.Tensor):
        raise TypeError("fast_deskew_zyx expects a torch.Tensor on the GPU (use _fast_deskew_czyx for numpy)")
    if raw_data.ndim != 3:
        raise ValueError(f"raw_data must be 3-D (Z, Y, X), got shape {tuple(raw_data.shape)}")
    t, code, dev = as_device_volume(raw_data)
    Z, Y, X = (int(s) for s in t.shape)
    out_shape, _ = get_deskewed_data_shape((Z, Y, X), ls_angle_deg, px_to_scan_ratio, keep_overhang,
                                           average_n_slices)
    mode, value = _fill_args(overhang_fill)
    ctx = get_context(dev)
    if row_sums is not None and (row_sums.dtype != torch.float64 or tuple(row_sums.shape) != (Z, Y) or row_sums.device != t.device
                                 or not row_sums.is_contiguous()):
        raise ValueError("row_sums must be a contiguous float64 (Z, Y) tensor on the volume's device")
    with torch.cuda.device(dev):
        out = device_empty(out_shape, torch.float32, dev)
        _lib.check(ctx.lib.bh_deskew_rows(ctx.handle, ptr(t), code, Z, Y, X, float(ls_angle_deg),
                                          float(px_to_scan_ratio), int(bool(keep_overhang)), int(average_n_slices),
                                          mode, value, ptr(out), None, ptr(row_sums) if row_sums is not None else None))
    return out


def deskew_fill_path(device=None) -> int:
    """How the last deskew on ``device`` filled the overhang (diagnostic, synchronises): 0 mask pipeline or no fill, 1 one pass,
    2 one pass followed by the mask pipeline (the data held exact zeros)."""
    dev = resolve_device(device if device is not None else "cuda")
    ctx = get_context(dev)
    path = C.c_int(0)
    with torch.cuda.device(dev):
        _lib.check(ctx.lib.bh_deskew_fill_path(ctx.handle, C.byref(path)))
    return int(path.value)


def _host_deskew_zyx(zyx, ls_angle_deg, px_to_scan_ratio, keep_overhang, average_n_slices=1, overhang_fill=0,
                     num_threads: int = 0) -> np.ndarray:
    """``fast_deskew_zyx`` on the host's CPU threads (``bh_host_deskew``: libbhcore's own C++, no GPU and no torch device
    involved) — what ``--cluster debug`` with ``device: cpu`` runs on a box without a GPU (BASELINE config 1; reference
    biahub/deskew.py:762-766).  Same sample positions and operation order as the GPU kernels: bit-identical results."""
    a = np.asarray(zyx)
    if a.ndim != 3:
        raise ValueError(f"raw_data must be 3-D (Z, Y, X), got shape {a.shape}")
    if a.dtype not in (np.float32, np.uint16, np.uint8, np.int16):
        a = a.astype(np.float32)
    a = np.ascontiguousarray(a)
    code = {np.dtype(np.float32): _lib.DT_F32, np.dtype(np.uint16): _lib.DT_U16, np.dtype(np.uint8): _lib.DT_U8,
            np.dtype(np.int16): _lib.DT_I16}[a.dtype]
    Z, Y, X = (int(n) for n in a.shape)
    out_shape, _ = get_deskewed_data_shape((Z, Y, X), ls_angle_deg, px_to_scan_ratio, keep_overhang, average_n_slices)
    mode, value = _fill_args(overhang_fill)
    out = np.empty(out_shape, dtype=np.float32)
    _lib.check(_lib.load().bh_host_deskew(a.ctypes.data_as(C.c_void_p), code, Z, Y, X, float(ls_angle_deg),
                                          float(px_to_scan_ratio), int(bool(keep_overhang)), int(average_n_slices), mode, value,
                                          out.ctypes.data_as(C.c_void_p), None, int(num_threads)))
    return out


def _fast_deskew_czyx(data, device="cuda", num_splits=1, **kwargs):
    """CZYX adapter, numpy in / numpy out (biahub/deskew.py:551-579).

    Takes channel 0 only, like the reference (:559).  ``num_splits`` > 1 splits along input X
    (independent in the transform) and concatenates in reversed order along output Y (:560-573).
    """
    zyx = np.asarray(data)[0]
    if torch.device(device).type == "cpu":  # the reference's default `device: cpu` (settings.py:348-383): libbhcore's host path
        if num_splits > 1:
            chunks = np.array_split(zyx, num_splits, axis=2)
            return np.concatenate([_host_deskew_zyx(np.ascontiguousarray(c), **kwargs) for c in reversed(chunks)], axis=1)[None]
        return _host_deskew_zyx(zyx, **kwargs)[None]
    dev = resolve_device(device)
    if num_splits > 1:
        chunks = np.array_split(zyx, num_splits, axis=2)
        results = [
            to_host(fast_deskew_zyx(as_device_volume(np.ascontiguousarray(c), dev)[0], **kwargs))
            for c in reversed(chunks)
        ]
        return np.concatenate(results, axis=1)[None]
    t, _, _ = as_device_volume(zyx, dev)
    return to_host(fast_deskew_zyx(t, **kwargs))[None]


def _fast_deskew_czyx_device(data, device="cuda", num_splits=1, **kwargs):
    """``_fast_deskew_czyx`` with the result left ON the GPU: a float32 ``(1, Z', Y', X')`` device tensor instead of a numpy
    array.  ``io.process_single_position`` recognises device tensors and lets the output store permute and compress them in HBM
    (``ZarrArray.encode_volume_device``), so a compressed store's bytes cross PCIe once, compressed."""
    dev = resolve_device(device)
    if isinstance(data, torch.Tensor):  # already in HBM (io.process_single_position's device read path)
        zyx = data[0]
        if num_splits > 1:
            results = [fast_deskew_zyx(as_device_volume(c.contiguous(), dev)[0], **kwargs)
                       for c in reversed(torch.tensor_split(zyx, num_splits, dim=2))]
            return torch.cat(results, dim=1)[None]
        return fast_deskew_zyx(as_device_volume(zyx, dev)[0], **kwargs)[None]
    zyx = np.asarray(data)[0]
    if num_splits > 1:
        chunks = np.array_split(zyx, num_splits, axis=2)
        results = [fast_deskew_zyx(as_device_volume(np.ascontiguousarray(c), dev)[0], **kwargs) for c in reversed(chunks)]
        return torch.cat(results, dim=1)[None]
    t, _, _ = as_device_volume(zyx, dev)
    return fast_deskew_zyx(t, **kwargs)[None]


_fast_deskew_czyx_device.device_resident = True  # result handed over in HBM
_fast_deskew_czyx_device.device_input = True     # takes a (C, Z, Y, X) device tensor as well as a numpy array


def deskew_zyx(
    raw_data: np.ndarray,
    ls_angle_deg: float,
    px_to_scan_ratio: float,
    keep_overhang: bool,
    device: str = "cuda",
    average_n_slices: int = 1,
    overhang_fill: Literal["zero", "mean"] = "zero",
) -> np.ndarray:
    """numpy-in / numpy-out deskew with the legacy signature (biahub/deskew.py:371-453).

    Runs the production resampling kernel (``fast_deskew_zyx`` semantics) and then, for ``overhang_fill="mean"`` with
    ``keep_overhang``, the legacy fill (``_fill_overhang_with_mean``: 6-connected SciPy dilation, :445-450).  The
    reference's legacy resampler (MONAI 3-D trilinear, absent here) differs from its own production path on the last
    averaged slab when ``Y % N != 0``; this entry keeps the signature, shapes, fill semantics and the ``ValueError``
    for overhang-only data.
    """
    if overhang_fill not in ("zero", "mean"):
        raise ValueError(f'overhang_fill must be "zero" or "mean", got {overhang_fill!r}')
    raw = np.asarray(raw_data)
    get_deskewed_data_shape(raw.shape, ls_angle_deg, px_to_scan_ratio, keep_overhang)  # raises first
    out = _fast_deskew_czyx(raw[None], device=device, ls_angle_deg=ls_angle_deg, px_to_scan_ratio=px_to_scan_ratio,
                            keep_overhang=keep_overhang, average_n_slices=average_n_slices, overhang_fill=0)[0]
    if keep_overhang and overhang_fill == "mean":
        out = _fill_overhang_with_mean(out, device=device)
    return out


def _deskew_czyx(data, **kwargs):
    """Legacy CZYX adapter (biahub/deskew.py:547-548)."""
    return deskew_zyx(data[0], **kwargs)[None]


def fill_overhang(data: torch.Tensor, fill_value: float | None = None, dilation_iterations: int = 3,
                  connectivity: int = 26) -> torch.Tensor:
    """Replace zero-padded overhang (biahub/deskew.py:339-368 ``_fill_overhang_torch``).

    Returns a new tensor; ``fill_value=None`` uses the mean of the un-masked voxels.  ``connectivity`` 26 is the
    production dilation (max_pool3d), 6 the SciPy cross of the legacy ``_fill_overhang_with_mean``.
    """
    t, code, dev = as_device_volume(data)
    if code != _lib.DT_F32:
        t = t.to(torch.float32)
    out = t.clone()
    Z, Y, X = (int(s) for s in out.shape)
    ctx = get_context(dev)
    mode = _lib.FILL_MEAN if fill_value is None else _lib.FILL_CONSTANT
    with torch.cuda.device(dev):
        _lib.check(ctx.lib.bh_overhang_fill_connectivity(ctx.handle, ptr(out), Z, Y, X, mode,
                                                         0.0 if fill_value is None else float(fill_value),
                                                         int(dilation_iterations), int(connectivity), None))
    return out


def _fill_overhang_with_mean(data: np.ndarray, dilation_iterations: int = 3, debug_plot_path=None,
                             device="cuda") -> np.ndarray:
    """Legacy overhang fill, numpy in / numpy out (biahub/deskew.py:277-336): exact zeros, grown by
    ``dilation_iterations`` steps of SciPy's default (6-connected) structuring element, become the mean of the rest.
    ``debug_plot_path`` is accepted and ignored (no plotting here)."""
    a = np.asarray(data)
    if a.ndim != 3:
        raise ValueError(f"expected a 3-D volume, got shape {a.shape}")
    if dilation_iterations < 1:  # SciPy reads iterations < 1 as "until the mask stops changing": everything, mean of nothing
        raise ValueError("dilation_iterations must be at least 1")
    t, _, _ = as_device_volume(a.astype(np.float32, copy=False), resolve_device(device))
    return to_host(fill_overhang(t, None, dilation_iterations, connectivity=6)).astype(a.dtype, copy=False)
