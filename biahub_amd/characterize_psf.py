"""Bead detection and extraction — mirror of the on-path parts of ``biahub/characterize_psf.py``.

``detect_peaks`` (characterize_psf.py:562-711) is the reference's torch approximation of ``peak_local_max``: box blur,
blocked max-pool with indices, then top-k / threshold / non-maximum suppression on at most ``max_num_peaks`` candidates.
The two pooling passes over the volume are one fused HIP kernel (``bh_block_peaks``, bit-identical to torch's CPU
pooling); the candidate filtering is the reference's logic on a few thousand points, here in NumPy.
``extract_beads`` (:173-190 + vendor BeadExtractor) recentres each point on the Gaussian-smoothed maximum of its crop
(``bh_patch_peaks``) and returns the crops.
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .device import as_device_volume, get_context, ptr


def block_peaks(vol: torch.Tensor, blur_kernel_size: int, block_size) -> tuple[np.ndarray, np.ndarray]:
    """(values, flat indices) of ``max_pool3d(avg_pool3d(vol))`` as ``detect_peaks`` configures them, flattened."""
    ctx = get_context(vol.device)
    blk = (C.c_int * 3)(*[int(b) for b in block_size])
    nb = (C.c_int64 * 3)()
    Z, Y, X = (int(s) for s in vol.shape)
    _lib.check(ctx.lib.bh_block_peaks(None, None, Z, Y, X, int(blur_kernel_size), blk, None, None, nb))
    n = int(nb[0]) * int(nb[1]) * int(nb[2])
    with torch.cuda.device(vol.device):
        values = torch.empty(n, dtype=torch.float32, device=vol.device)
        indices = torch.empty(n, dtype=torch.int64, device=vol.device)
    _lib.check(ctx.lib.bh_block_peaks(ctx.handle, ptr(vol), Z, Y, X, int(blur_kernel_size), blk, ptr(values), ptr(indices), nb))
    return values.cpu().numpy(), indices.cpu().numpy()


def detect_peaks(
    zyx_data: np.ndarray,
    block_size: int | tuple[int, int, int] = (8, 8, 8),
    nms_distance: int = 3,
    min_distance: int = 40,
    threshold_abs: float = 200.0,
    max_num_peaks: int = 500,
    exclude_border: tuple[int, int, int] | None = None,
    blur_kernel_size: int = 3,
    device: str = "cuda",
    verbose: bool = False,
):
    """Detect peaks with local maxima (characterize_psf.py:562-711); returns (N, 3) ZYX coordinates, brightest first."""
    if isinstance(block_size, int):
        block_size = (block_size,) * 3
    if not blur_kernel_size:
        raise ValueError("blur_kernel_size must be a positive odd number")  # the reference fails later (NameError)
    if blur_kernel_size % 2 != 1:
        raise ValueError(f"kernel_size={blur_kernel_size} must be an odd number")
    t, _, dev = as_device_volume(np.asarray(zyx_data).astype(np.float32) if not isinstance(zyx_data, torch.Tensor)
                                 else zyx_data.to(torch.float32), device)
    zyx_shape = tuple(int(s) for s in t.shape[-3:])
    peak_value, peak_idx = block_peaks(t.reshape(zyx_shape).contiguous(), blur_kernel_size, block_size)
    num_peaks = len(peak_idx)

    # top max_num_peaks brightest, brightest first (torch.topk; ties keep block order)
    order = np.argsort(-peak_value, kind="stable")[: min(max_num_peaks, num_peaks)]
    peak_value, peak_idx = peak_value[order], peak_idx[order]
    num_rejected_max_num_peaks = num_peaks - len(order)

    num_rejected_threshold_abs = 0
    if threshold_abs:
        keep = peak_value > threshold_abs
        num_rejected_threshold_abs = int((~keep).sum())
        peak_value, peak_idx = peak_value[keep], peak_idx[keep]

    coords = np.stack(np.unravel_index(peak_idx, zyx_shape), -1).astype(np.int64)
    f = coords.astype(np.float64)
    dist = np.sqrt(((f[:, None, :] - f[None, :, :]) ** 2).sum(-1)) if len(f) else np.zeros((0, 0))
    dist_mask = np.ones(len(coords), dtype=bool)
    nearby = np.argwhere(np.triu(dist < nms_distance, k=1))
    dist_mask[nearby[:, 1]] = False  # the peak in the second column is dimmer
    num_rejected_nms_distance = int((~dist_mask).sum())

    num_rejected_min_distance = 0
    if min_distance:
        close = dist < min_distance
        close[nearby[:, 0], nearby[:, 1]] = False
        dist_mask &= close.sum(1) < 2
        num_rejected_min_distance = int((~dist_mask).sum()) - num_rejected_nms_distance
    coords = coords[dist_mask]

    num_rejected_exclude_border = 0
    if exclude_border is not None:
        if not (isinstance(exclude_border, tuple) and len(exclude_border) == 3
                and all(isinstance(v, int) for v in exclude_border)):
            raise ValueError(f"invalid argument exclude_border={exclude_border}")
        for dim, size in enumerate(exclude_border):
            border_mask = (size < coords[:, dim]) & (coords[:, dim] < zyx_shape[dim] - size)
            coords = coords[border_mask]
            num_rejected_exclude_border += int((~border_mask).sum())

    if verbose:
        print(f"Number of peaks detected: {num_peaks}")
        print(f"Number of peaks rejected by max_num_peaks: {num_rejected_max_num_peaks}")
        print(f"Number of peaks rejected by threshold_abs: {num_rejected_threshold_abs}")
        print(f"Number of peaks rejected by nms_distance: {num_rejected_nms_distance}")
        print(f"Number of peaks rejected by min_distance: {num_rejected_min_distance}")
        print(f"Number of peaks rejected by exclude_border: {num_rejected_exclude_border}")
        print(f"Number of peaks returned: {len(coords)}")
    return coords


def _patch_margins(scale, patch_size):
    if patch_size is None:
        patch_size = (scale[0] * 15, scale[1] * 18, scale[2] * 18)
    return np.array(patch_size, dtype=np.float64) / np.array(scale, dtype=np.float64)


def _starts(points, margins):
    """BeadExtractor._create_slices: start = int(c - margin // 2), extent int(margin)."""
    pts = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    return np.array([[int(c - m // 2) for c, m in zip(pt, margins)] for pt in pts], dtype=np.int64).reshape(-1, 3)


def recentre_beads(vol: torch.Tensor, points, margins, sigma: float = 2.0) -> np.ndarray:
    """``BeadExtractor._find_closest_peak`` for every point that passes ``_in_margins``; returns the new centres."""
    shape = np.array(vol.shape, dtype=np.float64)
    pts = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    ok = ~(np.any(pts < margins / 2, axis=1) | np.any(pts >= shape - margins / 2, axis=1))
    pts = pts[ok]
    if len(pts) == 0:
        return np.zeros((0, 3), dtype=np.int64)
    patch = [int(m) for m in margins]
    starts = _starts(pts, margins).astype(np.int32)
    ctx = get_context(vol.device)
    peaks = (C.c_int64 * len(pts))()
    _lib.check(ctx.lib.bh_patch_peaks(ctx.handle, ptr(vol), *vol.shape, np.ascontiguousarray(starts).ctypes.data_as(
        C.POINTER(C.c_int)), len(pts), (C.c_int * 3)(*patch), float(sigma), peaks))
    local = np.stack(np.unravel_index(np.array(peaks, dtype=np.int64), patch), -1)
    offset = local - np.array(patch) // 2
    return np.array([[int(p[a] + offset[i, a]) for a in range(3)] for i, p in enumerate(pts)], dtype=np.int64)


def extract_beads(zyx_data, points, scale: tuple, patch_size: tuple = None, device="cuda"):
    """``characterize_psf.extract_beads`` (:173-190): recentred crops and their offsets; empty crops are dropped."""
    a = np.asarray(zyx_data)
    t, _, dev = as_device_volume(a.astype(np.float32), device)
    margins = _patch_margins(scale, patch_size)
    centres = recentre_beads(t, points, margins)
    beads, offsets = [], []
    for c in centres:
        st = _starts(c[None], margins)[0]
        sl = tuple(slice(int(s), int(s) + int(m)) for s, m in zip(st, margins))
        crop = a[sl]  # numpy slicing semantics, as the reference: truncated at the far border, empty for negative starts
        if crop.size > 0:
            beads.append(crop)
            offsets.append(tuple(int(v) for v in (np.array(c) - np.array(crop.shape) // 2)))
    return beads, offsets
