"""PSF estimate from bead volumes — mirror of ``biahub/estimate_psf.py`` (``estimate-psf`` command).

Detect beads, recentre and crop them, average the max-normalised crops, subtract the minimum and divide by the maximum
(estimate_psf.py:58-112).  The volume never leaves the GPU between detection and averaging: ``bh_block_peaks`` →
host-side candidate filtering → ``bh_patch_peaks`` → ``bh_average_patches``.
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .characterize_psf import _patch_margins, _starts, detect_peaks, recentre_beads
from .device import as_device_volume, get_context, ptr

# estimate_psf.py:58-67 ("Some of these settings can be moved to PsfFromBeadsSettings as needed")
BEAD_DETECTION_SETTINGS = {
    "block_size": (64, 64, 32),
    "blur_kernel_size": 3,
    "nms_distance": 32,
    "min_distance": 50,
    "threshold_abs": 200.0,
    "max_num_peaks": 2000,
    "exclude_border": (5, 10, 5),
}


def average_beads_device(vol: torch.Tensor, centres: np.ndarray, margins, normalise: bool = True):
    """Mean of the max-normalised full-size crops around ``centres``; returns (psf tensor, number of beads used)."""
    patch = [int(m) for m in margins]
    starts = _starts(centres, margins)
    shape = np.array(vol.shape)
    full = np.all(starts >= 0, axis=1) & np.all(starts + np.array(patch) <= shape, axis=1)  # same shape as a full crop
    starts = np.ascontiguousarray(starts[full].astype(np.int32))
    if len(starts) == 0:
        raise ValueError("No beads with a full-size patch were found.")
    ctx = get_context(vol.device)
    with torch.cuda.device(vol.device):
        out = torch.empty(patch, dtype=torch.float32, device=vol.device)
    _lib.check(ctx.lib.bh_average_patches(ctx.handle, ptr(vol), *vol.shape, starts.ctypes.data_as(C.POINTER(C.c_int)),
                                          len(starts), (C.c_int * 3)(*patch), int(normalise), ptr(out)))
    return out, len(starts)


def estimate_psf(pzyx_data, zyx_scale, patch_size=(101, 101, 101), bead_detection_settings: dict = None,
                 device="cuda", verbose: bool = False) -> np.ndarray:
    """The body of ``estimate_psf_cli`` (estimate_psf.py:69-112) for a list of bead volumes; returns the PSF (float32).

    Multi-position input: per-position normalised means are combined with bead-count weights (the reference concatenates
    all crops before the mean — the same number up to float rounding).
    """
    settings = dict(BEAD_DETECTION_SETTINGS)
    settings.update(bead_detection_settings or {})
    phys = tuple(a * b for a, b in zip(patch_size, zyx_scale))
    margins = _patch_margins(zyx_scale, phys)
    total, acc = 0, None
    single = len(pzyx_data) == 1
    for zyx in pzyx_data:
        t, _, dev = as_device_volume(np.asarray(zyx).astype(np.float32), device)
        peaks = detect_peaks(t, **settings, device=dev, verbose=verbose)
        centres = recentre_beads(t, peaks, margins)
        if len(centres) == 0:
            continue
        mean, n = average_beads_device(t, centres, margins, normalise=single)
        if single:  # one position: background subtraction and normalisation happen in the same call
            if verbose:
                print(f"Total beads: {n}")
            return mean.cpu().numpy()
        acc = mean.double() * n if acc is None else acc + mean.double() * n
        total += n
    if not total:
        raise ValueError("No beads were detected.")
    if verbose:
        print(f"Total beads: {total}")
    avg = (acc / total).float()
    avg = avg - avg.min()
    return (avg / avg.max()).cpu().numpy()
