"""Phase cross-correlation on MI355X — mirror of ``biahub/estimate_stabilization.py:199-256``.

Only the FFT kernel of the stabilisation *estimate* lives here: two R2C transforms, the
fused normalised conjugate product, C2R, |.| + first-occurrence argmax on device.  The surrounding bookkeeping
(focus finding, StackReg, per-position orchestration) is out of scope.
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .device import as_device_volume, get_context, ptr, resolve_device


def phase_cross_corr(ref_img, mov_img, normalization=None, output_path=None, verbose: bool = False, device="cuda"):
    """Translation between two equally shaped 3-D images: ``(shift, corr_shifted)`` like the reference.

    ``shift`` is a float32 array (the signed location of max |corr|), ``corr_shifted = fftshift(|corr|)``.
    ``normalization`` is ``None``, ``"magnitude"`` or ``"classic"`` (estimate_stabilization.py:233-238).
    """
    if normalization not in _lib.PCC_NORM:
        raise ValueError(f"unknown normalization {normalization!r}")
    dev = resolve_device(device if not isinstance(ref_img, torch.Tensor) else ref_img.device)
    a, _, _ = as_device_volume(ref_img, dev)
    b, _, _ = as_device_volume(mov_img, dev)
    a, b = a.to(torch.float32), b.to(torch.float32)
    if a.ndim != 3 or a.shape != b.shape:
        raise ValueError(f"expected two 3-D images of one shape, got {tuple(a.shape)} and {tuple(b.shape)}")
    Z, Y, X = (int(s) for s in a.shape)
    ctx = get_context(dev)
    shift = (C.c_float * 3)()
    with torch.cuda.device(dev):
        # irfftn without a shape: an odd last axis comes back one shorter (the reference's behaviour, kept)
        corr = torch.empty((Z, Y, X - (X & 1)), dtype=torch.float32, device=dev)
        _lib.check(ctx.lib.bh_phase_cross_corr(ctx.handle, ptr(a), ptr(b), Z, Y, X, _lib.PCC_NORM[normalization],
                                               shift, ptr(corr)))
    return np.array(list(shift), dtype=np.float32), corr.cpu().numpy()
