#!/usr/bin/env python3
"""Cubic (SciPy order 3) warp of a (512, 2048, 2048) float32 volume: HIP-event time of the whole operator and, under
rocprofv3 --kernel-trace --stats, the split between the prefilter passes and the gather.  Argument: angle in degrees about z
(default 2) — the registration-sized rotation of the ops table."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from biahub_amd import _lib
from biahub_amd.device import get_context
from biahub_amd.register import affine_device
dev = torch.device("cuda", 0)
ctx = get_context(dev); ctx.set_timing(True)
shape = (512, 2048, 2048)
vol = torch.rand(shape, device=dev)
ang = np.deg2rad(float(sys.argv[1]) if len(sys.argv) > 1 else 2.0)
c, s = np.cos(ang), np.sin(ang)
M = np.eye(4); M[1, 1] = c; M[1, 2] = -s; M[2, 1] = s; M[2, 2] = c
ctr = np.array(shape) / 2
M[:3, 3] = ctr - M[:3, :3] @ ctr + np.array([1.5, -3.25, 2.75])
for mode in ("cubic", "linear"):
    b = _lib.BOUNDARY_SCIPY_CONSTANT if mode == "cubic" else 0
    for _ in range(3):
        out = affine_device(vol, M, shape, mode, b) if mode == "cubic" else affine_device(vol, M, shape, mode)
        ms = ctx.elapsed_ms(_lib.T_AFFINE); del out
    print(f"{mode} {ms:.2f} ms ({8 * vol.numel() / ms / 1e6:.0f} GB/s by 8 B/voxel)")
