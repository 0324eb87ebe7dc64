"""Affine warp alone at the bench shape (for profiling): python tools/affine_probe.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from biahub_amd import _lib
from biahub_amd.device import get_context
from biahub_amd.register import affine_device
dev = torch.device("cuda", 0)
ctx = get_context(dev); ctx.set_timing(True)
shape = (512, 2048, 2048)
vol = torch.rand(shape, device=dev) * 1000
th = np.deg2rad(2.0)
M = np.array([[1.02, 0, 0, 3.5], [0, 1.02 * np.cos(th), -1.02 * np.sin(th), -12.25], [0, 1.02 * np.sin(th), 1.02 * np.cos(th), 20.75], [0, 0, 0, 1.0]])
for interp in ("linear", "nearestneighbor"):
    for _ in range(4):
        out = affine_device(vol, M, shape, interp); ms = ctx.elapsed_ms(_lib.T_AFFINE)
    print(interp, f"{ms:.3f} ms")
