"""Affine warp of a (512,2048,2048) float32 volume for rotations of growing angle about an oblique axis and about x (which kernel
takes them, how long): python tools/affine_angle_sweep.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from biahub_amd import _lib
from biahub_amd.device import empty, get_context
from biahub_amd.register import affine_device
dev = torch.device("cuda", 0)
ctx = get_context(dev); ctx.set_timing(True)
shape = (512, 2048, 2048)
vol = empty(shape, torch.float32, dev).uniform_(0, 1000)
c0 = np.array([(n - 1) / 2 for n in shape])
def rot(axis, deg):
    ax = np.asarray(axis, float); ax /= np.linalg.norm(ax)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    th = np.deg2rad(deg)
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    M = np.eye(4); M[:3, :3] = R; M[:3, 3] = c0 - R @ c0  # about the volume's centre
    return M
for name, axis in (("oblique axis (1, .4, .3)", (1.0, 0.4, 0.3)), ("z axis (in-plane)", (1.0, 0, 0)), ("y axis (couples z, x)", (0, 1.0, 0)), ("x axis (couples z, y)", (0, 0, 1.0))):
    for deg in (1, 2, 5, 10, 20, 45, 90):
        for _ in range(3):
            out = affine_device(vol, rot(axis, deg), shape, "linear"); ms = ctx.elapsed_ms(_lib.T_AFFINE)
        del out
        print(f"{name:28s} {deg:3d} deg: {ms:8.3f} ms", flush=True)
