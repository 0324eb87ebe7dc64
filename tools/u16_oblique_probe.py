import sys
import os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from biahub_amd import _lib
from biahub_amd.device import get_context
from biahub_amd.register import affine_device
dev = torch.device("cuda", 0)
ctx = get_context(dev); ctx.set_timing(True)
shape = (512, 2048, 2048)
vol = (torch.rand(shape, device=dev) * 60000).to(torch.uint16)
th = np.deg2rad(2.0)
ax = np.array([1.0, 0.4, 0.3]); ax /= np.linalg.norm(ax)
K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
ob = np.eye(4); ob[:3, :3] = 1.02 * (np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K); ob[:3, 3] = (3.5, -12.25, 20.75)
for _ in range(4):
    out = affine_device(vol, ob, shape, "linear"); ms = ctx.elapsed_ms(_lib.T_AFFINE)
print("u16 oblique", ms)
