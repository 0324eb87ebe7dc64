"""Affine warp alone at the bench shape (for profiling): python tools/affine_probe.py [similarity|identity|shift|oblique ...]"""
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
from biahub_amd.device import empty
vol = empty(shape, torch.float32, dev).uniform_(0, 1000)
th = np.deg2rad(2.0)
c, s = np.cos(th), np.sin(th)
MATS = {
    "similarity": np.array([[1.02, 0, 0, 3.5], [0, 1.02 * c, -1.02 * s, -12.25], [0, 1.02 * s, 1.02 * c, 20.75], [0, 0, 0, 1.0]]),
    "identity": np.eye(4),
    "shift": np.array([[1, 0, 0, 2.25], [0, 1, 0, -7.5], [0, 0, 1, 11.125], [0, 0, 0, 1.0]]),
    # the same 2 deg / 1.02x similarity about an oblique axis (z couples with y and x: the staged-tile kernel)
    "oblique": None,
    "strong": None,
}
ax = np.array([1.0, 0.4, 0.3]); ax /= np.linalg.norm(ax)
K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
Rm = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
ob = np.eye(4); ob[:3, :3] = 1.02 * Rm; ob[:3, 3] = (3.5, -12.25, 20.75)
MATS["oblique"] = ob
# 20 degrees about the same axis: strong z coupling, the staged-tile kernel
th2 = np.deg2rad(20.0)
Rm2 = np.eye(3) + np.sin(th2) * K + (1 - np.cos(th2)) * K @ K
st = np.eye(4); st[:3, :3] = Rm2; st[:3, 3] = (30.0, -120.0, 200.0)
MATS["strong"] = st
names = [a for a in sys.argv[1:] if a in MATS] or ["similarity"]
for name in names:
    for interp in ("linear", "nearestneighbor"):
        if interp != "linear" and name != "similarity":
            continue
        for _ in range(4):
            out = affine_device(vol, MATS[name], shape, interp); ms = ctx.elapsed_ms(_lib.T_AFFINE)
        print(name, interp, f"{ms:.3f} ms")
x = torch.empty_like(vol)
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); x.copy_(vol); e1.record(); torch.cuda.synchronize()
print(f"torch copy_ {e0.elapsed_time(e1):.3f} ms")
