"""Time the individual operators at BASELINE sizes (device-resident, HIP events inside the library)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from biahub_amd import _lib
from biahub_amd.deconvolve import tikhonov_zyx, transfer_function_device, richardson_lucy
from biahub_amd.register import affine_device
from biahub_amd.device import get_context

dev = torch.device("cuda", 0)
ctx = get_context(dev); ctx.set_timing(True)

def gaussian_psf(shape, sigma):
    ax = [torch.arange(n, dtype=torch.float64, device=dev) - (n - 1) / 2 for n in shape]
    g = [torch.exp(-0.5 * (a / s) ** 2) for a, s in zip(ax, sigma)]
    p = g[0][:, None, None] * g[1][None, :, None] * g[2][None, None, :]
    return (p / p.sum()).float()

which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "tikhonov"):
    for shape in ((256, 1024, 1024), (512, 2048, 2048)):
        V = np.prod(shape)
        vol = torch.rand(shape, device=dev) * 100
        tf = transfer_function_device(gaussian_psf((33, 17, 17), (3, 1.5, 1.5)), shape, dev)
        print(f"transfer_function {shape}: {ctx.elapsed_ms(_lib.T_TF):.2f} ms")
        for _ in range(3):
            out = tikhonov_zyx(vol, tf, 1e-3); ms = ctx.elapsed_ms(_lib.T_TIKHONOV)
        print(f"tikhonov {shape}: {ms:.2f} ms  -> 50V/t = {50*V/ms/1e6:.0f} GB/s ({50*V/ms/1e6/8000:.1%} of 8 TB/s)")
        del vol, tf, out
        ctx.release_workspace()
if which in ("all", "affine"):
    th = np.deg2rad(2.0)
    M = np.array([[1.02, 0, 0, 3.5], [0, 1.02 * np.cos(th), -1.02 * np.sin(th), -12.25],
                  [0, 1.02 * np.sin(th), 1.02 * np.cos(th), 20.75], [0, 0, 0, 1.0]])
    for shape in ((256, 1024, 1024), (512, 2048, 2048)):
        V = np.prod(shape)
        vol = torch.rand(shape, device=dev) * 100
        for interp in ("linear", "nearestneighbor"):
            for _ in range(3):
                out = affine_device(vol, M, shape, interp); ms = ctx.elapsed_ms(_lib.T_AFFINE)
            print(f"affine {interp} {shape}: {ms:.3f} ms -> 8V/t = {8*V/ms/1e6:.0f} GB/s ({8*V/ms/1e6/8000:.1%})")
        M2 = np.eye(4); M2[:3, 3] = (0.5, -2.25, 3.0)
        for _ in range(3):
            out = affine_device(vol, M2, shape, "linear"); ms = ctx.elapsed_ms(_lib.T_AFFINE)
        print(f"affine translation {shape}: {ms:.3f} ms -> {8*V/ms/1e6:.0f} GB/s")
        del vol, out
if which in ("all", "flatfield"):
    from biahub_amd.flat_field import flat_field_device
    for shape, dt in (((1068, 256, 1664), torch.uint16), ((512, 2048, 2048), torch.uint16), ((512, 2048, 2048), torch.float32)):
        V = np.prod(shape)
        vol = (torch.rand(shape, device=dev) * 4000 + 100).to(dt)
        for _ in range(3):
            out = flat_field_device(vol); ms = ctx.elapsed_ms(_lib.T_FLATFIELD)
        b = (2 * vol.element_size() + 4) * V
        print(f"flat_field {shape} {dt}: {ms:.2f} ms -> {b/ms/1e6:.0f} GB/s algorithmic (2 reads + 1 f32 write)")
        del vol, out
