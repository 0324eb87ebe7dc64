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
        vol16 = vol.to(torch.uint16)
        for _ in range(3):
            out = affine_device(vol16, M, shape, "linear"); ms = ctx.elapsed_ms(_lib.T_AFFINE)
        print(f"affine linear uint16 in {shape}: {ms:.3f} ms -> 6V/t = {6*V/ms/1e6:.0f} GB/s ({6*V/ms/1e6/8000:.1%})")
        del vol16
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
if which in ("all", "psf"):
    import time
    from biahub_amd.characterize_psf import block_peaks, detect_peaks, recentre_beads, _patch_margins
    from biahub_amd.estimate_psf import average_beads_device, BEAD_DETECTION_SETTINGS
    shape = (256, 1024, 1024)
    g = torch.Generator(device=dev).manual_seed(3)
    vol = torch.empty(shape, device=dev).normal_(110.0, 3.0, generator=g)
    n = 400
    cz, cy, cx = (torch.randint(30, s - 30, (n,), generator=g, device=dev) for s in shape)
    off = torch.arange(-6, 7, device=dev)
    dz, dy, dx = torch.meshgrid(off, off, off, indexing="ij")
    val = 2000 * torch.exp(-0.5 * ((dz / 1.5) ** 2 + (dy / 1.2) ** 2 + (dx / 1.2) ** 2))
    vol.index_put_((cz[:, None, None, None] + dz, cy[:, None, None, None] + dy, cx[:, None, None, None] + dx),
                   val.expand(n, -1, -1, -1), accumulate=True)
    torch.cuda.synchronize()
    for _ in range(2):
        t0 = time.perf_counter(); block_peaks(vol, 3, (64, 64, 32)); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"block_peaks {shape} blur 3 block (64,64,32): {(t1-t0)*1e3:.2f} ms")
    t0 = time.perf_counter(); peaks = detect_peaks(vol, **BEAD_DETECTION_SETTINGS, device=dev); t1 = time.perf_counter()
    margins = _patch_margins((1, 1, 1), (31, 41, 41))
    centres = recentre_beads(vol, peaks, margins); torch.cuda.synchronize(); t2 = time.perf_counter()
    psf, nb = average_beads_device(vol, centres, margins); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"detect_peaks: {len(peaks)} peaks in {(t1-t0)*1e3:.1f} ms; recentre {len(centres)} beads {(t2-t1)*1e3:.1f} ms; "
          f"average {nb} patches of (31,41,41): {(t3-t2)*1e3:.1f} ms")
if which in ("all", "deskew"):
    from biahub_amd.deskew import fast_deskew_zyx
    kw = dict(ls_angle_deg=36.17, px_to_scan_ratio=0.371, keep_overhang=True, average_n_slices=3)
    for shape in ((256, 1024, 1024), (512, 2048, 2048)):
        V = np.prod(shape)
        base = (torch.rand(shape, device=dev) * 400 + 100).round_()
        for dt in (torch.float32, torch.uint16):
            vol = base.to(dt)
            for fill in ("mean", 0):
                for _ in range(3):
                    out = fast_deskew_zyx(vol, overhang_fill=fill, **kw); ms = ctx.elapsed_ms(_lib.T_DESKEW)
                fl = ctx.elapsed_ms(_lib.T_FILL) if fill == "mean" else 0.0
                b = vol.element_size() * V + 4 * out.numel()
                print(f"deskew {shape} {dt} fill={fill}: kernel {ms:.2f} ms ({b/ms/1e6:.0f} GB/s algorithmic), fill passes {fl:.2f} ms")
            del vol
        del base, out
