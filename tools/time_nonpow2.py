import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from biahub_amd import _lib
from biahub_amd.deconvolve import tikhonov_zyx, transfer_function_device, richardson_lucy
from biahub_amd.device import get_context
dev = torch.device("cuda", 0)
ctx = get_context(dev); ctx.set_timing(True)
def gaussian_psf(shape, sigma):
    ax = [torch.arange(n, dtype=torch.float64, device=dev) - (n - 1) / 2 for n in shape]
    g = [torch.exp(-0.5 * (a / s) ** 2) for a, s in zip(ax, sigma)]
    p = g[0][:, None, None] * g[1][None, :, None] * g[2][None, None, :]
    return (p / p.sum()).float()
psf = gaussian_psf((33, 17, 17), (3, 1.5, 1.5))
for shape in ((342, 1024, 1517), (256, 1024, 1024), (384, 1024, 1536), (384, 1024, 1024), (1068, 256, 1664)):
    V = np.prod(shape)
    vol = torch.rand(shape, device=dev) * 100
    import os
    from biahub_amd.deconvolve import richardson_lucy_plan
    res = {}
    for force, name in (("0", "library FFT, 7-smooth pad-and-fold"), ("1", "fused engine, wrap-padded power-of-two box"), (None, "default")):
        if force is None:
            os.environ.pop("BH_RL_ENGINE_PAD", None)
        else:
            os.environ["BH_RL_ENGINE_PAD"] = force
        try:
            for _ in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                out = richardson_lucy(vol, psf, 10, 1e-6); torch.cuda.synchronize()
                dt = time.perf_counter() - t0
        except Exception as e:  # the forced engine box can be outside the engine's range
            print(f"RL x10 {shape} [{name}]: {type(e).__name__}: {e}", flush=True)
            continue
        res[name] = out.clone()
        print(f"RL x10 {shape} [{name}] plan {richardson_lucy_plan(psf.shape, shape)}: {dt*1e3:.1f} ms -> {V/dt/1e9:.2f} Gvox/s", flush=True)
        ctx.release_workspace()
    ks = list(res)
    if len(ks) >= 2:
        print(f"   max |engine - library| / max = {float((res[ks[0]] - res[ks[1]]).abs().max() / res[ks[0]].abs().max()):.2e}", flush=True)
    res.clear()
    tf = transfer_function_device(psf, shape, dev)
    for _ in range(2):
        out = tikhonov_zyx(vol, tf, 1e-3); ms = ctx.elapsed_ms(_lib.T_TIKHONOV)
    print(f"tikhonov {shape}: {ms:.1f} ms -> {V/ms/1e6:.2f} Gvox/s", flush=True)
    del vol, out, tf
    ctx.release_workspace()
