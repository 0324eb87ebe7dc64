import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from biahub_amd.deconvolve import PreparedRichardsonLucy
dev = torch.device("cuda", 0)
ax = [torch.arange(n, dtype=torch.float64, device=dev) - (n - 1) / 2 for n in (33, 17, 17)]
g = [torch.exp(-0.5 * (a / s) ** 2) for a, s in zip(ax, (3.0, 1.5, 1.5))]
psf = g[0][:, None, None] * g[1][None, :, None] * g[2][None, None, :]
psf = (psf / psf.sum()).float()
for shape in ((1068, 256, 1664),):
    vol = torch.rand(shape, device=dev) * 100
    out = torch.empty_like(vol)
    with PreparedRichardsonLucy(psf, shape, dev) as h:
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            h(vol, 10, 1e-6, out=out); torch.cuda.synchronize()
            print(f"RL x10 {shape} box {h.box} ({h.backend}): {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
