import sys, time
sys.path.insert(0, '.')
import torch
from biahub_amd.deconvolve import PreparedRichardsonLucy
from biahub_amd.device import empty, alloc_layout
dev = torch.device("cuda", 0)
ax = [torch.arange(n, dtype=torch.float64, device=dev) - (n - 1) / 2 for n in (33, 17, 17)]
g = [torch.exp(-0.5 * (a / s) ** 2) for a, s in zip(ax, (3.0, 1.5, 1.5))]
psf = g[0][:, None, None] * g[1][None, :, None] * g[2][None, None, :]
psf = (psf / psf.sum()).float()
for shape in ((256, 1024, 1024), (128, 512, 512), (342, 1024, 1517), (256, 2048, 2048)):
    vol = empty(shape, torch.float32, dev).uniform_(90, 400)
    out = empty(shape, torch.float32, dev)
    with PreparedRichardsonLucy(psf, shape, dev) as h:
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            h(vol, 10, 1e-6, out=out); torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3
    print(f"RL x10 {shape}: {ms:.2f} ms  {alloc_layout()['chunk_kib']}", flush=True)
    del vol, out
