#!/usr/bin/env python3
"""R-L x10 on deskewed, awkward volumes through a prepared handle, three times each — the command profiles/r03*_rl_awkward* were
taken with:  rocprofv3 --kernel-trace --stats --output-format csv -d out -o rl -- python3 tools/rl_awkward_profile.py [big]"""
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
shapes = [(342, 1024, 1517)] + ([(683, 2048, 3034)] if "big" in sys.argv else [])
for shape in shapes:
    vol = torch.rand(shape, device=dev) * 100
    out = torch.empty_like(vol)
    with PreparedRichardsonLucy(psf, shape, dev) as h:
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            h(vol, 10, 1e-6, out=out); torch.cuda.synchronize()
            print(f"RL x10 {shape} box {h.box} ({h.backend}, real OTF {h.otf_is_real}): {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
    del vol, out
