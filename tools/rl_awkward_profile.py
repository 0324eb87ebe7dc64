#!/usr/bin/env python3
"""R-L x10 on the deskewed, awkward (342,1024,1517) volume, three times — the command profiles/r01v_* were taken with:
   rocprofv3 --kernel-trace --stats --output-format csv -d out -o rl -- python3 tools/rl_awkward_profile.py"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from biahub_amd.deconvolve import richardson_lucy, richardson_lucy_plan

dev = torch.device("cuda", 0)
ax = [torch.arange(n, dtype=torch.float64, device=dev) - (n - 1) / 2 for n in (33, 17, 17)]
g = [torch.exp(-0.5 * (a / s) ** 2) for a, s in zip(ax, (3.0, 1.5, 1.5))]
psf = g[0][:, None, None] * g[1][None, :, None] * g[2][None, None, :]
psf = (psf / psf.sum()).float()
shape = (342, 1024, 1517)
vol = torch.rand(shape, device=dev) * 100
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = richardson_lucy(vol, psf, 10, 1e-6); torch.cuda.synchronize()
    print(f"RL x10 {shape} plan {richardson_lucy_plan(psf.shape, shape)}: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
