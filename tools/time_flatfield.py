#!/usr/bin/env python3
"""Flat field of a (512, 2048, 2048) uint16 stack: HIP-event time of the whole operator and of the median alone, third of
three runs, on three kinds of data: "wide" (uniform 100..4100: every pixel spans 12 bits along z), "camera" (offset 110 +
Gaussian noise sigma 4 + sparse bright blobs: most pixels span < 256 levels), "signal" (a smooth structure along z of a few
hundred to a few thousand counts with Poisson-like noise).  BHCORE_LIB selects the library (A/B against a variant)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from biahub_amd import _lib
from biahub_amd.device import get_context
from biahub_amd.flat_field import flat_field_device, median_z_device
dev = torch.device("cuda", 0)
ctx = get_context(dev); ctx.set_timing(True)
shape = (512, 2048, 2048)
g = torch.Generator(device=dev).manual_seed(3)
def make(kind):
    if kind == "wide":
        return (torch.rand(shape, device=dev, generator=g) * 4000 + 100).to(torch.uint16)
    if kind == "camera":
        v = torch.empty(shape, device=dev).normal_(110.0, 4.0, generator=g)
        n = 4096
        idx = tuple(torch.randint(0, s, (n,), device=dev, generator=g) for s in shape)
        v.index_put_(idx, torch.rand(n, device=dev, generator=g) * 3800 + 200, accumulate=True)
        return v.clamp_(0, 65535).to(torch.uint16)
    z = torch.linspace(-1, 1, shape[0], device=dev).view(-1, 1, 1)
    amp = torch.rand((1, shape[1], shape[2]), device=dev, generator=g) * 2500 + 300
    v = amp * torch.exp(-4 * z * z) + 100
    v += torch.randn(shape, device=dev, generator=g) * v.sqrt()
    return v.clamp_(0, 65535).to(torch.uint16)
for kind in ("wide", "camera", "signal"):
    v = make(kind)
    for _ in range(3):
        out = flat_field_device(v); ms = ctx.elapsed_ms(_lib.T_FLATFIELD); del out
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for _ in range(3):
        ev[0].record(); m = median_z_device(v); ev[1].record(); torch.cuda.synchronize()
    print(f"{kind:7s} flat field {ms:.2f} ms ({8 * v.numel() / ms / 1e6 / 8000:.3f} of 8 TB/s by 8 B/voxel); median alone {ev[0].elapsed_time(ev[1]):.2f} ms", flush=True)
    del v, m
