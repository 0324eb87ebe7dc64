#!/usr/bin/env python3
"""Flat field of a (512, 2048, 2048) uint16 stack: HIP-event time of the whole operator, third of three runs.
BH_FF_REREAD=1: the median kernel reads its columns twice (round-2 form)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from biahub_amd import _lib
from biahub_amd.device import get_context
from biahub_amd.flat_field import flat_field_device, median_z_device
dev = torch.device("cuda", 0)
ctx = get_context(dev); ctx.set_timing(True)
v = (torch.rand((512, 2048, 2048), device=dev) * 4000 + 100).to(torch.uint16)
for _ in range(3):
    out = flat_field_device(v); ms = ctx.elapsed_ms(_lib.T_FLATFIELD); del out
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for _ in range(3):
    ev[0].record(); m = median_z_device(v); ev[1].record(); torch.cuda.synchronize()
print(f"flat field {ms:.2f} ms ({8 * v.numel() / ms / 1e6 / 8000:.3f} of 8 TB/s by 8 B/voxel); median alone {ev[0].elapsed_time(ev[1]):.2f} ms")
