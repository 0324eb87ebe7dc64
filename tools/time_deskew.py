#!/usr/bin/env python3
"""Deskew + fill passes alone at the bench shape (HIP-event times of the library's own timers, fifth of five runs).
BHCORE_LIB=<variant .so> / BH_DESKEW_PERS=0 select what runs: tools/build_variant.py NAME --src=deskew.hip -D..."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from biahub_amd import _lib
from biahub_amd.deskew import fast_deskew_zyx
from biahub_amd.device import get_context

dev = torch.device("cuda", 0)
ctx = get_context(dev)
ctx.set_timing(True)
shape = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (512, 2048, 2048)
vol = (torch.rand(shape, device=dev) * 300 + 100).round_()
for fill in ("mean", 0):
    for _ in range(5):
        out = fast_deskew_zyx(vol, 36.17, 0.371, True, 3, fill)
        dk, fl = ctx.elapsed_ms(_lib.T_DESKEW), (ctx.elapsed_ms(_lib.T_FILL) if fill == "mean" else 0.0)
        del out
    print(f"fill={fill}: deskew kernel {dk:.2f} ms, fill passes {fl:.2f} ms", flush=True)
