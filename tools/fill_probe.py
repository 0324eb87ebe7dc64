"""Fill-pass probe: bh_overhang_fill on volumes with a large zero wedge, aligned vs unaligned row lengths."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from biahub_amd import _lib
from biahub_amd.device import get_context, ptr
dev = torch.device("cuda", 0)
ctx = get_context(dev); ctx.set_timing(True)
for X in (3034, 3072, 3040):
    Z, Y = 683, 2048
    v = torch.rand((Z, Y, X), device=dev) + 1.0
    v[:, :, : X // 2] = 0          # half of every row masked: one contiguous run per row
    for _ in range(3):
        w = v.clone()
        _lib.check(ctx.lib.bh_overhang_fill(ctx.handle, ptr(w), Z, Y, X, _lib.FILL_MEAN, 0.0, 3, None))
        ms = ctx.elapsed_ms(_lib.T_FILL)
    print(f"X={X}: fill passes {ms:.2f} ms for {w.numel()*4/1e9:.1f} GB volume, {(w == 0).sum().item()} zeros left")
