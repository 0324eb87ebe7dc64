import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from biahub_amd import _lib
from biahub_amd.deskew import fast_deskew_zyx
from biahub_amd.device import get_context
dev = torch.device("cuda", 0)
ctx = get_context(dev); ctx.set_timing(True)
vol = (torch.rand((512, 2048, 2048), device=dev) * 300 + 100).round_()
for dt in (torch.float32, torch.uint16):
    v = vol.to(dt)
    for fill in ("mean", 0):
        for _ in range(4):
            out = fast_deskew_zyx(v, 36.17, 0.371, True, 3, fill)
            dk, fl = ctx.elapsed_ms(_lib.T_DESKEW), (ctx.elapsed_ms(_lib.T_FILL) if fill == "mean" else 0.0)
            del out
        print(f"{dt} fill={fill}: deskew kernel {dk:.2f} ms, fill passes {fl:.2f} ms", flush=True)
