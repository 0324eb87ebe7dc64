"""Time the deskew kernel configurations at full size (BH_DESKEW_CFG is read per launch)."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from biahub_amd import _lib
from biahub_amd.deskew import fast_deskew_zyx
from biahub_amd.device import get_context

dev = torch.device("cuda", 0)
shape = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (512, 2048, 2048)
vol = torch.rand(shape, device=dev) * 1000
ctx = get_context(dev); ctx.set_timing(True)
kw = dict(ls_angle_deg=36.17, px_to_scan_ratio=0.371, keep_overhang=True, average_n_slices=3)
ref = None
for dt in ("f32", "u16"):
    v = vol if dt == "f32" else vol.to(torch.int32).to(torch.uint16)
    for cfg in (0, 1, 2, 3, 4):
        os.environ["BH_DESKEW_CFG"] = str(cfg)
        ms = []
        for _ in range(4):
            out = fast_deskew_zyx(v, overhang_fill=0, **kw)
            ms.append(ctx.elapsed_ms(_lib.T_DESKEW))
        V, Vo = np.prod(shape), out.numel()
        nbytes = (4 if dt == "f32" else 2) * V + 4 * Vo
        if ref is None and dt == "f32": ref = out.clone()
        same = float((out - ref).abs().max()) if dt == "f32" else None
        print(f"{dt} cfg {cfg}: {min(ms):7.3f} ms  {nbytes/min(ms)/1e6:8.1f} GB/s  maxdiff_vs_cfg0={same}", flush=True)
        del out
os.environ.pop("BH_DESKEW_CFG")
out = fast_deskew_zyx(vol, overhang_fill="mean", **kw)
print("default cfg + fill mean: deskew %.3f ms, fill passes %.3f ms" % (ctx.elapsed_ms(_lib.T_DESKEW), ctx.elapsed_ms(_lib.T_FILL)))
