"""R-L x10 on deskewed-like volumes whose rows fit 5 * 2^k: the 3 * 2^k box (default) against the 5 * 2^k one (BH_RL_X5=1)."""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from biahub_amd.deconvolve import PreparedRichardsonLucy  # noqa: E402

dev = torch.device("cuda", 0)
ax = [torch.arange(n, dtype=torch.float64, device=dev) - (n - 1) / 2 for n in (33, 17, 17)]
g = [torch.exp(-0.5 * (a / s) ** 2) for a, s in zip(ax, (3.0, 1.5, 1.5))]
psf = g[0][:, None, None] * g[1][None, :, None] * g[2][None, None, :]
psf = (psf / psf.sum()).float()
for shape in ((342, 1024, 1100), (342, 1024, 2100)):
    vol = torch.rand(shape, device=dev) * 100
    out = torch.empty_like(vol)
    res = {}
    for x5 in ("", "1"):
        if x5:
            os.environ["BH_RL_X5"] = x5
        else:
            os.environ.pop("BH_RL_X5", None)
        with PreparedRichardsonLucy(psf, shape, dev) as h:
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                h(vol, 10, 1e-6, out=out)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) * 1e3
            print(f"RL x10 {shape} box {h.box} ({h.backend}) BH_RL_X5={x5!r}: {ms:.1f} ms", flush=True)
            res[x5] = out.clone()
    err = float((res[""] - res["1"]).abs().max() / res["1"].abs().max())
    print(f"  max relative difference between the two boxes: {err:.2e}", flush=True)
