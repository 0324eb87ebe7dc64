import sys, time
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from biahub_amd.estimate_stabilization import phase_cross_corr_device
dev = torch.device("cuda", 0)
vol = torch.rand((512, 2048, 2048), device=dev) * 1000
mov = torch.roll(vol, (1, -3, 17), (0, 1, 2))
for norm in (None, "magnitude"):
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sh, _ = phase_cross_corr_device(vol, mov, norm, want_corr=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(norm, sh, f"{dt*1e3:.2f} ms")
