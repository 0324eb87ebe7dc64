import sys, time
sys.path.insert(0, '.')
import torch
from biahub_amd.device import empty
dev = torch.device('cuda', 0)
for name, t in (('pooled', empty((683, 2048, 3034), torch.float32, dev)), ('plain', torch.empty((683, 2048, 3034), dtype=torch.float32, device=dev))):
    for w in (3034, 1650, 1024, 512):
        v = t[:, :, :w]
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); v.fill_(2.0); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{name}: fill_ of [:, :, :{w}] of rows of 3034 floats: {dt * 1e3:.3f} ms = {t.shape[0] * t.shape[1] * w * 4 / dt / 1e12:.2f} TB/s", flush=True)
    del t
