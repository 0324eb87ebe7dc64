"""Does the R-L iteration time depend on where its buffers sit?  One process, one physical allocation each, views at different
offsets: python tools/placement_probe.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from biahub_amd import _lib
from biahub_amd.device import get_context, ptr
from bench import PSF_SHAPE, PSF_SIGMA, gaussian_psf, synthetic_position
dev = torch.device("cuda", 0)
ctx = get_context(dev); ctx.set_timing(True)
shape = (512, 2048, 2048)
V = int(np.prod(shape))
d = synthetic_position(shape, 1, dev)
psf = gaussian_psf(PSF_SHAPE, PSF_SIGMA, dev)
slack = 64 << 20
big_out = torch.empty(V + slack // 4, dtype=torch.float32, device=dev)
big_in = torch.empty(V + slack // 4, dtype=torch.float32, device=dev)
def run(off_in, off_out, it=4):
    din = big_in[off_in // 4: off_in // 4 + V].view(shape); din.copy_(d)
    out = big_out[off_out // 4: off_out // 4 + V].view(shape)
    for _ in range(2):
        _lib.check(ctx.lib.bh_richardson_lucy(ctx.handle, ptr(din), ptr(psf), 33, 17, 17, *shape, it, 1e-6, ptr(out)))
        ms = ctx.elapsed_ms(_lib.T_RL_ITER)
    return ms
for off_out in (0, 4096, 65536, 1 << 20, (2 << 20) + 4096, (4 << 20) + 8192, 8 << 20, (16 << 20) + 256, 32 << 20):
    print(f"out +{off_out:>9d} B   in +0: {run(0, off_out):.3f} ms/iter", flush=True)
for off_in in (4096, 1 << 20, (2 << 20) + 4096, 8 << 20):
    print(f"out +0   in +{off_in:>9d} B: {run(off_in, 0):.3f} ms/iter", flush=True)
