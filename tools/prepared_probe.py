"""Timing of the prepared inverse filter: create (first: allocates; second: pooled block), apply, one-shot."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from biahub_amd.apply_inverse_transfer_function import PreparedInverseFilter, apply_inverse_transfer_function_zyx
dev = torch.device("cuda", 0)
shape = (512, 2048, 2048)
vol = torch.rand(shape, device=dev) * 100 + 50
H = torch.rand(shape, device=dev) + 0.1
def t(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return r, (time.perf_counter() - t0) * 1e3
for i in range(3):
    p, ms_c = t(lambda: PreparedInverseFilter(H, shape, 0, 1e-3, "f32", dev))
    _, ms_a = t(lambda: p(vol, True))
    _, ms_a2 = t(lambda: p(vol, True))
    p.close()
    _, ms_o = t(lambda: apply_inverse_transfer_function_zyx(vol, H, 0, 1e-3, True))
    print(f"round {i}: create {ms_c:.1f} ms, apply {ms_a:.1f} / {ms_a2:.1f} ms, one-shot {ms_o:.1f} ms")
