#!/bin/bash
# deskew + fill at config 2 with the input / output volumes from torch's default allocator, from device.volume_pool (shuffled
# 2-MiB chunks), and mixed: fresh processes, alternating
for i in 1 2 3; do
  for mode in plain pooled in_pooled out_pooled; do
    python - $mode <<'PY' 2>/dev/null
import sys
sys.path.insert(0, '.')
import torch
from biahub_amd import _lib
from biahub_amd.device import get_context, volume_pool, ptr
mode = sys.argv[1]
dev = torch.device("cuda", 0)
ctx = get_context(dev); ctx.set_timing(True)
import contextlib
pin = volume_pool(dev) if mode in ("pooled", "in_pooled") else contextlib.nullcontext()
pout = (lambda: volume_pool(dev)) if mode in ("pooled", "out_pooled") else (lambda: contextlib.nullcontext())
with pin:
    vol = (torch.rand((512, 2048, 2048), device=dev) * 300 + 100).round_()
import os
os.environ["BH_VOLUME_POOL"] = "0"  # the operator's own allocation stays out of the pool: outputs are made here
pass
shape = (683, 2048, 3034)
res = []
for fill, fm in (("mean", 2), (0, 0)):
    for _ in range(4):
        with pout():
            out = torch.empty(shape, dtype=torch.float32, device=dev)
        mean = __import__("ctypes").c_float()
        _lib.check(ctx.lib.bh_deskew(ctx.handle, ptr(vol), _lib.DT_F32, 512, 2048, 2048, 36.17, 0.371, 1, 3, fm, 0.0, ptr(out), None))
        dk, fl = ctx.elapsed_ms(_lib.T_DESKEW), (ctx.elapsed_ms(_lib.T_FILL) if fm else 0.0)
        del out
    res.append(f"fill={fill}: deskew {dk:.2f} fill passes {fl:.2f}")
print(mode, "|", " | ".join(res), flush=True)
PY
  done
done
