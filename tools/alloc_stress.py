#!/usr/bin/env python3
"""Stress of the workspace allocator: Richardson-Lucy on alternating shapes, so that the spectrum scratch is freed and rebuilt
(often at the same virtual address) again and again, handles come and go, and torch's pool hands blocks back and forth — every
result must equal the first one of its shape bit for bit.  tools/alloc_stress.py [rounds]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from biahub_amd.deconvolve import richardson_lucy  # noqa: E402
from biahub_amd.device import alloc_layout, get_context  # noqa: E402

dev = torch.device("cuda", 0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(7)
shapes = [(16, 512, 1024), (8, 1024, 2048), (4, 2048, 1024), (512, 32, 1024), (64, 128, 512), (256, 32, 1024)]
cases = []
for s in shapes:
    pshape = tuple(int(min(2 * rng.integers(1, 5) + 1, n)) for n in s)
    cases.append((s, (rng.random(s) * 300).astype(np.float32), (rng.random(pshape) + 0.05).astype(np.float32)))
first = {}
bad = 0
for r in range(rounds):
    order = rng.permutation(len(cases))
    for k in order:
        s, vol, psf = cases[k]
        out = richardson_lucy(torch.from_numpy(vol).to(dev), torch.from_numpy(psf).to(dev), 2, 1e-6).cpu().numpy()
        if not np.isfinite(out).all() or (k in first and not np.array_equal(out, first[k])):
            bad += 1
            ref = first.get(k)
            print(f"round {r} shape {s}: MISMATCH finite={np.isfinite(out).all()} max|out|={np.abs(out).max():.4g}"
                  + (f" max|diff|={np.abs(out - ref).max():.4g}" if ref is not None else ""), flush=True)
        first.setdefault(k, out)
    if r % 5 == 4:
        get_context(dev).release_workspace()  # everything goes back to the driver and is rebuilt
print(f"{rounds} rounds x {len(cases)} shapes: {bad} mismatches; layout {alloc_layout()}", flush=True)
sys.exit(1 if bad else 0)
