#!/usr/bin/env python3
"""The CPU baseline at the HEADLINE size, once per round (too long for the default bench line): the oracle's Richardson-Lucy x10
+ deskew (fill mean) on one (512, 2048, 2048) float32 volume on all host cores -> one JSON object on stdout.
    python tools/cpu_baseline_full.py > profiles/rNN_cpu_baseline_full_size.json
Needs ~90 GB of host memory (the deskew oracle materialises its un-averaged intermediate like the reference does)."""
import json
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from oracle import oracle_np as O  # noqa: E402  (baseline only)

shape = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (512, 2048, 2048)
vol = O.synthetic_volume(shape, seed=7, n_blobs=64)
psf = O.gaussian_psf(bench.PSF_SHAPE, bench.PSF_SIGMA)
t0 = time.perf_counter()
rl = O.richardson_lucy_zyx(vol, psf, 10, 1e-6)
t1 = time.perf_counter()
print(f"R-L x10: {t1 - t0:.1f} s", file=sys.stderr, flush=True)
dk = O.fast_deskew_zyx(rl, bench.DESKEW["ls_angle_deg"], bench.DESKEW["px_to_scan_ratio"], True, bench.DESKEW["average_n_slices"],
                       bench.DESKEW["overhang_fill"])
t2 = time.perf_counter()
print(json.dumps({"shape": list(shape), "seconds": t2 - t0, "rl_seconds": t1 - t0, "deskew_seconds": t2 - t1,
                  "voxels_per_s": float(np.prod(shape) / (t2 - t0)), "cores": os.cpu_count(), "cpu_model": bench.cpu_model(),
                  "threads": {"scipy_fft_workers": os.cpu_count(), "torch": torch.get_num_threads()}, "kind": "port",
                  "deskewed_shape": list(dk.shape)}))
