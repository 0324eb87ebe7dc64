#!/usr/bin/env python3
"""Headline benchmark: raw-input voxels/s for Richardson-Lucy (10 iterations) + deskew of a
(512, 2048, 2048) float32 volume per GPU (BASELINE.json configs[1]), positions sharded over ranks.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one synthetic position resident in HBM: the reference's
pipeline order (README.md:140-148) deconvolve (raw coordinates) -> deskew, with the deskew settings
of settings/example_deskew_settings.yml (36.17 deg, 0.371, keep_overhang, N=3, overhang_fill mean).
Rank 0 prints ONE JSON line.  Every rank processes its own position (weak scaling, no data-path
collective); only the timing barrier and the max-over-ranks reduction use RCCL.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from biahub_amd import _lib, parallel  # noqa: E402
from biahub_amd.deconvolve import richardson_lucy  # noqa: E402
from biahub_amd.deskew import fast_deskew_zyx, get_deskewed_data_shape  # noqa: E402
from biahub_amd.device import get_context  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
DESKEW = dict(ls_angle_deg=36.17, px_to_scan_ratio=0.371, keep_overhang=True, average_n_slices=3,
              overhang_fill="mean")
PSF_SHAPE, PSF_SIGMA = (33, 17, 17), (3.0, 1.5, 1.5)


def gaussian_psf(shape, sigma, device):
    ax = [torch.arange(n, dtype=torch.float64, device=device) - (n - 1) / 2 for n in shape]
    g = [torch.exp(-0.5 * (a / s) ** 2) for a, s in zip(ax, sigma)]
    psf = g[0][:, None, None] * g[1][None, :, None] * g[2][None, None, :]
    return (psf / psf.sum()).to(torch.float32)


def synthetic_position(shape, seed, device):
    """Camera-count-like volume generated on device: offset 110 + Gaussian noise + sparse bright blobs."""
    g = torch.Generator(device=device).manual_seed(seed)
    Z, Y, X = shape
    vol = torch.empty(shape, dtype=torch.float32, device=device)
    vol.normal_(110.0, 4.0, generator=g)
    n_blobs = max(16, (Z * Y * X) // 2**19)
    zz = torch.randint(2, Z - 2, (n_blobs,), generator=g, device=device)
    yy = torch.randint(2, Y - 2, (n_blobs,), generator=g, device=device)
    xx = torch.randint(2, X - 2, (n_blobs,), generator=g, device=device)
    amp = torch.rand((n_blobs,), generator=g, device=device) * 3800 + 200
    for dz, dy, dx, w in [(0, 0, 0, 1.0), (1, 0, 0, 0.6), (-1, 0, 0, 0.6), (0, 1, 0, 0.6), (0, -1, 0, 0.6),
                          (0, 0, 1, 0.6), (0, 0, -1, 0.6)]:
        vol.index_put_((zz + dz, yy + dy, xx + dx), amp * w, accumulate=True)
    return vol.round_().clamp_(0, 65535)


def pmc_traffic(key, shape):
    """HBM bytes per launch from the committed PMC passes (profiles/r01_pmc_hbm_traffic.json: separate
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this command, fetch doubled for gfx950 as
    MI355X_MICROARCH.md prescribes).  Only valid for the shape it was collected on."""
    f = ROOT / "profiles" / "r01_pmc_hbm_traffic.json"
    if not f.exists():
        return None
    rec = json.load(open(f))
    if list(shape) != rec.get("shape"):
        return None
    return rec.get(key)


def cpu_baseline(iterations):
    """The oracle (CPU restatement) timed on a bounded sample of the same workload, rank 0 only."""
    from oracle import oracle_np as O  # checker / baseline only — never the thing measured as `value`

    shape = (192, 768, 768)
    vol = O.synthetic_volume(shape, seed=7, n_blobs=32)
    psf = O.gaussian_psf(PSF_SHAPE, PSF_SIGMA)
    t0 = time.perf_counter()
    rl = O.richardson_lucy_zyx(vol, psf, iterations, 1e-6)
    O.fast_deskew_zyx(rl, DESKEW["ls_angle_deg"], DESKEW["px_to_scan_ratio"], True, DESKEW["average_n_slices"],
                      DESKEW["overhang_fill"])
    dt = time.perf_counter() - t0
    return {
        "value": float(np.prod(shape) / dt),
        "unit": "voxels/s",
        "cores": os.cpu_count(),
        "kind": "port",
        "sample": f"oracle RL({iterations} it, scipy.fft workers=all) + deskew(fill mean) on one {shape} float32 "
                  f"volume, {dt:.1f} s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--shape", type=int, nargs=3, default=[512, 2048, 2048], metavar=("Z", "Y", "X"))
    ap.add_argument("--iterations", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank, local_rank, world = parallel.world_info()
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    local_rank %= max(1, torch.cuda.device_count())  # one rank per GPU; wraps only in a BH_DIST_BACKEND=gloo rehearsal
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    parallel.init("nccl", dev)  # RCCL; used only for the timing barrier / max-over-ranks, never on the data path
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # one synthetic position per rank and step: position index = rank (round-robin shard of a `world`-position plate)
    my_positions = parallel.shard_positions(range(world), rank, world)
    assert my_positions == [rank]

    shape = tuple(args.shape)
    V = int(np.prod(shape))
    out_shape, _ = get_deskewed_data_shape(shape, DESKEW["ls_angle_deg"], DESKEW["px_to_scan_ratio"], True,
                                           DESKEW["average_n_slices"])
    V_out = int(np.prod(out_shape))
    psf = gaussian_psf(PSF_SHAPE, PSF_SIGMA, dev)
    vol = synthetic_position(shape, 0xB1A0 + rank, dev)  # position index = rank (weak scaling)
    ctx = get_context(dev)
    ctx.set_timing(True)

    def step():
        rl = richardson_lucy(vol, psf, args.iterations, 1e-6)
        return fast_deskew_zyx(rl, **DESKEW)

    def fence():
        torch.cuda.synchronize(dev)
        parallel.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        out = step()
        del out
    rl_ms, dk_ms, fill_ms = [], [], []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
        # per-kernel HIP-event timings recorded by the library on ITS stream (reading them syncs that stream,
        # which the step's result needs anyway before the next position overwrites the workspace)
        rl_ms.append(ctx.elapsed_ms(_lib.T_RL_ITER))
        dk_ms.append(ctx.elapsed_ms(_lib.T_DESKEW))
        fill_ms.append(ctx.elapsed_ms(_lib.T_FILL))
        del out
    fence()
    dt = time.perf_counter() - t0
    dt = parallel.max_over_ranks(dt, dev)

    if rank == 0:
        rl_iter_s = float(np.mean(rl_ms)) / 1e3
        deskew_s = float(np.mean(dk_ms)) / 1e3
        fill_s = float(np.mean(fill_ms)) / 1e3
        rl_bytes = 112.0 * V            # 4 real 3-D FFTs (3-pass model) + fused pointwise
        dk_bytes = 4.0 * (V + V_out)    # read every input voxel once, write every output voxel once
        result = {
            "metric": "voxels/s for deskew+10-iter R-L deconv, 2048^2x512 f32",
            "value": world * args.steps * V / dt,
            "unit": "voxels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"1 position/GPU: R-L {args.iterations} it (PSF {PSF_SHAPE}) then deskew "
                            f"{shape}->{tuple(out_shape)} (36.17 deg, 0.371, N=3, fill mean), input resident in HBM",
                "raw_shape_zyx": list(shape),
                "deskewed_shape_zyx": list(out_shape),
                "positions_per_step": world,
            },
            "roofline": {
                "kernel": "one Richardson-Lucy iteration = 8 in-place passes of csrc/fftconv.hip: 2 x (col_pass Y fwd, "
                          "col_pass Z fwd*OTF*inv, col_pass Y inv, x_inv_kernel<FUSE>: inverse X + RL epilogue + next forward X)",
                "bound": "hbm",
                "achieved": rl_bytes / rl_iter_s / 1e9,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": rl_bytes / rl_iter_s / 1e9 / HBM_PEAK_GBS,
                "traffic": pmc_traffic("rl_iteration", shape),
                "algorithmic_bytes": rl_bytes,
                "ms": rl_iter_s * 1e3,
            },
            "roofline_deskew": {
                "kernel": "deskew_kernel (fused shear-interpolate + N-mean)",
                "bound": "hbm",
                "achieved": dk_bytes / deskew_s / 1e9,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": dk_bytes / deskew_s / 1e9 / HBM_PEAK_GBS,
                "traffic": pmc_traffic("deskew_kernel", shape),
                "algorithmic_bytes": dk_bytes,
                "ms": deskew_s * 1e3,
                "fill_passes_ms": fill_s * 1e3,
            },
            "workspace_gb": ctx.workspace_bytes() / 1e9,
        }
        if not args.no_cpu_baseline and world == 1:  # reported on rank 0 at N=1 only
            result["cpu_baseline"] = cpu_baseline(args.iterations)
        print(json.dumps(result), flush=True)
    if world > 1:
        parallel.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
