#!/usr/bin/env python3
"""BASELINE config 4 rehearsal: positions of a synthetic plate (T=4, C=2, 1024x1024x256) sharded by position,
each (t, c) volume run through deskew -> Richardson-Lucy/Tikhonov deconvolve -> stabilize on the GPU.

    python tools/plate_bench.py --positions 8                     # one GPU
    torchrun --nproc-per-node 8 tools/plate_bench.py --positions 64

Reports device-resident throughput and the host-boundary (PCIe-inclusive) throughput of the numpy operator
adapters (`func(czyx) -> czyx`, what `process_single_position` drives).
"""
import argparse, json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from biahub_amd import parallel
from biahub_amd.deconvolve import richardson_lucy, tikhonov_zyx, transfer_function_device
from biahub_amd.deskew import fast_deskew_zyx, _fast_deskew_czyx
from biahub_amd.register import affine_device

ap = argparse.ArgumentParser()
ap.add_argument("--positions", type=int, default=8)
ap.add_argument("--T", type=int, default=4)
ap.add_argument("--C", type=int, default=2)
ap.add_argument("--shape", type=int, nargs=3, default=[256, 1024, 1024])
ap.add_argument("--deconv", choices=["rl", "tikhonov"], default="rl")
args = ap.parse_args()

rank, local_rank, world = parallel.world_info()
local_rank %= max(1, torch.cuda.device_count())
torch.cuda.set_device(local_rank)
dev = torch.device("cuda", local_rank)
parallel.init(None, dev)
shape = tuple(args.shape)
V = int(np.prod(shape))
DK = dict(ls_angle_deg=36.17, px_to_scan_ratio=0.371, keep_overhang=True, average_n_slices=3, overhang_fill="mean")
ax = [torch.arange(n, dtype=torch.float64, device=dev) - (n - 1) / 2 for n in (33, 17, 17)]
g = [torch.exp(-0.5 * (a / s) ** 2) for a, s in zip(ax, (3.0, 1.5, 1.5))]
psf = (g[0][:, None, None] * g[1][None, :, None] * g[2][None, None, :])
psf = (psf / psf.sum()).float()
tf = transfer_function_device(psf, shape, dev) if args.deconv == "tikhonov" else None
shifts = [np.eye(4)] * args.T
for t in range(args.T):
    m = np.eye(4); m[:3, 3] = (0.25 * t, -1.5 * t, 2.25 * t); shifts[t] = m

def one_position(pos):
    gen = torch.Generator(device=dev).manual_seed(0xB1A0 + pos)
    n = 0
    for t in range(args.T):
        for c in range(args.C):
            vol = (torch.rand(shape, generator=gen, device=dev) * 400 + 100).round_()
            dec = richardson_lucy(vol, psf, 10, 1e-6) if args.deconv == "rl" else tikhonov_zyx(vol, tf, 1e-3)
            dsk = fast_deskew_zyx(dec, **DK)
            stab = affine_device(dsk, shifts[t], tuple(dsk.shape), "linear")
            n += V
            del vol, dec, dsk, stab
    torch.cuda.synchronize(dev)
    return n

one_position(10_000 + rank)  # warm-up: plans, OTF, allocator
parallel.barrier(); torch.cuda.synchronize(dev)
t0 = time.perf_counter()
st = parallel.process_positions(range(args.positions), one_position, rank, world)
torch.cuda.synchronize(dev); parallel.barrier()
dt = parallel.max_over_ranks(time.perf_counter() - t0, dev)
rows = parallel.gather_stats(st, dev)
if rank == 0:
    out = {"workload": f"{args.positions} positions x T={args.T} x C={args.C} x {shape}: {args.deconv} deconvolve -> deskew -> stabilize",
           "n_gpus": world, "seconds": dt, "voxels_per_s_resident": sum(r.voxels for r in rows) / dt,
           "positions_done": sum(r.n_done for r in rows), "positions_failed": sum(r.n_failed for r in rows)}
    # host boundary: numpy uint16 CZYX in -> numpy float32 CZYX out through the operator adapter (PCIe both ways)
    czyx = (np.random.default_rng(0).random((1,) + shape) * 400 + 100).astype(np.uint16)
    _fast_deskew_czyx(czyx, device=str(dev), **DK)
    t1 = time.perf_counter(); o = _fast_deskew_czyx(czyx, device=str(dev), **DK); t2 = time.perf_counter()
    out["deskew_adapter_host_boundary"] = {"seconds": t2 - t1, "voxels_per_s": V / (t2 - t1),
                                           "h2d_bytes": czyx.nbytes, "d2h_bytes": o.nbytes,
                                           "note": "pageable numpy memory, torch copies; uint16 in (2 B/voxel), float32 out"}
    print(json.dumps(out))
if world > 1:
    torch.distributed.destroy_process_group()
