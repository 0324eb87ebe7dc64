#!/usr/bin/env python3
"""BASELINE config 4 rehearsal: positions of a synthetic plate (T=4, C=2, 1024x1024x256) sharded by position,
each (t, c) uint16 volume run through flat-field -> Richardson-Lucy/Tikhonov deconvolve -> deskew -> phase-cross-correlation
drift estimate against t = 0 -> stabilize on the GPU (per-stage seconds are reported for rank 0).

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
from biahub_amd.deconvolve import PreparedRichardsonLucy, tikhonov_zyx, transfer_function_device
from biahub_amd.deskew import fast_deskew_zyx, _fast_deskew_czyx
from biahub_amd.estimate_stabilization import PreparedPhaseCrossCorr
from biahub_amd.flat_field import flat_field_device
from biahub_amd.register import affine_device

ap = argparse.ArgumentParser()
ap.add_argument("--positions", type=int, default=8)
ap.add_argument("--T", type=int, default=4)
ap.add_argument("--C", type=int, default=2)
ap.add_argument("--shape", type=int, nargs=3, default=[256, 1024, 1024])
ap.add_argument("--deconv", choices=["rl", "tikhonov"], default="rl")
ap.add_argument("--order", choices=["deconv-first", "deskew-first"], default="deconv-first",
                help="deskew-first deconvolves the deskewed (awkward-shaped) volume: fused engine at a padded box (DESIGN.md 2.3)")
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

_rl = {}


def richardson_lucy(vol, psf_, iterations, eps):
    """The plate's PSF is prepared once per volume shape (bh_richardson_lucy_create), as the reference computes its transfer
    function once per plate (biahub/deconvolve.py:140-149); every unit applies it."""
    key = tuple(vol.shape)
    if key not in _rl:
        _rl[key] = PreparedRichardsonLucy(psf_, key, dev)
    return _rl[key](vol, iterations, eps)


stage = {"flat_field": 0.0, "deconvolve": 0.0, "deskew": 0.0, "estimate_shift": 0.0, "stabilize": 0.0}


def timed(name, fn):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize(dev)
    stage[name] += time.perf_counter() - t0
    return out


def one_position(pos):
    gen = torch.Generator(device=dev).manual_seed(0xB1A0 + pos)
    n = 0
    base = (torch.rand(shape, generator=gen, device=dev) * 400 + 100)
    ref = {}
    for t in range(args.T):
        for c in range(args.C):
            # the same scene drifting by (0, -2t, 3t) voxels, camera counts as uint16
            raw = (torch.roll(base, (0, -2 * t, 3 * t), (0, 1, 2)) + 20 * c).round_().to(torch.uint16)
            flat = timed("flat_field", lambda: flat_field_device(raw))
            if args.order == "deconv-first":
                dec = timed("deconvolve", lambda: richardson_lucy(flat, psf, 10, 1e-6) if args.deconv == "rl"
                            else tikhonov_zyx(flat, tf, 1e-3))
                dsk = timed("deskew", lambda: fast_deskew_zyx(dec, **DK))
            else:
                dec = timed("deskew", lambda: fast_deskew_zyx(flat, **DK))
                dsk = timed("deconvolve", lambda: richardson_lucy(dec, psf, 10, 1e-6))
            if t == 0:  # the reference timepoint's spectrum is kept for the position's other timepoints
                ref[c] = timed("estimate_shift", lambda: PreparedPhaseCrossCorr(flat, device=dev))
                m = np.eye(4)
            else:  # drift of the raw volume against t = 0 (estimate_stabilization.py:259-310), applied in deskewed space
                sh, _ = timed("estimate_shift", lambda: ref[c](flat, "magnitude"))
                m = shifts[t]
                assert tuple(sh) == (0.0, 2.0 * t, -3.0 * t), sh
            stab = timed("stabilize", lambda: affine_device(dsk, m, tuple(dsk.shape), "linear"))
            n += V
            del raw, dec, dsk, stab
    torch.cuda.synchronize(dev)
    for h in ref.values():
        h.close()
    return n

one_position(10_000 + rank)  # warm-up: plans, OTF, allocator
for k in stage:
    stage[k] = 0.0
parallel.barrier(); torch.cuda.synchronize(dev)
t0 = time.perf_counter()
st = parallel.process_positions(range(args.positions), one_position, rank, world)
torch.cuda.synchronize(dev); parallel.barrier()
dt = parallel.max_over_ranks(time.perf_counter() - t0, dev)
rows = parallel.gather_stats(st, dev)
if rank == 0:
    chain = f"{args.deconv} deconvolve -> deskew" if args.order == "deconv-first" else "deskew -> rl deconvolve (deskewed shape)"
    out = {"workload": f"{args.positions} positions x T={args.T} x C={args.C} x {shape}: flat-field -> {chain} -> PCC drift -> stabilize",
           "n_gpus": world, "seconds": dt, "voxels_per_s_resident": sum(r.voxels for r in rows) / dt,
           "positions_done": sum(r.n_done for r in rows), "positions_failed": sum(r.n_failed for r in rows),
           "stage_seconds_rank0": {k: round(v, 4) for k, v in stage.items()}}
    # host boundary: numpy uint16 CZYX in -> numpy float32 CZYX out through the operator adapter (PCIe both ways)
    czyx = (np.random.default_rng(0).random((1,) + shape) * 400 + 100).astype(np.uint16)
    _fast_deskew_czyx(czyx, device=str(dev), **DK)
    t1 = time.perf_counter(); o = _fast_deskew_czyx(czyx, device=str(dev), **DK); t2 = time.perf_counter()
    out["deskew_adapter_host_boundary"] = {"seconds": t2 - t1, "voxels_per_s": V / (t2 - t1),
                                           "h2d_bytes": czyx.nbytes, "d2h_bytes": o.nbytes,
                                           "note": "pageable numpy in (uint16, 2 B/voxel), float32 out through pinned host blocks (biahub_amd.device.to_host)"}
    print(json.dumps(out))
if world > 1:
    torch.distributed.destroy_process_group()
