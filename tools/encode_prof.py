#!/usr/bin/env python3
"""Where encode_volume_device spends its time for a deskewed float32 volume going to a Blosc-lz4 store (steady state: third of
four volumes onward): permutation, LZ4 + frame assembly, offsets read-back, pinned allocation, download."""
import sys, time, tempfile
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from biahub_amd import io, codecs
from biahub_amd.device import to_host

dev = torch.device("cuda", 0)
shape = (1, 1, 342, 1024, 1517)
comp = {"id": "blosc", "cname": "lz4", "clevel": 1, "shuffle": 2, "blocksize": 0}
with tempfile.TemporaryDirectory(dir="/dev/shm") as tmp:
    io.create_empty_position(Path(tmp) / "p", ["a"], shape, chunks=(1, 1, int(sys.argv[1]) if len(sys.argv) > 1 else 32, 1024, 1517), dtype=np.float32, version="0.4", compressor=comp)
    arr = io.open_ome_zarr(Path(tmp) / "p").data
    g = torch.Generator(device=dev).manual_seed(1)
    for k in range(5):
        if "deskew" in sys.argv:  # what the CLI bench stores: the deskewed (interpolated, mean-filled) camera-like stack
            from biahub_amd.deskew import fast_deskew_zyx
            raw = (torch.poisson(torch.full((256, 1024, 1024), 6.0, device=dev), generator=g) + 110
                   + (60 * torch.sin(torch.arange(1024, device=dev) / 50.0)).floor()).to(torch.uint16)
            vol = fast_deskew_zyx(raw, ls_angle_deg=36.17, px_to_scan_ratio=0.371, keep_overhang=True, average_n_slices=3, overhang_fill="mean")
            if not isinstance(vol, torch.Tensor):
                vol = torch.from_numpy(vol).to(dev)
            assert tuple(vol.shape) == shape[2:], vol.shape
        else:
            vol = (torch.empty(shape[2:], device=dev).normal_(110, 4, generator=g) + 60 * torch.sin(torch.arange(shape[-1], device=dev) / 50.0)).round()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        commit = arr.encode_volume_device(0, 0, vol)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        commit()
        t2 = time.perf_counter()
        print(f"volume {k}: encode_volume_device {t1 - t0:.3f} s, host half (file writes) {t2 - t1:.3f} s", flush=True)
    # the pieces, steady state
    v8 = vol.view(torch.uint8).reshape(-1)
    zc = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    cbytes = zc * 1024 * 1517 * 4
    nch = -(-342 // zc)
    stage = torch.zeros(nch * cbytes, dtype=torch.uint8, device=dev)
    pad = torch.zeros(cbytes, dtype=torch.uint8, device=dev)
    def tm(f, n=3):
        best = 1e9
        for _ in range(n):
            torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        return best, r
    bsz = codecs.default_blocksize(4)
    t, _ = tm(lambda: [codecs.filter_device(v8[i * cbytes:(i + 1) * cbytes] if (i + 1) * cbytes <= v8.numel() else pad, stage[i * cbytes:(i + 1) * cbytes], bsz, 4, 2) for i in range(nch)])
    print(f"filter {nch} chunks: {t * 1e3:.1f} ms")
    t, (packed, offs) = tm(lambda: codecs.blosc_lz4_compress_device(stage, nch, cbytes, bsz, 4, 2))
    print(f"lz4 + frames: {t * 1e3:.1f} ms -> {offs[-1] / 1e6:.0f} MB of {stage.numel() / 1e6:.0f}")
    for _ in range(3):
        t0 = time.perf_counter(); h = to_host(packed[: offs[-1]]); t1 = time.perf_counter()
        print(f"to_host: {1e3 * (t1 - t0):.1f} ms"); del h
