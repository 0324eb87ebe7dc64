#!/usr/bin/env python3
"""Chunk-codec timings: a uint16 camera volume in an iohub-style Blosc (zstd level 1, bit shuffle) store, host path
(NumPy permutation on the I/O threads) against the device path (bh_blosc_filter / bh_blosc_unfilter), both zarr versions,
plus the kernels alone.    python tools/codec_bench.py [--shape 256 1024 1024]"""
import argparse, json, sys, tempfile, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from biahub_amd import codecs, io

ap = argparse.ArgumentParser()
ap.add_argument("--shape", type=int, nargs=3, default=[256, 1024, 1024])
ap.add_argument("--zc", type=int, default=32)
args = ap.parse_args()
dev = torch.device("cuda", 0)
Z, Y, X = args.shape
rng = np.random.default_rng(0)
vol = (rng.poisson(6, (Z, Y, X)) + 110 + (60 * np.sin(np.arange(X) / 50.0)).astype(np.int64)).astype(np.uint16)
dvol = torch.from_numpy(vol).to(dev)
out = {"volume": f"uint16 {tuple(args.shape)} = {vol.nbytes / 1e6:.0f} MB, chunks (1,1,{args.zc},{Y},{X}), blosc zstd-1 bitshuffle"}


def timed(fn, reps=2):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best, r


with tempfile.TemporaryDirectory(dir="/dev/shm") as tmp:
    for version, ratio in (("0.4", None), ("0.5", (1, 1, 4, 1, 1))):
        p = Path(tmp) / f"p{version}"
        io.create_empty_position(p, ["a", "b"], (1, 2, Z, Y, X), chunks=(1, 1, args.zc, Y, X), dtype=np.uint16, version=version,
                                 compressor="blosc", shards_ratio=ratio)
        arr = io.open_ome_zarr(p).data
        tw_h, _ = timed(lambda: arr.write_volume(0, 0, vol))
        tw_d, _ = timed(lambda: arr.write_volume_device(0, 1, dvol))
        tr_h, a = timed(lambda: torch.from_numpy(arr.read_volume(0, 0)).to(dev))
        tr_d, b = timed(lambda: arr.read_volume_device(0, 1, dev))
        assert torch.equal(a, dvol) and torch.equal(b, dvol)
        stored = sum(f.stat().st_size for f in (p / "0").rglob("*") if f.is_file())
        out[f"ngff_{version}" + ("_sharded" if ratio else "")] = {
            "write_host_s": round(tw_h, 3), "write_device_s": round(tw_d, 3), "read_host_to_gpu_s": round(tr_h, 3),
            "read_device_s": round(tr_d, 3), "read_device_MBps": round(vol.nbytes / tr_d / 1e6), "write_device_MBps": round(vol.nbytes / tw_d / 1e6),
            "stored_bytes_over_raw": round(stored / (2 * vol.nbytes), 3)}
    # the same store with the lz4 inner codec: the block codec runs on the GPU (csrc/lz4.hip), only compressed frames cross PCIe
    comp = {"id": "blosc", "cname": "lz4", "clevel": 1, "shuffle": 2, "blocksize": 0}
    p = Path(tmp) / "plz4"
    io.create_empty_position(p, ["a", "b"], (1, 2, Z, Y, X), chunks=(1, 1, args.zc, Y, X), dtype=np.uint16, version="0.4", compressor=comp)
    arr = io.open_ome_zarr(p).data
    tw_d, _ = timed(lambda: arr.write_volume_device(0, 1, dvol))
    tr_d, b = timed(lambda: arr.read_volume_device(0, 1, dev))
    assert torch.equal(b, dvol) and np.array_equal(arr.read_volume(0, 1), vol)
    import os
    os.environ["BH_LZ4_DEVICE"] = "0"
    tw_h, _ = timed(lambda: arr.write_volume_device(0, 0, dvol))
    tr_h, a = timed(lambda: arr.read_volume_device(0, 0, dev))
    del os.environ["BH_LZ4_DEVICE"]
    assert torch.equal(a, dvol)
    stored = sum(f.stat().st_size for f in (p / "0" / "0" / "1").rglob("*") if f.is_file())
    out["ngff_0.4_blosc_lz4"] = {"write_device_codec_s": round(tw_d, 3), "read_device_codec_s": round(tr_d, 3),
                                 "write_host_codec_s": round(tw_h, 3), "read_host_codec_s": round(tr_h, 3),
                                 "write_device_codec_MBps": round(vol.nbytes / tw_d / 1e6), "read_device_codec_MBps": round(vol.nbytes / tr_d / 1e6),
                                 "stored_bytes_over_raw": round(stored / vol.nbytes, 3)}
# kernels alone
src = dvol.view(torch.uint8).reshape(-1)
dst = torch.empty_like(src)
for mode, name in ((1, "shuffle"), (2, "bitshuffle")):
    for fn, tag in ((codecs.filter_device, "filter"), (codecs.unfilter_device, "unfilter")):
        fn(src, dst, 256 << 10, 2, mode)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10):
            fn(src, dst, 256 << 10, 2, mode)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        out[f"kernel_{name}_{tag}"] = {"ms": round(ms, 3), "GBps_read_plus_write": round(2 * src.numel() / ms / 1e6)}
# LZ4 on the device: permuted volume -> Blosc frames, and back
nchunks = Z // args.zc
cbytes = args.zc * Y * X * 2
filt = torch.empty_like(src)
for i in range(nchunks):
    codecs.filter_device(src[i * cbytes:(i + 1) * cbytes], filt[i * cbytes:(i + 1) * cbytes], 256 << 10, 2, 2)
packed, offs = codecs.blosc_lz4_compress_device(filt, nchunks, cbytes, 256 << 10, 2, 2)
t, (packed, offs) = timed(lambda: codecs.blosc_lz4_compress_device(filt, nchunks, cbytes, 256 << 10, 2, 2), reps=3)
out["kernel_lz4_compress_frames"] = {"ms": round(t * 1e3, 2), "GBps_in": round(src.numel() / t / 1e9, 1), "ratio": round(offs[-1] / src.numel(), 3)}
host = packed[: offs[-1]].cpu().numpy()
frames = [host[offs[i]: offs[i + 1]].tobytes() for i in range(nchunks)]
back = torch.empty_like(src)


def decode_all():
    for i, fr in enumerate(frames):
        codecs.blosc_lz4_decode_blocks_device(fr, back[i * cbytes:(i + 1) * cbytes])


t, _ = timed(decode_all, reps=3)
assert torch.equal(back, filt)
out["kernel_lz4_decompress_frames_incl_upload"] = {"ms": round(t * 1e3, 2), "GBps_out": round(src.numel() / t / 1e9, 1),
                                                   "note": "chunk by chunk: one upload, launch and synchronisation per frame"}
back.zero_()
t, _ = timed(lambda: codecs.blosc_lz4_decode_frames_device(frames, back, [i * cbytes for i in range(nchunks)]), reps=3)
assert torch.equal(back, filt)
out["kernel_lz4_decompress_volume_incl_upload"] = {"ms": round(t * 1e3, 2), "GBps_out": round(src.numel() / t / 1e9, 1),
                                                   "note": "the volume's frames in one upload and one launch (read_volume_device)"}
print(json.dumps(out, indent=1))
