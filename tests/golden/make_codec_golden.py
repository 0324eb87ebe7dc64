#!/usr/bin/env python3
"""Golden Blosc-1 streams for biahub_amd.codecs, made with the real c-blosc (libblosc 1.21.0, the library numcodecs 0.15.1
wraps — reference uv.lock:3160-3161 — and the chunk compressor of the OME-Zarr stores iohub writes) found on this image
at /opt/conda/lib/libblosc.so.1 and driven through ctypes.  Output: tests/golden/blosc_streams.npz with, per case, the
original bytes and the compressed stream.  Only data is committed; the library itself is not used by the product or tests.

    python tests/golden/make_codec_golden.py
"""
import ctypes
import sys
from pathlib import Path

import numpy as np

LIB = "/opt/conda/lib/libblosc.so.1"
OUT = Path(__file__).resolve().parent / "blosc_streams.npz"


def main():
    lib = ctypes.CDLL(LIB)
    lib.blosc_get_version_string.restype = ctypes.c_char_p
    lib.blosc_compress_ctx.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
    lib.blosc_decompress_ctx.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    rng = np.random.default_rng(7)

    def image(n, dtype):  # smooth camera-like counts + noise: compressible, like the chunks of a real store
        x = np.arange(n)
        v = np.floor(300 + 200 * np.sin(x / 37.0)) + rng.poisson(2, n)
        if np.dtype(dtype).kind == "f":
            return (v / 8.0).astype(dtype)
        return (v.astype(np.int64) & (2 ** (8 * np.dtype(dtype).itemsize) - 1 if np.dtype(dtype).itemsize < 8 else -1)).astype(dtype)

    cases = []  # (name, array, cname, clevel, shuffle, blocksize)
    for dtype in ("u1", "u2", "f4", "f8"):
        for shuffle in (0, 1, 2):
            cases.append((f"zstd1_{dtype}_s{shuffle}", image(6000, dtype), b"zstd", 1, shuffle, 0))
    for cname in (b"lz4", b"lz4hc", b"zlib", b"blosclz", b"zstd"):
        for shuffle in (1, 2):
            cases.append((f"{cname.decode()}5_u2_s{shuffle}", image(9011, "u2"), cname, 5, shuffle, 0))
    # several blocks + a leftover block (and a leftover that is not a multiple of 8 elements for bitshuffle)
    for shuffle in (0, 1, 2):
        cases.append((f"blocks_u2_s{shuffle}", image(12003, "u2"), b"zstd", 3, shuffle, 4096))
        cases.append((f"blocks_f4_s{shuffle}", image(5001, "f4"), b"lz4", 3, shuffle, 2048))
        cases.append((f"blosclz_blocks_u2_s{shuffle}", image(12003, "u2"), b"blosclz", 9, shuffle, 8192))
    # tiny buffers (below the 128-byte minimum: stored), small buffers (no split), empty
    for n in (0, 1, 7, 63, 64, 100, 127, 128, 200, 1000):
        cases.append((f"small{n}_u2_s1", image(n, "u2"), b"zstd", 1, 1, 0))
        cases.append((f"small{n}_u2_s2", image(n, "u2"), b"lz4", 1, 2, 0))
    # incompressible data (blocks or splits stored raw), clevel 0 (memcpy), highly compressible runs
    noise = rng.integers(0, 65536, 3000, dtype=np.uint16)
    for shuffle in (0, 1, 2):
        cases.append((f"noise_u2_s{shuffle}", noise, b"zstd", 1, shuffle, 0))
        cases.append((f"noise_lz4_u2_s{shuffle}", noise, b"lz4", 1, shuffle, 0))
        cases.append((f"clevel0_u2_s{shuffle}", image(5000, "u2"), b"zstd", 0, shuffle, 0))
        cases.append((f"zeros_u2_s{shuffle}", np.zeros(20000, np.uint16), b"zstd", 1, shuffle, 0))
        cases.append((f"zeros_blosclz_u2_s{shuffle}", np.zeros(20000, np.uint16), b"blosclz", 5, shuffle, 0))
    # LZ4 at the sizes the device codec (csrc/lz4.hip) meets: several 64-KiB / 256-KiB blocks, c-blosc splitting them into
    # `typesize` streams, long runs (match lengths far beyond one extension byte), stored streams beside compressed ones
    cases.append(("lz4big_u2_s2", image(300_000, "u2"), b"lz4", 1, 2, 0))
    cases.append(("lz4big_f4_s2", image(200_000, "f4"), b"lz4", 1, 2, 262144))
    cases.append(("lz4big_u2_s1", image(250_001, "u2"), b"lz4", 5, 1, 65536))
    runs = np.zeros(400_000, np.uint16)
    runs[::50_001] = 777
    cases.append(("lz4runs_u2_s2", runs, b"lz4", 1, 2, 0))
    mixed = np.concatenate([image(150_000, "u2"), rng.integers(0, 65536, 150_000, dtype=np.uint16)])
    cases.append(("lz4mixed_u2_s2", mixed, b"lz4", 1, 2, 131072))
    cases.append(("lz4mixed_u2_s0", mixed, b"lz4", 1, 0, 131072))
    # a chunk the size iohub writes for one camera plane (uint16 128x256), default numcodecs settings of iohub
    cases.append(("plane_u2_zstd1_bitshuffle", image(128 * 256, "u2").reshape(128, 256), b"zstd", 1, 2, 0))

    out = {"blosc_version": np.array(lib.blosc_get_version_string().decode())}
    for name, arr, cname, clevel, shuffle, blocksize in cases:
        raw = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
        dest = np.empty(raw.size + 16 + 4 * (raw.size // 64 + 64), np.uint8)
        n = lib.blosc_compress_ctx(clevel, shuffle, arr.dtype.itemsize, raw.size, raw.ctypes.data, dest.ctypes.data, dest.size,
                                   cname, blocksize, 1)
        assert n > 0, (name, n)
        back = np.empty_like(raw)
        assert lib.blosc_decompress_ctx(dest.ctypes.data, back.ctypes.data, back.size, 1) == raw.size
        assert np.array_equal(back, raw)
        out[f"{name}__raw"] = raw
        out[f"{name}__blosc"] = dest[:n].copy()
        out[f"{name}__typesize"] = np.array(arr.dtype.itemsize)
    np.savez_compressed(OUT, **out)
    print(f"{OUT}: {len(cases)} cases, {OUT.stat().st_size} bytes")


if __name__ == "__main__":
    sys.exit(main())
