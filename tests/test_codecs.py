"""Chunk codecs (biahub_amd/codecs.py) against streams written by the real c-blosc 1.21.0 (tests/golden/blosc_streams.npz,
made by tests/golden/make_codec_golden.py) and against the published check values of CRC-32C."""
import ctypes
import os
from pathlib import Path

import numpy as np
import pytest

from biahub_amd import codecs as C
from oracle import codec_np as OC

GOLDEN = Path(__file__).parent / "golden"


def _cases():
    z = np.load(GOLDEN / "blosc_streams.npz")
    names = sorted(k[: -len("__blosc")] for k in z.files if k.endswith("__blosc"))
    return z, names


def test_blosc_decode_matches_c_blosc_streams():
    z, names = _cases()
    assert len(names) >= 60 and str(z["blosc_version"]) == "1.21.0"
    seen = set()
    for n in names:
        stream, raw, ts = z[f"{n}__blosc"], z[f"{n}__raw"], int(z[f"{n}__typesize"])
        h = C.BloscHeader(stream)
        assert (h.nbytes, h.typesize, h.cbytes) == (raw.size, ts, stream.size), n
        got = C.blosc_decompress(stream)
        assert np.array_equal(got, raw), n
        # two-stage form used by the device path: entropy decode, then the permutation on its own
        hh, shuffled = C.blosc_decode_blocks(stream)
        if not hh.memcpyed:  # product (native host code) and oracle (NumPy) both undo what the real library did
            assert np.array_equal(C.unfilter(shuffled, hh.nbytes, hh.blocksize, hh.typesize, hh.shuffle_mode), raw), n
            assert np.array_equal(OC.unfilter(shuffled, hh.nbytes, hh.blocksize, hh.typesize, hh.shuffle_mode), raw), n
        seen.add((h.codec, h.shuffle_mode, h.memcpyed, -(-h.nbytes // max(1, h.blocksize)) > 1))
    # the fixture really covers every inner codec, both permutations, stored buffers and multi-block streams
    assert {c for c, *_ in seen} >= {"zstd", "lz4", "zlib", "blosclz"}
    assert {m for _, m, *_ in seen} == {0, 1, 2}
    assert any(s[2] for s in seen) and any(s[3] for s in seen)


@pytest.mark.parametrize("typesize", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("mode", [1, 2])
def test_native_host_permutations_match_oracle(typesize, mode):
    """libbhcore's host code (bh_host_blosc_filter / _unfilter) against the NumPy restatement, ragged cases included."""
    rng = np.random.default_rng(typesize + 10 * mode)
    for nbytes, blocksize in ((100_003, 4096 * typesize), (70_001, 70_001), (typesize * 8 * 37 + typesize - 1, 1 << 20),
                              (100, 64), (3, 1 << 10), (12 * typesize, 5 * typesize)):
        raw = rng.integers(0, 256, nbytes, dtype=np.uint8)
        want = OC.filter_blocks(raw, blocksize, typesize, mode)
        got = C.filter_host(raw, blocksize, typesize, mode)
        assert np.array_equal(got, want), (nbytes, blocksize)
        assert np.array_equal(C.unfilter(want, nbytes, blocksize, typesize, mode), raw)
        assert np.array_equal(OC.unfilter(want, nbytes, blocksize, typesize, mode), raw)


def test_blosc_rejects_truncated_and_corrupt_streams():
    z, _ = _cases()
    s = z["zstd1_u2_s1__blosc"]
    with pytest.raises(ValueError):
        C.blosc_decompress(s[:10])
    with pytest.raises(ValueError):
        C.blosc_decompress(s[: s.size // 2])


@pytest.mark.parametrize("typesize", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("n", [0, 5, 64, 1000, 1003])
def test_permutations_invert(typesize, n):
    rng = np.random.default_rng(n * 31 + typesize)
    b = rng.integers(0, 256, n, dtype=np.uint8)
    assert np.array_equal(OC.unshuffle(OC.shuffle(b, typesize), typesize), b)
    assert np.array_equal(OC.bitunshuffle(OC.bitshuffle(b, typesize), typesize), b)


def test_bitshuffle_layout_known_answer():
    # 8 uint16 elements whose only set bit is bit i of element i: plane 8 j + k holds element k's bit -> one-hot bytes
    v = (1 << np.arange(8)).astype("<u2")
    out = OC.bitshuffle(v.view(np.uint8), 2)
    want = np.zeros(16, np.uint8)
    want[:8] = 1 << np.arange(8)  # low byte planes 0..7: plane k has element k set, at bit k (LSB first)
    assert np.array_equal(out, want)


@pytest.mark.parametrize("cname", ["zstd", "lz4", "zlib"])
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("dtype,n", [("u2", 100_003), ("f4", 40_000), ("u1", 5000), ("u2", 60), ("u2", 0)])
def test_blosc_writer_round_trip_and_is_readable_by_c_blosc(cname, mode, dtype, n):
    rng = np.random.default_rng(3)
    arr = (300 + 50 * np.sin(np.arange(n) / 20) + rng.poisson(3, n)).astype(dtype)
    stream = C.blosc_compress(arr, arr.dtype.itemsize, cname, 1, mode, blocksize=4096 * arr.dtype.itemsize)
    assert np.array_equal(C.blosc_decompress(stream), arr.view(np.uint8))
    if n > 1000 and dtype == "u2" and mode:
        assert len(stream) < arr.nbytes
    lib_path = "/opt/conda/lib/libblosc.so.1"  # the real library, when this image has it: it must accept our streams
    if os.path.exists(lib_path):
        lib = ctypes.CDLL(lib_path)
        lib.blosc_decompress_ctx.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        noise = rng.integers(0, 256, max(arr.nbytes, 1), dtype=np.uint8)[: arr.nbytes].view(arr.dtype)  # stored blocks / buffer
        mixed = np.concatenate([noise, arr])
        for a in (arr, noise, mixed):
            src = np.frombuffer(C.blosc_compress(a, a.dtype.itemsize, cname, 1, mode, blocksize=4096 * a.dtype.itemsize), np.uint8).copy()
            back = np.empty(a.nbytes, np.uint8)
            assert lib.blosc_decompress_ctx(src.ctypes.data, back.ctypes.data, back.size, 1) == a.nbytes
            assert np.array_equal(back, a.view(np.uint8))


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("n", [720, 100_000])
def test_blosc_writer_incompressible_data_is_stored_unpermuted(mode, n):
    noise = np.random.default_rng(4).integers(1, 60000, n).astype(np.uint16)
    stream = C.blosc_compress(noise, 2, "zstd", 1, mode)
    h = C.BloscHeader(stream)
    assert h.memcpyed and len(stream) == noise.nbytes + 16
    assert np.array_equal(C.blosc_decompress(stream), noise.view(np.uint8))
    # half noise, half zeros: compressible as a whole, with incompressible blocks stored inside the stream
    mixed = np.concatenate([noise, np.zeros(n, np.uint16)])
    stream = C.blosc_compress(mixed, 2, "zstd", 1, mode, blocksize=1024)
    assert not C.BloscHeader(stream).memcpyed
    assert np.array_equal(C.blosc_decompress(stream), mixed.view(np.uint8))


def test_crc32c_check_values():
    assert C.crc32c(b"123456789") == 0xE3069283          # the CRC catalogue's check value for CRC-32C
    assert C.crc32c(b"\x00" * 32) == 0x8A9136AA          # RFC 3720 B.4
    assert C.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert C.crc32c(bytes(range(32))) == 0x46DD794E
    assert C.crc32c(b"6789", C.crc32c(b"12345")) == 0xE3069283


def test_plain_codecs_round_trip():
    raw = np.frombuffer((b"camera counts " * 500), np.uint8)
    for kind in ("zstd", "zlib", "gzip", "lz4", "crc32c"):
        c = C.ChunkCodec(kind, level=3)
        assert np.array_equal(c.decode(c.encode(raw), raw.size), raw), kind
    assert C.zstd_frame_content_size(C.zstd_compress(raw, 1)) == raw.size
