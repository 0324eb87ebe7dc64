"""Property tests (hypothesis) for the chunk codecs and the zarr reader/writer: whatever the dtype, chunk grid, shard ratio,
compressor or Blosc parameters, a volume written is the volume read, and permutations invert."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from biahub_amd import codecs as C
from biahub_amd import io
from oracle import codec_np as OC

DT = st.sampled_from(["u1", "u2", "i2", "f4"])


@settings(max_examples=60, deadline=None)
@given(dtype=DT, n=st.integers(0, 5000), mode=st.sampled_from([0, 1, 2]), cname=st.sampled_from(["zstd", "lz4", "zlib"]),
       blocksize=st.sampled_from([0, 64, 96, 1000, 4096]), noise=st.booleans(), seed=st.integers(0, 2**16))
def test_blosc_round_trip_any_parameters(dtype, n, mode, cname, blocksize, noise, seed):
    rng = np.random.default_rng(seed)
    a = (rng.integers(0, 250, n) if noise else (np.arange(n) // 7) % 200).astype(dtype)
    ts = a.dtype.itemsize
    stream = C.blosc_compress(a, ts, cname, 1, mode, blocksize=blocksize)
    h = C.BloscHeader(stream)
    assert h.nbytes == a.nbytes and h.cbytes == len(stream)
    assert np.array_equal(C.blosc_decompress(stream), a.view(np.uint8).reshape(-1))
    hh, shuffled = C.blosc_decode_blocks(stream)
    if not hh.memcpyed and n:  # the product's native permutation and the NumPy oracle agree on the writer's own blocks
        assert np.array_equal(C.unfilter(shuffled, hh.nbytes, hh.blocksize, hh.typesize, hh.shuffle_mode),
                              OC.unfilter(shuffled, hh.nbytes, hh.blocksize, hh.typesize, hh.shuffle_mode))


@settings(max_examples=25, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(dtype=st.sampled_from(["u2", "f4", "u1"]), version=st.sampled_from(["0.4", "0.5"]),
       comp=st.sampled_from([None, "blosc", {"id": "zstd", "level": 1}, {"id": "gzip", "level": 1}]),
       zyx=st.tuples(st.integers(1, 9), st.integers(1, 12), st.integers(1, 14)),
       chunk=st.tuples(st.integers(1, 2), st.integers(1, 2), st.integers(1, 5), st.integers(1, 6), st.integers(1, 7)),
       ratio=st.one_of(st.none(), st.tuples(st.integers(1, 2), st.integers(1, 2), st.integers(1, 3), st.integers(1, 2), st.integers(1, 3))),
       seed=st.integers(0, 2**16))
def test_zarr_round_trip_any_layout(tmp_path_factory, dtype, version, comp, zyx, chunk, ratio, seed):
    if version == "0.4":
        ratio = None
        if isinstance(comp, dict) and comp["id"] == "gzip":
            comp = {"id": "zlib", "level": 1}
    rng = np.random.default_rng(seed)
    shape = (2, 2) + zyx
    path = tmp_path_factory.mktemp("z") / "p"
    io.create_empty_position(path, ["a", "b"], shape, chunks=chunk, dtype=np.dtype(dtype), version=version, compressor=comp,
                             shards_ratio=ratio)
    arr = io.open_ome_zarr(path).data
    vols = {}
    for t, c in ((0, 0), (1, 1), (0, 1)):
        vols[t, c] = (rng.integers(0, 200, zyx)).astype(dtype)
        arr[t, c] = vols[t, c]
    arr = io.open_ome_zarr(path).data
    for (t, c), v in vols.items():
        assert np.array_equal(arr[t, c], v), (t, c)
    assert not arr[1, 0].any()  # never written: fill value
