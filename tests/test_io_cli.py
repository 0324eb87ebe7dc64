"""I/O layer, per-position driver and CLI surface.  CPU tests use numpy operators; GPU tests run the real steps."""

import json
from pathlib import Path

import numpy as np
import pytest
from click.testing import CliRunner

from biahub_amd import io
from biahub_amd.cli import cli, expand_eat_all
from conftest import rel_err
from oracle import oracle_np as O

DESKEW_YML = ("pixel_size_um: 0.116\nls_angle_deg: 36.17\npx_to_scan_ratio: 0.371\nscan_step_um: 0.313\n"
              "keep_overhang: true\naverage_n_slices: 3\noverhang_fill: mean\n")


def make_plate(path, positions=(("A", "1", "0"), ("B", "2", "0")), shape=(2, 2, 8, 12, 16), dtype=np.uint16, seed=0,
               compressor=None, scale=(1, 1, 1, 0.116, 0.116)):
    rng = np.random.default_rng(seed)
    io.create_empty_plate(path, positions, [f"ch{c}" for c in range(shape[1])], shape, scale=scale, dtype=dtype,
                          compressor=compressor)
    data = {}
    for key in positions:
        pos = io.open_ome_zarr(path.joinpath(*key))
        for t in range(shape[0]):
            for c in range(shape[1]):
                v = (rng.random(shape[2:]) * 1000 + 100).astype(dtype)
                pos.data[t, c] = v
                data[key + (t, c)] = v
    return data


def test_zarr_roundtrip_and_layout(tmp_path):
    for comp in (None, {"id": "zlib", "level": 1}):
        store = tmp_path / f"p_{'z' if comp else 'raw'}.zarr"
        data = make_plate(store, compressor=comp)
        assert json.loads((store / ".zattrs").read_text())["plate"]["wells"][0]["path"] == "A/1"
        assert (store / "A" / "1" / "0" / "0" / ".zarray").exists()  # row/col/fov/array "0"
        pos = io.open_ome_zarr(store / "B" / "2" / "0")
        assert pos.channel_names == ["ch0", "ch1"] and pos.data.shape == (2, 2, 8, 12, 16)
        assert pos.scale[-2:] == [0.116, 0.116] and pos.version == "0.4"
        for (r, c_, f, t, c), v in data.items():
            got = io.open_ome_zarr(store / r / c_ / f).data[t, c]
            assert got.dtype == np.uint16 and np.array_equal(got, v)
    io.create_empty_plate(store, [("A", "1", "0")], ["ch0", "ch1"], (2, 2, 8, 12, 16), dtype=np.uint16,
                          compressor={"id": "zlib", "level": 1})  # idempotent
    assert np.array_equal(io.open_ome_zarr(store / "A" / "1" / "0").data[0, 0], data[("A", "1", "0", 0, 0)])
    meta = json.loads((store / "A" / "1" / "0" / "0" / ".zarray").read_text())
    meta["compressor"] = {"id": "lzma"}
    (store / "A" / "1" / "0" / "0" / ".zarray").write_text(json.dumps(meta))
    with pytest.raises(NotImplementedError, match="lzma"):
        io.open_ome_zarr(store / "A" / "1" / "0")


@pytest.mark.parametrize("version,compressor,chunks,shards_ratio", [
    ("0.4", "blosc", None, None),                                         # what iohub writes for NGFF 0.4
    ("0.4", {"id": "blosc", "cname": "lz4", "clevel": 5, "shuffle": 1}, (1, 1, 3, 5, 7), None),  # ragged grid on every axis
    ("0.4", {"id": "zstd", "level": 3}, (2, 2, 4, 12, 16), None),        # chunks spanning several (t, c): read-modify-write
    ("0.5", None, None, None),
    ("0.5", "blosc", (1, 1, 4, 6, 8), None),
    ("0.5", "blosc", (1, 1, 4, 6, 8), (1, 1, 2, 2, 2)),                   # sharded, one (t, c) per shard
    ("0.5", {"id": "zstd", "level": 1}, (1, 1, 3, 5, 7), (2, 2, 3, 3, 3)),  # shards spanning (t, c) and overhanging the array
    ("0.5", {"id": "gzip", "level": 1}, (2, 1, 8, 12, 16), (1, 2, 1, 1, 1)),
])
def test_zarr_v2_v3_codecs_sharding_round_trip(tmp_path, version, compressor, chunks, shards_ratio):
    store = tmp_path / "p.zarr"
    shape = (2, 2, 8, 12, 16)
    rng = np.random.default_rng(5)
    names = ["ch0", "ch1"]
    io.create_empty_plate(store, [("A", "1", "0"), ("B", "2", "0")], names, shape, chunks=chunks, scale=(1, 1, 0.3, 0.1, 0.1),
                          dtype=np.uint16, version=version, compressor=compressor, shards_ratio=shards_ratio)
    pos = io.open_ome_zarr(store / "A/1/0")
    assert np.array_equal(pos.data[1, 1], np.zeros(shape[2:], np.uint16))  # nothing written yet: fill value
    data = {}
    for t in range(2):
        for c in range(2):
            data[t, c] = (rng.poisson(3, shape[2:]) + 100 * t + 10 * c).astype(np.uint16)
            pos.data[t, c] = data[t, c]
    pos = io.open_ome_zarr(store / "A/1/0")  # re-open: metadata round trip
    assert pos.version == version and pos.channel_names == names and pos.scale[2:] == [0.3, 0.1, 0.1]
    assert pos.data.zarr_format == (2 if version == "0.4" else 3) and pos.data.sharded == bool(shards_ratio)
    for (t, c), v in data.items():
        got = pos.data[t, c]
        assert got.dtype == np.uint16 and np.array_equal(got, v), (t, c)
    pos.data[0, 1] = data[1, 0]  # overwrite one volume: the neighbours in shared chunks / shards stay
    assert np.array_equal(pos.data[0, 1], data[1, 0]) and np.array_equal(pos.data[0, 0], data[0, 0])
    assert np.array_equal(pos.data[1, 1], data[1, 1])
    pos.update_zattrs({"extra_metadata": {"k": 1}})
    assert io.open_ome_zarr(store / "A/1/0").zattrs["extra_metadata"] == {"k": 1}
    if version == "0.5":
        root = json.loads((store / "zarr.json").read_text())
        assert root["node_type"] == "group" and root["attributes"]["ome"]["version"] == "0.5"
        assert root["attributes"]["ome"]["plate"]["wells"][1]["path"] == "B/2"
        arr = json.loads((store / "A/1/0/0/zarr.json").read_text())
        assert arr["data_type"] == "uint16" and arr["dimension_names"] == ["T", "C", "Z", "Y", "X"]
        assert (arr["codecs"][0]["name"] == "sharding_indexed") == bool(shards_ratio)
        assert (store / "A/1/0/0/c/0/0/0/0/0").exists()
    with pytest.raises(ValueError, match="zarr v"):
        io.create_empty_plate(store, [("C", "3", "0")], names, shape, version="0.5" if version == "0.4" else "0.4")


def test_zarr_v3_shard_layout_and_index_checksum(tmp_path):
    """The shard file is what the zarr v3 sharding spec says: inner chunks, then (offset, nbytes) uint64 pairs in C order
    of the inner grid (2^64 - 1 twice for an absent chunk), then the CRC-32C of those pairs."""
    from biahub_amd import codecs

    io.create_empty_position(tmp_path / "p", ["a"], (1, 1, 4, 4, 8), chunks=(1, 1, 2, 4, 4), dtype=np.uint8, version="0.5",
                             shards_ratio=(1, 1, 2, 1, 2))
    pos = io.open_ome_zarr(tmp_path / "p")
    vol = np.arange(4 * 4 * 8, dtype=np.uint8).reshape(4, 4, 8)
    pos.data[0, 0] = vol
    raw = (tmp_path / "p/0/c/0/0/0/0/0").read_bytes()
    assert len(raw) == 4 * 32 + 4 * 16 + 4
    index = np.frombuffer(raw[128:-4], "<u8").reshape(1, 1, 2, 1, 2, 2)
    assert codecs.crc32c(raw[128:-4]) == int.from_bytes(raw[-4:], "little")
    for kz in range(2):
        for kx in range(2):
            off, nb = index[0, 0, kz, 0, kx]
            assert nb == 32
            assert np.array_equal(np.frombuffer(raw[off:off + nb], np.uint8).reshape(2, 4, 4), vol[2 * kz:2 * kz + 2, :, 4 * kx:4 * kx + 4])
    corrupt = bytearray(raw)
    corrupt[130] ^= 1
    (tmp_path / "p/0/c/0/0/0/0/0").write_bytes(bytes(corrupt))
    with pytest.raises(OSError, match="checksum"):
        pos.data[0, 0]


def test_reads_chunks_written_by_c_blosc(tmp_path):
    """A v2 array whose chunk files are the golden streams of the real c-blosc (as a numcodecs/iohub writer leaves them)."""
    z = np.load(Path(__file__).parent / "golden" / "blosc_streams.npz")
    raw = z["plane_u2_zstd1_bitshuffle__raw"].view("<u2").reshape(128, 256)
    arr = tmp_path / "fov" / "0"
    arr.mkdir(parents=True)
    (tmp_path / "fov" / ".zgroup").write_text('{"zarr_format": 2}')
    (arr / ".zarray").write_text(json.dumps({
        "zarr_format": 2, "shape": [1, 2, 1, 128, 256], "chunks": [1, 1, 1, 128, 256], "dtype": "<u2", "order": "C",
        "compressor": {"id": "blosc", "cname": "zstd", "clevel": 1, "shuffle": 2, "blocksize": 0}, "fill_value": 0,
        "filters": None, "dimension_separator": "/"}))
    (arr / "0/0/0/0").mkdir(parents=True)
    (arr / "0/0/0/0/0").write_bytes(z["plane_u2_zstd1_bitshuffle__blosc"].tobytes())
    a = io.ZarrArray(arr)
    assert np.array_equal(a[0, 0], raw[None])
    assert np.array_equal(a[0, 1], np.zeros((1, 128, 256), np.uint16))  # chunk never written


def test_process_single_position_contract(tmp_path):
    src = tmp_path / "in.zarr"
    data = make_plate(src, positions=(("A", "1", "0"),), shape=(3, 2, 4, 5, 6), dtype=np.float32)
    io.open_ome_zarr(src / "A/1/0").data[1, 1] = np.zeros((4, 5, 6), np.float32)  # an empty frame is skipped
    dst = tmp_path / "out.zarr"
    io.create_empty_plate(dst, [("A", "1", "0")], ["ch0", "ch1"], (3, 2, 4, 5, 6))
    calls = []

    def op(czyx, gain, input_time_index):  # declares input_time_index -> it is injected (stabilize.py:35)
        calls.append((input_time_index, czyx.shape))
        return czyx * gain + input_time_index

    n = io.process_single_position(op, src / "A/1/0", dst / "A/1/0", gain=2.0, resume=True, resume_token="tok",
                                   extra_metadata={"biahub-test": {"gain": 2.0}})
    assert n == 5 and all(s == (1, 4, 5, 6) for _, s in calls)  # one channel per call, one unit skipped
    out = io.open_ome_zarr(dst / "A/1/0")
    assert np.allclose(out.data[2, 0], data[("A", "1", "0", 2, 0)] * 2 + 2)
    assert not out.data[1, 1].any()
    assert out.zattrs["extra_metadata"]["biahub-test"] == {"gain": 2.0}
    calls.clear()
    assert io.process_single_position(op, src / "A/1/0", dst / "A/1/0", gain=2.0, resume=True, resume_token="tok") == 0
    assert io.process_single_position(op, src / "A/1/0", dst / "A/1/0", gain=2.0, resume=True, resume_token="new") == 5
    n = io.process_single_position(lambda czyx: czyx[::-1], src / "A/1/0", dst / "A/1/0",
                                   input_channel_indices=[[0, 1]], output_channel_indices=[[0, 1]],
                                   input_time_indices=[0], output_time_indices=[2])
    assert n == 1 and np.array_equal(io.open_ome_zarr(dst / "A/1/0").data[2, 0], data[("A", "1", "0", 0, 1)])


def test_cli_help_and_eat_all():
    assert expand_eat_all(["deskew", "-i", "a", "b", "c", "-c", "x.yml", "-o", "o"]) == \
        ["deskew", "-i", "a", "-i", "b", "-i", "c", "-c", "x.yml", "-o", "o"]
    r = CliRunner()
    for cmd in ("deskew", "deconvolve", "rl-deconvolve", "register", "stabilize", "flip", "estimate-registration", "flat-field", "estimate-psf", "estimate-stabilization", "process-with-config", "optimize-registration"):
        res = r.invoke(cli, [cmd, "--help"])
        assert res.exit_code == 0 and "Usage" in res.output


def test_cli_deskew_init_on_cpu(tmp_path):
    """``--init`` needs no GPU: geometry via the host-only C-ABI call, plate creation, RESOURCES contract."""
    src = tmp_path / "in.zarr"
    make_plate(src, shape=(2, 1, 16, 24, 20))
    cfg = tmp_path / "deskew.yml"
    cfg.write_text(DESKEW_YML)
    out = tmp_path / "out.zarr"
    args = expand_eat_all(["deskew", "-i", str(src / "A/1/0"), str(src / "B/2/0"), "-c", str(cfg), "-o", str(out), "--init"])
    res = CliRunner().invoke(cli, args)
    assert res.exit_code == 0, res.output
    line = [l for l in res.output.splitlines() if l.startswith("RESOURCES:")][0]
    assert set(json.loads(line[len("RESOURCES:"):])) == {"cpus", "mem_gb", "time_minutes"}
    want, voxel = O.get_deskewed_data_shape((16, 24, 20), 36.17, 0.371, True, 3, 0.116)
    pos = io.open_ome_zarr(out / "B/2/0")
    assert pos.data.shape == (2, 1) + want and pos.data.dtype == np.float32
    np.testing.assert_allclose(pos.scale, (1, 1) + tuple(voxel))
    res = CliRunner().invoke(cli, args[:-1] + ["--cluster", "slurm"])
    assert res.exit_code != 0 and "in-process" in res.output


@pytest.mark.gpu
def test_cli_deskew_blosc_input_to_ngff05_output(gpu, tmp_path):
    """An iohub-style input (NGFF 0.4, Blosc zstd bit-shuffled chunks) deskewed into an NGFF 0.5 (zarr v3) store with
    the same codec, chosen by `output_ome_zarr_version` (reference settings.py:43, utils/ngff.py:27-39)."""
    src = tmp_path / "in.zarr"
    shape = (1, 1, 16, 24, 20)
    data = make_plate(src, positions=(("A", "1", "0"),), shape=shape, compressor="blosc")
    assert json.loads((src / "A/1/0/0/.zarray").read_text())["compressor"]["id"] == "blosc"
    cfg = tmp_path / "deskew.yml"
    cfg.write_text(DESKEW_YML + "output_ome_zarr_version: '0.5'\n")
    out = tmp_path / "deskewed.zarr"
    res = CliRunner().invoke(cli, expand_eat_all(["deskew", "-i", str(src / "A/1/0"), "-c", str(cfg), "-o", str(out), "--cluster", "debug"]))
    assert res.exit_code == 0, res.output
    got = io.open_ome_zarr(out / "A/1/0")
    assert got.version == "0.5" and got.data.zarr_format == 3 and got.data.codecs[0].kind == "blosc"
    meta = json.loads((out / "A/1/0/0/zarr.json").read_text())
    assert meta["codecs"][1]["configuration"] == {"cname": "zstd", "clevel": 1, "shuffle": "bitshuffle", "typesize": 4, "blocksize": 0}
    want = O.fast_deskew_zyx(data[("A", "1", "0", 0, 0)].astype(np.float32), 36.17, 0.371, True, 3, "mean")
    assert np.abs(got.data[0, 0] - want).max() <= 1e-5 * want.max()
    assert "biahub-deskew" in got.zattrs["extra_metadata"]


@pytest.mark.gpu
def test_cli_deskew_device_resident_blosc_lz4_store(gpu, tmp_path, monkeypatch):
    """`BH_ZARR_COMPRESSOR=blosc-lz4`: the deskew result stays in HBM until the output store has bit-shuffled AND lz4-compressed
    it on the GPU (csrc/lz4.hip); the chunks are ordinary Blosc frames — the host reader (pyarrow's lz4, as numcodecs' c-blosc
    would) decodes them to the oracle's volume.  Volumes large enough for several 256-KiB blocks per chunk."""
    src = tmp_path / "in.zarr"
    shape = (1, 2, 48, 64, 128)
    data = make_plate(src, positions=(("A", "1", "0"),), shape=shape, compressor="blosc")
    cfg = tmp_path / "deskew.yml"
    cfg.write_text(DESKEW_YML)
    out = tmp_path / "deskewed.zarr"
    monkeypatch.setenv("BH_ZARR_COMPRESSOR", "blosc-lz4")
    res = CliRunner().invoke(cli, expand_eat_all(["deskew", "-i", str(src / "A/1/0"), "-c", str(cfg), "-o", str(out), "--cluster", "debug"]))
    assert res.exit_code == 0, res.output
    meta = json.loads((out / "A/1/0/0/.zarray").read_text())
    assert meta["compressor"]["id"] == "blosc" and meta["compressor"]["cname"] == "lz4"
    got = io.open_ome_zarr(out / "A/1/0")
    raw_bytes = int(np.prod(got.data.shape)) * 4
    stored = sum(f.stat().st_size for f in (out / "A/1/0/0").rglob("*") if f.is_file())
    assert stored < 0.8 * raw_bytes  # the overhang fill alone is half the volume: it really is compressed
    monkeypatch.setenv("BH_LZ4_DEVICE", "0")  # read back through the host codec only
    for c in (0, 1):
        want = O.fast_deskew_zyx(data[("A", "1", "0", 0, c)].astype(np.float32), 36.17, 0.371, True, 3, "mean")
        assert np.abs(got.data[0, c] - want).max() <= 1e-5 * want.max()


@pytest.mark.gpu
@pytest.mark.parametrize("stores", ["blosc-lz4", "raw"])
def test_cli_rl_deconvolve_device_resident(gpu, tmp_path, monkeypatch, stores):
    """`rl-deconvolve` with both ends in HBM (deconvolve.richardson_lucy_czyx_device): the volumes arrive through the store's
    device read path and the estimate leaves through its device codec (Blosc-lz4 output) or a plain download (uncompressed
    stores: no staged read, numpy in); every (t, c) unit against the oracle; BH_PIPE_DEVICE_INPUT=0 gives the same store."""
    src = tmp_path / "in.zarr"
    shape = (1, 2, 24, 32, 64)
    data = make_plate(src, positions=(("A", "1", "0"),), shape=shape, compressor="blosc" if stores != "raw" else None)
    psf = np.zeros((5, 5, 5), np.float32)
    zz, yy, xx = np.ogrid[-2:3, -2:3, -2:3]
    psf[:] = np.exp(-0.5 * (zz ** 2 / 1.2 ** 2 + yy ** 2 / 0.9 ** 2 + xx ** 2 / 0.9 ** 2))
    psf /= psf.sum()
    io.create_empty_plate(tmp_path / "psf.zarr", [("0", "0", "0")], ["PSF"], (1, 1) + psf.shape, chunks=(1, 1) + psf.shape, dtype=np.float32)
    io.open_ome_zarr(tmp_path / "psf.zarr" / "0/0/0").data[0, 0] = psf
    cfg = tmp_path / "rl.yml"
    cfg.write_text("iterations: 4\neps: 1.0e-6\n")
    if stores != "raw":
        monkeypatch.setenv("BH_ZARR_COMPRESSOR", stores)
    outs = []
    for k, dev_in in enumerate(("1", "0")):
        monkeypatch.setenv("BH_PIPE_DEVICE_INPUT", dev_in)
        out = tmp_path / f"rl{k}.zarr"
        res = CliRunner().invoke(cli, expand_eat_all(["rl-deconvolve", "-i", str(src / "A/1/0"), "-c", str(cfg), "-p", str(tmp_path / "psf.zarr"),
                                                      "-o", str(out)]))
        assert res.exit_code == 0, res.output
        outs.append(io.open_ome_zarr(out / "A/1/0"))
    for c in (0, 1):
        want = O.richardson_lucy_zyx(data[("A", "1", "0", 0, c)].astype(np.float32), psf, 4, 1e-6)
        got = outs[0].data[0, c]
        assert got.dtype == np.float32 and np.abs(got - want).max() <= 1e-4 * np.abs(want).max()
        assert np.array_equal(got, outs[1].data[0, c])


@pytest.mark.gpu
def test_cli_steps_end_to_end(gpu, tmp_path):
    src = tmp_path / "in.zarr"
    shape = (2, 2, 16, 24, 20)
    data = make_plate(src, shape=shape)
    r = CliRunner()
    # deskew --cluster debug
    cfg = tmp_path / "deskew.yml"
    cfg.write_text(DESKEW_YML)
    out = tmp_path / "deskewed.zarr"
    res = r.invoke(cli, expand_eat_all(["deskew", "-i", str(src / "A/1/0"), str(src / "B/2/0"), "-c", str(cfg), "-o",
                                        str(out), "--cluster", "debug"]))
    assert res.exit_code == 0, res.output
    assert res.output.count("Deskew complete:") == 2
    assert (tmp_path / "slurm_output" / "submitit_jobs_ids.log").exists()
    got = io.open_ome_zarr(out / "B/2/0")
    # _fast_deskew_czyx takes channel 0 of each single-channel unit; every (t, c) unit is deskewed independently
    want = O.fast_deskew_zyx(data[("B", "2", "0", 1, 1)].astype(np.float32), 36.17, 0.371, True, 3, "mean")
    assert np.abs(got.data[1, 1] - want).max() <= 1e-5 * want.max()
    assert "biahub-deskew" in got.zattrs["extra_metadata"]
    # deconvolve --local with a psf store
    psf_store = tmp_path / "psf.zarr"
    psf = O.gaussian_psf((5, 5, 5), (1.0, 1.0, 1.0))
    io.create_empty_plate(psf_store, [("0", "0", "0")], ["PSF"], (1, 1, 5, 5, 5), scale=(1, 1, 1, 0.116, 0.116))
    io.open_ome_zarr(psf_store / "0/0/0").data[0, 0] = psf
    (tmp_path / "decon.yml").write_text("regularization_strength: 0.001\n")
    dec = tmp_path / "sub" / "decon.zarr"
    dec.parent.mkdir()
    res = r.invoke(cli, ["deconvolve", "-i", str(src / "A/1/0"), "-p", str(psf_store), "-c", str(tmp_path / "decon.yml"),
                         "-o", str(dec), "--local"])
    assert res.exit_code == 0, res.output
    vol = data[("A", "1", "0", 0, 1)].astype(np.float32)
    want = O.tikhonov_zyx(vol, O.compute_transfer_function(psf, vol.shape), 1e-3)
    assert np.abs(io.open_ome_zarr(dec / "A/1/0").data[0, 1] - want).max() <= 1e-4 * np.abs(want).max()
    assert io.open_ome_zarr(dec.parent / "transfer_function.zarr").data.shape == (1, 1, 16, 24, 20)
    # stabilize
    mats = [np.eye(4).tolist(), [[1, 0, 0, 0.5], [0, 1, 0, -1.25], [0, 0, 1, 2.0], [0, 0, 0, 1]]]
    (tmp_path / "stab.yml").write_text(json.dumps({
        "stabilization_estimation_channel": "ch0", "stabilization_type": "xyz", "stabilization_channels": ["ch1"],
        "affine_transform_zyx_list": mats, "time_indices": "all"}))
    stab = tmp_path / "stab.zarr"
    res = r.invoke(cli, ["stabilize", "-i", str(src / "A/1/0"), "-c", str(tmp_path / "stab.yml"), "-o", str(stab), "--local"])
    assert res.exit_code == 0, res.output
    sp = io.open_ome_zarr(stab / "A/1/0")
    for c in (0, 1):  # every channel is stabilized, listed in stabilization_channels or not (stabilize.py:150-151)
        want = O.apply_affine_transform(data[("A", "1", "0", 1, c)], np.array(mats[1]), shape[-3:])
        assert np.abs(sp.data[1, c] - want).max() <= 1e-5 * want.max()
    assert sp.data.dtype == np.float32 and list(sp.scale) == [1.0] * 5  # output_voxel_size is the output scale
    # a time subset: the output holds len(time_indices) frames and takes output_voxel_size (stabilize.py:183-219)
    (tmp_path / "stab1.yml").write_text(json.dumps({
        "stabilization_estimation_channel": "ch0", "stabilization_type": "xyz", "stabilization_channels": ["ch1"],
        "affine_transform_zyx_list": mats, "time_indices": [1], "output_voxel_size": [1, 1, 0.5, 0.25, 0.25]}))
    res = r.invoke(cli, ["stabilize", "-i", str(src / "A/1/0"), "-c", str(tmp_path / "stab1.yml"), "-o",
                         str(tmp_path / "stab1.zarr"), "--local"])
    assert res.exit_code == 0, res.output
    sp1 = io.open_ome_zarr(tmp_path / "stab1.zarr" / "A/1/0")
    assert sp1.data.shape == (1, 2) + shape[-3:] and list(sp1.scale) == [1, 1, 0.5, 0.25, 0.25]
    assert np.array_equal(sp1.data[0, 0], sp.data[1, 0]) and np.array_equal(sp1.data[0, 1], sp.data[1, 1])
    # one settings file per FOV: each position takes the file whose name contains row_col_fov (stabilize.py:262-267)
    (tmp_path / "A_1_0.yml").write_text((tmp_path / "stab.yml").read_text())
    (tmp_path / "B_2_0.yml").write_text(json.dumps({
        "stabilization_estimation_channel": "ch0", "stabilization_type": "xyz", "stabilization_channels": ["ch1"],
        "affine_transform_zyx_list": [np.eye(4).tolist(), np.eye(4).tolist()], "time_indices": "all"}))
    res = r.invoke(cli, expand_eat_all(["stabilize", "-i", str(src / "A/1/0"), str(src / "B/2/0"), "-c", str(tmp_path / "A_1_0.yml"),
                         "-c", str(tmp_path / "B_2_0.yml"), "-o", str(tmp_path / "stab2.zarr"), "--local"]))
    assert res.exit_code == 0, res.output
    assert np.array_equal(io.open_ome_zarr(tmp_path / "stab2.zarr" / "A/1/0").data[1, 1], sp.data[1, 1])
    assert np.array_equal(io.open_ome_zarr(tmp_path / "stab2.zarr" / "B/2/0").data[1, 1],
                          data[("B", "2", "0", 1, 1)].astype(np.float32))  # identity transforms for that FOV
    # register
    (tmp_path / "reg.yml").write_text(json.dumps({
        "source_channel_names": ["ch0"], "target_channel_name": "ch1", "affine_transform_zyx": mats[1],
        "keep_overhang": True}))
    reg = tmp_path / "reg.zarr"
    res = r.invoke(cli, ["register", "-s", str(src / "A/1/0"), "-t", str(src / "B/2/0"), "-c", str(tmp_path / "reg.yml"),
                         "-o", str(reg), "--local"])
    assert res.exit_code == 0, res.output
    rp = io.open_ome_zarr(reg / "A/1/0")
    assert rp.channel_names == ["ch0", "ch1"]
    want = O.apply_affine_transform(data[("A", "1", "0", 0, 0)], np.array(mats[1]), shape[-3:])
    assert np.abs(rp.data[0, 0] - want).max() <= 1e-5 * want.max()
    assert np.array_equal(rp.data[0, 1], data[("B", "2", "0", 0, 1)].astype(np.float32))  # target channel copied
    # flip in place, bit-exact
    res = r.invoke(cli, ["flip", "-i", str(src / "A/1/0"), "-x", "-y"])
    assert res.exit_code == 0, res.output
    assert np.array_equal(io.open_ome_zarr(src / "A/1/0").data[1, 0], data[("A", "1", "0", 1, 0)][:, ::-1, ::-1])


def test_estimate_registration_settings(tmp_path):
    from biahub_amd.settings import AffineTransformSettings, EstimateRegistrationSettings

    s = EstimateRegistrationSettings(target_channel_name="Phase3D", source_channel_name="GFP", estimation_method="ants")
    assert s.ants_registration_settings.sobel_filter is False and s.affine_transform_settings.transform_type == "euclidean"
    assert np.array_equal(s.affine_transform_settings.approx_transform, np.eye(4))
    with pytest.raises(ValueError, match="4x4"):
        AffineTransformSettings(approx_transform=[[1, 0], [0, 1]])
    with pytest.raises(ValueError):
        EstimateRegistrationSettings(target_channel_name="a", source_channel_name="b", estimation_method="icp")
    # methods that need napari / bead matching are refused by the CLI with a pointer to the reference package
    cfg = tmp_path / "e.yml"
    cfg.write_text("target_channel_name: ch0\nsource_channel_name: ch0\nestimation_method: manual\n")
    src = tmp_path / "in.zarr"
    make_plate(src, positions=(("A", "1", "0"),), shape=(1, 1, 4, 8, 8))
    res = CliRunner().invoke(cli, ["estimate-registration", "-s", str(src / "A/1/0"), "-t", str(src / "A/1/0"), "-o",
                                   str(tmp_path / "out.yml"), "-c", str(cfg)])
    assert res.exit_code != 0 and "not available" in res.output


@pytest.mark.gpu
def test_cli_estimate_registration_then_register(gpu, tmp_path):
    """BASELINE config 3 in miniature: estimate-registration writes register settings that undo a known warp."""
    import yaml
    from biahub_amd.register import apply_affine_transform

    shape = (40, 128, 128)
    arm_a = O.synthetic_volume(shape, seed=33, n_blobs=300)
    th = np.deg2rad(2.0)
    M = np.array([[1.02, 0, 0, 0.6], [0, 1.02 * np.cos(th), -1.02 * np.sin(th), -2.25],
                  [0, 1.02 * np.sin(th), 1.02 * np.cos(th), 4.75], [0, 0, 0, 1.0]])
    arm_b = apply_affine_transform(arm_a, M, shape)
    arm_b = np.where(arm_b == 0, 110.0, arm_b).astype(np.float32)
    for name, vol, ch in (("a.zarr", arm_a, "GFP"), ("b.zarr", arm_b, "Phase3D")):
        io.create_empty_plate(tmp_path / name, [("A", "1", "0")], [ch], (1, 1) + shape, scale=(1, 1, 0.2, 0.1, 0.1))
        io.open_ome_zarr(tmp_path / name / "A/1/0").data[0, 0] = vol
    cfg = tmp_path / "est.yml"
    cfg.write_text("target_channel_name: Phase3D\nsource_channel_name: GFP\nestimation_method: ants\n"
                   "affine_transform_settings:\n  transform_type: similarity\n")
    out_yml = tmp_path / "est" / "registration.yml"
    r = CliRunner()
    res = r.invoke(cli, ["estimate-registration", "-s", str(tmp_path / "a.zarr/A/1/0"), "-t", str(tmp_path / "b.zarr/A/1/0"),
                         "-o", str(out_yml), "-c", str(cfg), "--local"])
    assert res.exit_code == 0, res.output
    est = yaml.safe_load(out_yml.read_text())
    assert est["source_channel_names"] == ["GFP"] and est["target_channel_name"] == "Phase3D"
    T = np.array(est["affine_transform_zyx"])
    assert np.abs(T[:3, :3] - M[:3, :3]).max() < 2e-3
    centre = np.append((np.array(shape) - 1) / 2, 1)
    assert np.linalg.norm((T @ centre - M @ centre)[:3]) < 0.1
    assert np.allclose(np.load(out_yml.parent / "xyz_transforms" / "0.npy"), T)
    # optimize-registration: a rough manual transform in RegistrationSettings form is refined on the overlap (crop=True)
    th0 = np.deg2rad(1.5)
    rough = [[1, 0, 0, 0.0], [0, float(np.cos(th0)), float(-np.sin(th0)), -1.5], [0, float(np.sin(th0)), float(np.cos(th0)), 4.0],
             [0, 0, 0, 1]]
    rough_yml = tmp_path / "rough.yml"
    rough_yml.write_text(yaml.safe_dump({"source_channel_names": ["GFP"], "target_channel_name": "Phase3D",
                                         "affine_transform_zyx": rough, "time_indices": 0}))
    opt_yml = tmp_path / "opt" / "optimized.yml"
    res = r.invoke(cli, ["optimize-registration", "-s", str(tmp_path / "a.zarr/A/1/0"), "-t", str(tmp_path / "b.zarr/A/1/0"),
                         "-c", str(rough_yml), "-o", str(opt_yml)])
    assert res.exit_code == 0, res.output
    To = np.array(yaml.safe_load(opt_yml.read_text())["affine_transform_zyx"])
    assert np.abs(To[:3, :3] - M[:3, :3]).max() < 3e-3 and np.linalg.norm((To @ centre - M @ centre)[:3]) < 0.15
    # the written settings feed `register` unchanged
    est["keep_overhang"] = True
    out_yml.write_text(yaml.safe_dump(est))
    reg = tmp_path / "reg.zarr"
    res = r.invoke(cli, ["register", "-s", str(tmp_path / "a.zarr/A/1/0"), "-t", str(tmp_path / "b.zarr/A/1/0"), "-c",
                         str(out_yml), "-o", str(reg), "--local"])
    assert res.exit_code == 0, res.output
    got = io.open_ome_zarr(reg / "A/1/0").data[0, 0]
    core = (slice(8, 32), slice(16, 112), slice(16, 112))
    assert np.abs(got[core] - arm_b[core]).mean() < 0.02 * arm_b[core].mean()


@pytest.mark.gpu
def test_cli_flat_field(gpu, tmp_path):
    """``flat-field --cluster debug``: listed channels corrected, the rest copied as float32 (flat_field.py:144-155)."""
    src = tmp_path / "in.zarr"
    shape = (2, 2, 16, 12, 20)
    data = make_plate(src, positions=(("A", "1", "0"),), shape=shape)
    cfg = tmp_path / "ff.yml"
    cfg.write_text("channel_names: [ch1]\n")
    out = tmp_path / "ff.zarr"
    res = CliRunner().invoke(cli, ["flat-field", "-i", str(src / "A/1/0"), "-c", str(cfg), "-o", str(out), "--cluster", "debug"])
    assert res.exit_code == 0, res.output
    assert "Flat field channels: ['ch1']" in res.output and "RESOURCES:" in res.output
    got = io.open_ome_zarr(out / "A/1/0")
    assert got.data.dtype == np.float32 and got.data.shape == shape
    want = O.flat_field_zyx(data[("A", "1", "0", 1, 1)]).astype(np.float32)
    assert np.abs(got.data[1, 1] - want).max() <= 2e-7 * want.max()
    assert np.array_equal(got.data[1, 0], data[("A", "1", "0", 1, 0)].astype(np.float32))
    assert "biahub-flat_field" in got.zattrs["extra_metadata"]
    cfg.write_text("channel_names: [nope]\n")
    res = CliRunner().invoke(cli, ["flat-field", "-i", str(src / "A/1/0"), "-c", str(cfg), "-o", str(tmp_path / "x.zarr")])
    assert res.exit_code != 0 and "not found" in res.output


@pytest.mark.gpu
def test_cli_estimate_psf_feeds_deconvolve(gpu, tmp_path):
    """``estimate-psf`` writes the psf store that ``deconvolve -p`` reads (estimate_psf.py:114-121, deconvolve.py:131-139)."""
    rng = np.random.default_rng(3)
    shape = (40, 200, 200)  # the CLI's fixed detection settings want beads >= 50 voxels apart (estimate_psf.py:58-67)
    vol = rng.normal(110.0, 3.0, shape).astype(np.float32)
    zz, yy, xx = np.ogrid[: shape[0], : shape[1], : shape[2]]
    for c in ((12, 40, 40), (20, 140, 50), (28, 50, 150), (14, 150, 150)):
        vol += (2000 * np.exp(-0.5 * (((zz - c[0]) / 1.5) ** 2 + ((yy - c[1]) / 1.2) ** 2 + ((xx - c[2]) / 1.2) ** 2))).astype(np.float32)
    store = tmp_path / "beads.zarr"
    io.create_empty_plate(store, [("A", "1", "0")], ["beads"], (1, 1) + shape, scale=(1, 1, 0.25, 0.1, 0.1))
    io.open_ome_zarr(store / "A/1/0").data[0, 0] = vol
    cfg = tmp_path / "psf.yml"
    cfg.write_text("axis0_patch_size: 11\naxis1_patch_size: 15\naxis2_patch_size: 15\n")
    out = tmp_path / "psf.zarr"
    res = CliRunner().invoke(cli, ["estimate-psf", "-i", str(store / "A/1/0"), "-c", str(cfg), "-o", str(out)])
    assert res.exit_code == 0, res.output
    assert "Total beads: 4" in res.output
    pos = io.open_ome_zarr(out / "0/0/0")
    psf = pos.data[0, 0]
    assert pos.channel_names == ["PSF"] and psf.shape == (11, 15, 15) and psf.dtype == np.float32
    np.testing.assert_allclose(pos.scale, (1, 1, 0.25, 0.1, 0.1))
    assert psf.max() == 1.0 and psf.min() == 0.0 and np.unravel_index(int(psf.argmax()), psf.shape) == (5, 7, 7)
    half = psf[5, 7, :]   # sigma 1.2 voxels along x: exp(-0.5 (1/1.2)^2) = 0.707 one voxel off the peak
    assert abs(half[8] - np.exp(-0.5 / 1.44)) < 0.03


@pytest.mark.gpu
def test_cli_estimate_stabilization_then_stabilize(gpu, tmp_path):
    """``estimate-stabilization`` (xyz, phase-cross-corr) writes per-FOV settings that ``stabilize`` consumes."""
    import yaml

    rng = np.random.default_rng(5)
    base = (rng.random((16, 32, 40)) * 1000 + 100).astype(np.float32)
    rolls = [(0, 0, 0), (1, -2, 3), (2, 1, -1)]
    src = tmp_path / "in.zarr"
    io.create_empty_plate(src, [("A", "1", "0")], ["ch0"], (3, 1) + base.shape, scale=(1, 1, 0.2, 0.1, 0.1))
    pos = io.open_ome_zarr(src / "A/1/0")
    for t, r in enumerate(rolls):
        pos.data[t, 0] = np.roll(base, r, axis=(0, 1, 2))
    cfg = tmp_path / "est.yml"
    cfg.write_text("stabilization_estimation_channel: ch0\nstabilization_channels: [ch0]\nstabilization_type: xyz\n"
                   "stabilization_method: phase-cross-corr\nphase_cross_corr_settings:\n  normalization: magnitude\n"
                   "verbose: true\n")
    out = tmp_path / "stab_est"
    r = CliRunner()
    res = r.invoke(cli, ["estimate-stabilization", "-i", str(src / "A/1/0"), "-o", str(out), "-c", str(cfg), "--local"])
    assert res.exit_code == 0, res.output
    yml = out / "xyz_stabilization_settings" / "A_1_0.yml"
    est = yaml.safe_load(yml.read_text())
    mats = np.array(est["affine_transform_zyx_list"])
    assert mats.shape == (3, 4, 4) and np.array_equal(mats[0], np.eye(4))
    # get_tform_from_pcc correlates frame t against frame 0 and files the shift as (dx, dy, dz) (estimate_stabilization.py:288-305)
    stack = np.stack([np.roll(base, r_, axis=(0, 1, 2)) for r_ in rolls])
    for t in (1, 2):
        want, _, _ = O.get_tform_from_pcc(t, stack, np.broadcast_to(stack[0], stack.shape), "custom", "magnitude")
        assert np.array_equal(mats[t], want) and np.array_equal(mats[t][:3, 3], np.array(rolls[t], float)[::-1])
    assert (out / "shifts_per_position" / "A_1_0.csv").exists() and not (out / "transforms_per_position").exists()
    assert est["stabilization_method"] == "phase-cross-corr" and est["output_voxel_size"] == [1, 1, 0.2, 0.1, 0.1]
    # t_reference "previous": every timepoint against its predecessor (the prepared handle rolls its stored spectrum forward)
    cfg.write_text("stabilization_estimation_channel: ch0\nstabilization_channels: [ch0]\nstabilization_type: xyz\n"
                   "stabilization_method: phase-cross-corr\nphase_cross_corr_settings:\n  t_reference: previous\n")
    out_prev = tmp_path / "stab_prev"
    res = r.invoke(cli, ["estimate-stabilization", "-i", str(src / "A/1/0"), "-o", str(out_prev), "-c", str(cfg), "--local"])
    assert res.exit_code == 0, res.output
    prev = np.array(yaml.safe_load((out_prev / "xyz_stabilization_settings" / "A_1_0.yml").read_text())["affine_transform_zyx_list"])
    before = np.stack([stack[0], stack[0], stack[1]])
    for t in (1, 2):
        want, _, _ = O.get_tform_from_pcc(t, stack, before, "custom", None)
        assert np.array_equal(prev[t], want), (t, prev[t], want)
    assert np.array_equal(prev[2][:3, 3], (np.array(rolls[2], float) - np.array(rolls[1], float))[::-1])
    res = r.invoke(cli, ["stabilize", "-i", str(src / "A/1/0"), "-c", str(yml), "-o", str(tmp_path / "stab.zarr"), "--local"])
    assert res.exit_code == 0, res.output
    cfg.write_text("stabilization_estimation_channel: ch0\nstabilization_channels: [ch0]\nstabilization_type: z\n")
    res = r.invoke(cli, ["estimate-stabilization", "-i", str(src / "A/1/0"), "-o", str(out), "-c", str(cfg)])
    assert res.exit_code != 0 and "not available" in res.output


@pytest.mark.gpu
def test_cli_process_with_config_binning(gpu, tmp_path):
    """``process-with-config`` with the binning function: output shape and scale follow the factor (process_data.py:211-236)."""
    src = tmp_path / "in.zarr"
    shape = (2, 2, 8, 12, 16)
    data = make_plate(src, positions=(("A", "1", "0"),), shape=shape)
    cfg = tmp_path / "p.yml"
    cfg.write_text("processing_functions:\n- function: biahub.process_data.binning_czyx\n  input_channels: [ch0]\n"
                   "  kwargs:\n    binning_factor_zyx: [1, 2, 2]\n    mode: sum\n")
    out = tmp_path / "binned.zarr"
    res = CliRunner().invoke(cli, ["process-with-config", "-i", str(src / "A/1/0"), "-c", str(cfg), "-o", str(out), "--local"])
    assert res.exit_code == 0, res.output
    pos = io.open_ome_zarr(out / "A/1/0")
    assert pos.data.shape == (2, 2, 8, 6, 8) and pos.data.dtype == np.float32
    np.testing.assert_allclose(pos.scale, (1, 1, 1, 0.232, 0.232))
    czyx = np.stack([data[("A", "1", "0", 1, c)] for c in range(2)])
    assert np.array_equal(np.stack([pos.data[1, 0], pos.data[1, 1]]), O.binning_czyx(czyx, (1, 2, 2), "sum").astype(np.float32))
    cfg.write_text("processing_functions:\n- function: np.mean\n  input_channels: [ch0]\n")
    res = CliRunner().invoke(cli, ["process-with-config", "-i", str(src / "A/1/0"), "-c", str(cfg), "-o", str(out), "--local"])
    assert res.exit_code != 0


@pytest.mark.gpu
def test_cli_two_ranks_under_torchrun(gpu, tmp_path):
    """The N>1 product path: ``torchrun --nproc-per-node 2 -m biahub_amd deskew`` shards positions over two ranks, each
    bound to GPU ``LOCAL_RANK % device_count`` by ``parallel.init`` (reference fan-out: biahub/deskew.py:715-749).  On
    the 1-GPU test box both ranks wrap onto cuda:0, so the status exchange runs over gloo (RCCL refuses two ranks on one
    device); on an 8-GPU node the same command lands rank r on GPU r over RCCL."""
    import os
    import subprocess
    import sys

    import torch

    src = tmp_path / "in.zarr"
    keys = [(row, str(col), "0") for row in "AB" for col in (1, 2)]
    data = make_plate(src, positions=keys, shape=(1, 2, 16, 24, 20))
    cfg = tmp_path / "deskew.yml"
    cfg.write_text(DESKEW_YML)
    out = tmp_path / "deskewed.zarr"
    env = dict(os.environ, BH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=str(Path(__file__).resolve().parent.parent))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29631", "-m", "biahub_amd", "deskew", "-i", *[str(src.joinpath(*k)) for k in keys],
           "-c", str(cfg), "-o", str(out), "--cluster", "debug"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    n = torch.cuda.device_count()
    for rank in (0, 1):  # each rank reports the GPU it was bound to and its share of the plate
        assert f"[rank {rank}/2] deskew: device cuda:{rank % n}, 2 of 4 position(s)" in res.stderr, res.stderr
    assert res.stdout.count("Deskew complete:") == 4
    for k in keys:
        got = io.open_ome_zarr(out.joinpath(*k))
        for c in (0, 1):
            want = O.fast_deskew_zyx(data[k + (0, c)].astype(np.float32), 36.17, 0.371, True, 3, "mean")
            assert np.abs(got.data[0, c] - want).max() <= 1e-5 * want.max()
    # a position that cannot be processed: its traceback is logged, the rank that saw it exits non-zero
    import shutil

    shutil.rmtree(src / "B" / "2" / "0" / "0" / "0")  # drop the chunk files of one position's t=0
    (src / "B" / "2" / "0" / "0" / ".zarray").write_text("{ not json")
    res = subprocess.run(cmd[:-6] + ["-c", str(cfg), "-o", str(tmp_path / "again.zarr"), "--cluster", "debug"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode != 0


RECON_YML = {
    "input_channel_names": ["BF"], "time_indices": "all", "reconstruction_dimension": 3,
    "phase": {"transfer_function": {"wavelength_illumination": 0.450, "yx_pixel_size": 0.1, "z_pixel_size": 0.25, "z_padding": 0,
                                    "index_of_refraction_media": 1.3, "numerical_aperture_detection": 1.2,
                                    "numerical_aperture_illumination": 0.5, "invert_phase_contrast": False},
              "apply_inverse": {"reconstruction_algorithm": "Tikhonov", "regularization_strength": 1e-3}}}


def test_reconstruction_settings_surface():
    """The reference's reconstruct configurations load (tests/test_cli/test_reconstruct_cli.py:13-37,
    nextflow/configs/a549/reconstruct.yml); what this package does not reconstruct is refused by name."""
    from biahub_amd.compute_transfer_function import _refuse_unsupported
    from biahub_amd.settings import ReconstructionSettings

    s = ReconstructionSettings(**RECON_YML)
    assert s.output_channel_names == ["Phase3D"] and s.phase.apply_inverse.TV_iterations == 1
    a549 = {"input_channel_names": ["BF - Oblique"], "time_indices": "all", "reconstruction_dimension": 3,
            "phase": {"transfer_function": {"wavelength_illumination": 0.532, "z_padding": 5, "index_of_refraction_media": 1.4,
                                            "numerical_aperture_detection": 1.35, "numerical_aperture_illumination": 0.52,
                                            "invert_phase_contrast": False},
                      "apply_inverse": {"reconstruction_algorithm": "Tikhonov", "regularization_strength": 0.01,
                                        "TV_rho_strength": 0.001, "TV_iterations": 1}}}
    s = ReconstructionSettings(**a549)
    assert s.phase.transfer_function.yx_pixel_size is None and s.phase.transfer_function.z_padding == 5
    _refuse_unsupported(s)
    f = ReconstructionSettings(input_channel_names=["GFP"], fluorescence={})
    assert f.output_channel_names == ["GFP_Density3D"] and f.fluorescence.transfer_function.wavelength_emission == 0.507
    with pytest.raises(ValueError, match="cannot be combined"):
        ReconstructionSettings(input_channel_names=["GFP"], fluorescence={}, phase={})
    with pytest.raises(ValueError, match="specify one of"):
        ReconstructionSettings(input_channel_names=["GFP"])
    with pytest.raises(ValueError):
        ReconstructionSettings(input_channel_names=["GFP"], phase={"transfer_function": {"unknown_key": 1}})
    with pytest.raises(NotImplementedError, match="birefringence"):
        _refuse_unsupported(ReconstructionSettings(birefringence={"transfer_function": {"swing": 0.1}}))
    with pytest.raises(NotImplementedError, match="3-D"):
        _refuse_unsupported(ReconstructionSettings(input_channel_names=["BF"], phase={}, reconstruction_dimension=2))
    with pytest.raises(NotImplementedError, match="Tikhonov"):
        _refuse_unsupported(ReconstructionSettings(input_channel_names=["BF"], phase={"apply_inverse": {"reconstruction_algorithm": "TV"}}))


def test_apply_inv_tf_cli_init_only(tmp_path):
    """``apply-inv-tf --init`` lays the output plate out and prints the resource line without a GPU
    (reference: tests/test_cli/test_reconstruct_cli.py:69-91)."""
    import yaml

    src = tmp_path / "input.zarr"
    make_plate(src, positions=(("A", "1", "0"), ("B", "1", "0")), shape=(1, 1, 5, 8, 8), dtype=np.float32,
               scale=(1, 1, 0.25, 0.1, 0.1))
    for key in (("A", "1", "0"), ("B", "1", "0")):  # make_plate names channels ch0..: the config asks for "BF"
        pos = io.open_ome_zarr(src.joinpath(*key))
        pos.zattrs["omero"]["channels"][0]["label"] = "BF"
        pos.update_zattrs({"biahub-deskew": {"angle": 30.0, "fov": key[0]}, "waveorder": {"v": 1}, "unrelated": 7})
    cfg = tmp_path / "reconstruct.yml"
    cfg.write_text(yaml.dump(RECON_YML))
    out = tmp_path / "output.zarr"
    res = CliRunner().invoke(cli, expand_eat_all(["apply-inv-tf", "--init", "-i", str(src / "A/1/0"), str(src / "B/1/0"), "-c", str(cfg),
                                                  "-o", str(out)]))
    assert res.exit_code == 0, res.output
    assert "RESOURCES:" in res.output and "2 positions, 1 output channels" in res.output
    pos = io.open_ome_zarr(out / "B/1/0")
    assert pos.channel_names == ["Phase3D"] and pos.data.shape == (1, 1, 5, 8, 8) and pos.data.dtype == np.float32
    assert pos.scale == [1.0, 1.0, 0.25, 0.1, 0.1]
    # per-position provenance follows the position (metadata_sources / PROVENANCE_METADATA_KEYS:
    # biahub/apply_inverse_transfer_function.py:66-72); other attributes do not
    assert pos.zattrs["biahub-deskew"] == {"angle": 30.0, "fov": "B"} and pos.zattrs["waveorder"] == {"v": 1}
    assert "unrelated" not in pos.zattrs
    res = CliRunner().invoke(cli, ["apply-inv-tf", "-i", str(src / "A/1/0"), "-c", str(cfg), "-o", str(out)])
    assert res.exit_code != 0 and "--transfer-function-dirpath / -t is required unless using --init." in res.output


@pytest.mark.gpu
def test_cli_compute_tf_apply_inv_tf_reconstruct(gpu, tmp_path):
    """compute-tf -> apply-inv-tf --cluster debug, and reconstruct (= both), as the reference's CLI test drives them
    (tests/test_cli/test_reconstruct_cli.py:94-157), with the values held to the oracle's restatement of waveorder."""
    import yaml

    src = tmp_path / "input.zarr"
    shape = (2, 2, 8, 16, 20)
    data = make_plate(src, positions=(("A", "1", "0"), ("B", "1", "0")), shape=shape, dtype=np.float32, scale=(1, 1, 0.25, 0.1, 0.1))
    for key in (("A", "1", "0"), ("B", "1", "0")):
        pos = io.open_ome_zarr(src.joinpath(*key))
        pos.zattrs["omero"]["channels"][1]["label"] = "BF"
        pos.update_zattrs({})
    cfg_d = json.loads(json.dumps(RECON_YML))
    cfg_d["phase"]["transfer_function"]["z_padding"] = 2
    cfg = tmp_path / "reconstruct.yml"
    cfg.write_text(yaml.dump(cfg_d))
    r = CliRunner()
    tf = tmp_path / "tf.zarr"
    res = r.invoke(cli, ["compute-tf", "-i", str(src / "A/1/0"), "-c", str(cfg), "-o", str(tf)])
    assert res.exit_code == 0, res.output
    assert "Transfer function computed and saved to" in res.output
    store = io.open_ome_zarr(tf)
    assert store.array_keys() == ["imaginary_potential_transfer_function", "real_potential_transfer_function"]
    H = store["real_potential_transfer_function"][0, 0]
    assert H.shape == (12, 16, 20) and H.dtype == np.complex64
    assert store.zattrs["settings"]["phase"]["transfer_function"]["z_padding"] == 2
    wre, _ = O.wo_phase_transfer_function_3d((8, 16, 20), 0.1, 0.25, 0.45, 2, 1.3, 0.5, 1.2)
    assert np.abs(H - wre).max() <= 2e-5 * np.abs(wre).max()
    out = tmp_path / "output.zarr"
    res = r.invoke(cli, expand_eat_all(["apply-inv-tf", "--cluster", "debug", "-i", str(src / "A/1/0"), str(src / "B/1/0"), "-t", str(tf),
                                        "-c", str(cfg), "-o", str(out)]))
    assert res.exit_code == 0, res.output
    assert res.output.count("Apply-inv-tf complete:") == 2 and "RESOURCES:" in res.output
    got = io.open_ome_zarr(out / "B/1/0")
    assert got.channel_names == ["Phase3D"] and got.data.shape == (2, 1, 8, 16, 20)
    for t in (0, 1):
        want = O.wo_apply_inverse_transfer_function(data[("B", "1", "0", t, 1)], wre, 2, 1e-3, True)
        assert np.abs(got.data[t, 0] - want).max() <= 1e-4 * np.abs(want).max()
    # reconstruct = compute-tf + apply-inv-tf; the transfer function lands next to the output, named after the config
    out2 = tmp_path / "rec" / "output.zarr"
    out2.parent.mkdir()
    res = r.invoke(cli, ["reconstruct", "-i", str(src / "A/1/0"), "-c", str(cfg), "-o", str(out2)])
    assert res.exit_code == 0, res.output
    assert (out2.parent / "transfer_function_reconstruct.zarr").exists()
    assert np.array_equal(io.open_ome_zarr(out2 / "A/1/0").data[1, 0], io.open_ome_zarr(out / "A/1/0").data[1, 0])
    # a fluorescence configuration through the same commands
    fcfg = tmp_path / "fluor.yml"
    fcfg.write_text(yaml.dump({"input_channel_names": ["ch0"], "reconstruction_dimension": 3,
                               "fluorescence": {"transfer_function": {"yx_pixel_size": 0.1, "z_pixel_size": 0.25},
                                                "apply_inverse": {"regularization_strength": 1e-2}}}))
    out3 = tmp_path / "fl" / "output.zarr"
    out3.parent.mkdir()
    res = r.invoke(cli, ["reconstruct", "-i", str(src / "A/1/0"), "-c", str(fcfg), "-o", str(out3)])
    assert res.exit_code == 0, res.output
    otf = O.wo_fluorescence_transfer_function_3d((8, 16, 20), 0.1, 0.25, 0.507, 0, 1.3, 1.2)
    want = O.wo_apply_inverse_transfer_function(data[("A", "1", "0", 0, 0)], otf, 0, 1e-2, False)
    g3 = io.open_ome_zarr(out3 / "A/1/0")
    assert g3.channel_names == ["ch0_Density3D"] and np.abs(g3.data[0, 0] - want).max() <= 1e-4 * np.abs(want).max()


def test_config1_host_deskew_golden_and_cli_without_gpu(tmp_path, deskew_cases):
    """BASELINE config 1 as it reads — `deskew ... --cluster debug` with the reference's default `device: cpu` and NO GPU: the
    operator runs libbhcore's own host implementation (bh_host_deskew; never the test oracle).  The host path against every
    golden vector the reference produced (shapes, N, fills, dtypes, splits), then the CLI end to end on a (64, 256, 256)
    position with example_deskew_settings.yml's parameters, checked against the oracle."""
    import torch

    from biahub_amd.deskew import _fast_deskew_czyx

    z, meta = deskew_cases
    for m in meta:
        kw = dict(ls_angle_deg=m["angle"], px_to_scan_ratio=m["ratio"], keep_overhang=m["keep_overhang"],
                  average_n_slices=m["n"], overhang_fill=m["fill"])
        got = _fast_deskew_czyx(z[m["name"] + "__in"][None], device="cpu", num_splits=m["splits"] or 1, **kw)[0]
        ref = z[m["name"] + "__out"]
        ref = ref[0] if ref.ndim == 4 else ref
        assert got.shape == ref.shape and got.dtype == np.float32, m
        assert rel_err(got, ref) <= 1e-5, (m, rel_err(got, ref))
    if torch.cuda.is_available():
        pytest.skip("GPU present: the CLI would (rightly) take the GPU; the host path itself is covered above")
    src = tmp_path / "in.zarr"
    data = make_plate(src, positions=(("A", "1", "0"),), shape=(1, 1, 64, 256, 256), dtype=np.uint16)
    cfg = tmp_path / "deskew.yml"
    cfg.write_text(DESKEW_YML)  # no `device:` entry: DeskewSettings defaults to "cpu" like the reference
    out = tmp_path / "out.zarr"
    res = CliRunner().invoke(cli, expand_eat_all(["deskew", "-i", str(src / "A/1/0"), "-c", str(cfg), "-o", str(out),
                                                  "--cluster", "debug"]))
    assert res.exit_code == 0, res.output
    got = io.open_ome_zarr(out / "A/1/0").data[0, 0]
    want = O.fast_deskew_zyx(data[("A", "1", "0", 0, 0)].astype(np.float32), 36.17, 0.371, True, 3, "mean")
    assert got.shape == want.shape == (86, 256, 380)
    assert rel_err(got, want) <= 1e-5
