"""`biahub concatenate` host logic against vectors captured from the reference's own functions
(tests/golden/concatenate.json, made by `tests/golden/make_golden.py concatenate`), and the command end to end."""
import json
from pathlib import Path

import numpy as np
import pytest
import yaml
from click.testing import CliRunner
from pydantic import ValidationError

from biahub_amd import concatenate as cc
from biahub_amd import io
from biahub_amd.cli import cli
from biahub_amd.settings import ConcatenateSettings

G = json.loads((Path(__file__).parent / "golden" / "concatenate.json").read_text())


def test_slice_helpers_match_reference():
    for r in G["get_path_slice_param"]:
        assert cc.get_path_slice_param(r["param"], r["index"], r["total"]) == r["result"], r
    for r in G["get_slice"]:
        if "error" in r:
            with pytest.raises(ValueError) as e:
                cc.get_slice(r["param"], r["max"])
            assert str(e.value) == r["error"]
        else:
            s = cc.get_slice(r["param"], r["max"])
            assert [s.start, s.stop] == r["result"], r
    for r in G["create_path_slicing_params"]:
        got = cc.create_path_slicing_params(r["z"], r["y"], r["x"], tuple(r["shape"]))
        assert [[s.start, s.stop] for s in got] == r["result"]
    for r in G["calculate_cropped_size"]:
        assert list(cc.calculate_cropped_size([slice(*q) for q in r["slices"]])) == r["result"]
    for r in G["validate_slicing_params_zyx"]:
        params = [[slice(*q) for q in sl] for sl in r["slices"]]
        if r["error"] is None:
            cc.validate_slicing_params_zyx(params)
        else:
            with pytest.raises(ValueError) as e:
                cc.validate_slicing_params_zyx(params)
            assert str(e.value) == r["error"]


def test_concatenate_settings_match_reference():
    assert len(G["settings"]) >= 25
    for r in G["settings"]:
        if "dump" in r:
            assert ConcatenateSettings(**r["config"]).model_dump() == r["dump"], r["config"]
        else:
            with pytest.raises(ValidationError) as e:
                ConcatenateSettings(**r["config"])
            assert sorted({err["msg"] for err in e.value.errors()}) == r["errors"], r["config"]


def _plates(tmp):
    for name, meta in G["channel_combiner_plates"].items():
        io.create_empty_plate(tmp / f"{name}.zarr", [tuple(p.split("/")) for p in meta["positions"]], meta["channels"],
                              tuple(meta["shape"]), dtype=np.uint16)


def test_channel_combiner_matches_reference(tmp_path):
    _plates(tmp_path)
    assert len(G["channel_combiner"]) == 4
    for r in G["channel_combiner"]:
        args = ([str(tmp_path / g) for g in r["globs"]], r["channels"], r["slicing_zyx"])
        if "error" in r:
            with pytest.raises(ValueError) as e:
                cc.get_channel_combiner_metadata(*args)
            assert str(e.value) == r["error"]
            continue
        paths, names, cin, cout, sl = cc.get_channel_combiner_metadata(*args)
        assert [str(p.relative_to(tmp_path)) for p in paths] == r["paths"]  # natural order: A/2 before A/10
        assert (names, cin, cout) == (r["names"], r["input_idx"], r["output_idx"])
        assert [[[q.start, q.stop] for q in s] for s in sl] == r["slices"]


def test_concatenate_init_creates_sharded_v3_plate_and_resolve_mode(tmp_path):
    _plates(tmp_path)
    cfg = tmp_path / "c.yml"
    cfg.write_text("concat_data_paths:\n\ntime_indices: all\nchannel_names:\n- all\n- all\n")
    resolved = tmp_path / "resolved.yml"
    r = CliRunner().invoke(cli, ["concatenate", "-c", str(cfg), "-o", str(resolved), "--concat-data-paths", str(tmp_path / "p1.zarr/*/*/*"),
                                 "--concat-data-paths", str(tmp_path / "p2.zarr/*/*/*")])
    assert r.exit_code == 0, r.output
    out = yaml.safe_load(resolved.read_text())
    assert out["concat_data_paths"] == [str(tmp_path / "p1.zarr/*/*/*"), str(tmp_path / "p2.zarr/*/*/*")]
    assert out["output_ome_zarr_version"] == "0.5"
    out.update(Z_slice=[0, 2], chunks_czyx=[1, 1, 6, 8], shards_ratio=[2, 1, 2, 1, 1])
    resolved.write_text(yaml.safe_dump(out))
    store = tmp_path / "out.zarr"
    r = CliRunner().invoke(cli, ["concatenate", "-c", str(resolved), "-o", str(store), "--init"])
    assert r.exit_code == 0, r.output
    assert "RESOURCES:" in r.output and "Created" in r.output
    pos = io.open_ome_zarr(store / "A/10/0")
    assert pos.version == "0.5" and pos.channel_names == ["DAPI", "Cy5", "GFP", "RFP"]
    assert pos.data.shape == (2, 4, 2, 6, 8) and pos.data.sharded
    assert pos.data.inner == (1, 1, 1, 6, 8) and pos.data.chunks == (2, 1, 2, 6, 8)
    # incompatible sources: same crop request, different result sizes
    bad = dict(out, Z_slice=[[0, 2], [0, 3]])
    resolved.write_text(yaml.safe_dump(bad))
    r = CliRunner().invoke(cli, ["concatenate", "-c", str(resolved), "-o", str(tmp_path / "bad.zarr"), "--init"])
    assert r.exit_code != 0 and "Inconsistent slice sizes" in str(r.exception)


@pytest.mark.gpu
@pytest.mark.parametrize("version,shards", [("0.4", None), ("0.5", [2, 1, 1, 1, 1])])
def test_concatenate_cli_end_to_end(gpu, tmp_path, version, shards):
    """Two sources, chosen channels, per-source crops, a time subset; every output voxel equals the NumPy slice of its source
    (the crop is `bh_crop_flip`: bit-exact)."""
    rng = np.random.default_rng(4)
    shape = (3, 2, 6, 10, 12)
    data = {}
    for name, chans in (("a", ["DAPI", "Cy5"]), ("b", ["GFP", "DAPI"])):
        io.create_empty_plate(tmp_path / f"{name}.zarr", [("A", "1", "0"), ("B", "2", "0")], chans, shape, dtype=np.uint16,
                              scale=(1, 1, 0.5, 0.1, 0.1), compressor="blosc")
        for pos in ("A/1/0", "B/2/0"):
            arr = io.open_ome_zarr(tmp_path / f"{name}.zarr" / pos).data
            for t in range(shape[0]):
                for c in range(shape[1]):
                    data[name, pos, t, c] = rng.integers(1, 60000, shape[2:]).astype(np.uint16)
                    arr[t, c] = data[name, pos, t, c]
    cfg = tmp_path / "c.yml"
    cfg.write_text(yaml.safe_dump({
        "concat_data_paths": [str(tmp_path / "a.zarr/*/*/*"), str(tmp_path / "b.zarr/*/*/*")],
        "channel_names": [["Cy5"], "all"], "time_indices": [0, 2],
        "Z_slice": [[0, 4], [2, 6]], "Y_slice": [1, 9], "X_slice": "all",
        "output_ome_zarr_version": version, "shards_ratio": shards, "ensure_unique_positions": True}))
    out = tmp_path / "sub" / "out.zarr"
    out.parent.mkdir()
    r = CliRunner().invoke(cli, ["concatenate", "-c", str(cfg), "-o", str(out), "--cluster", "debug"])
    assert r.exit_code == 0, (r.output, r.exception)
    assert (out.parent / "slurm_output" / "submitit_jobs_ids.log").exists()
    # the second source's positions collide with the first's: ensure_unique_positions renames them <col>d1
    for src, key, crop_z, chan_map in (("a", "A/1/0", slice(0, 4), {0: 1}), ("b", "A/1d1/0", slice(2, 6), {1: 0, 2: 1})):
        pos = io.open_ome_zarr(out / key)
        assert pos.version == version and pos.channel_names == ["Cy5", "GFP", "DAPI"] and pos.scale[2:] == [0.5, 0.1, 0.1]
        assert pos.data.shape == (2, 3, 4, 8, 12) and pos.data.dtype == np.uint16
        assert "biahub-concatenate" in pos.zattrs["extra_metadata"]
        src_pos = key.replace("1d1", "1")
        for to, ti in enumerate((0, 2)):
            written = set()
            for c_out, c_in in chan_map.items():
                assert np.array_equal(pos.data[to, c_out], data[src, src_pos, ti, c_in][crop_z, 1:9, :]), (key, to, c_out)
                written.add(c_out)
            for c_out in set(range(3)) - written:
                assert not pos.data[to, c_out].any()
