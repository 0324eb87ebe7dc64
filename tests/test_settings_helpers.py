"""CPU: the kept YAML/settings/helper surface against values captured from the reference (tests/golden/helpers.json)."""

import os

import numpy as np
import pytest
import yaml

from biahub_amd import register, settings
from biahub_amd.utils import cluster, config, paths


@pytest.mark.parametrize("fname,model", [
    ("example_deskew_settings.yml", settings.DeskewSettings),
    ("example_registration_settings.yml", settings.RegistrationSettings),
    ("example_stabilize_timelapse_settings.yml", settings.StabilizationSettings),
])
def test_example_yaml_loads_like_reference(helpers_golden, tmp_path, fname, model):
    rec = helpers_golden[fname]
    f = tmp_path / fname
    f.write_text(rec["yaml"])
    m = config.yaml_to_model(f, model)
    assert m.model_dump(mode="json") == rec["dump"]
    assert config.settings_fingerprint(m) == rec["fingerprint"]  # same resume token as the reference
    out = tmp_path / "round.yml"
    config.model_to_yaml(m, out)
    assert model(**yaml.safe_load(out.read_text())).model_dump() == m.model_dump()


def test_deskew_settings_rules(helpers_golden):
    m = settings.DeskewSettings(pixel_size_um=0.116, ls_angle_deg=36.1749, scan_step_um=0.3125)
    assert m.model_dump(mode="json") == helpers_golden["deskew_derived_ratio"]
    with pytest.raises(ValueError):
        settings.DeskewSettings(pixel_size_um=0.116, ls_angle_deg=30)  # neither ratio nor scan step
    with pytest.raises(ValueError):
        settings.DeskewSettings(pixel_size_um=0.116, ls_angle_deg=60, px_to_scan_ratio=0.3)
    with pytest.raises(ValueError):
        settings.DeskewSettings(pixel_size_um=0.116, ls_angle_deg=30, px_to_scan_ratio=0.3, typo=1)  # extra="forbid"
    d = settings.DeconvolveSettings()
    assert d.model_dump(mode="json") == helpers_golden["deconvolve_default"]["dump"]
    assert config.settings_fingerprint(d) == helpers_golden["deconvolve_default"]["fingerprint"]
    with pytest.raises(ValueError):
        settings.RegistrationSettings(source_channel_names=["a"], target_channel_name="b",
                                      affine_transform_zyx=[[1, 0, 0], [0, 1, 0], [0, 0, 1]])


def test_estimate_resources_and_cluster(helpers_golden, monkeypatch):
    for rec in helpers_golden["estimate_resources"]:
        if rec["ci"]:
            monkeypatch.setenv("CI", rec["ci"])
        else:
            monkeypatch.delenv("CI", raising=False)
        if "shape" in rec:
            assert list(cluster.estimate_resources(tuple(rec["shape"]), **rec["kw"])) == rec["out"], rec
        else:
            got = [cluster.get_submitit_cluster(False, None), cluster.get_submitit_cluster(True, None),
                   cluster.get_submitit_cluster(False, "debug")]
            assert got == rec["cluster"]
    with pytest.raises(ValueError):
        cluster.estimate_resources((1, 2, 3))


def test_echo_resources_contract(capsys):
    cluster.echo_resources(np.int64(4), 16, 30)
    assert capsys.readouterr().out.strip() == 'RESOURCES:{"cpus": 4, "mem_gb": 16, "time_minutes": 30}'


def test_output_paths_and_sbatch(helpers_golden, tmp_path):
    ins = ["/data/in.zarr/A/1/0", "/data/in.zarr/B/2/0", "/data/other.zarr/A/1/0"]
    g = helpers_golden["output_paths"]
    assert [str(p) for p in paths.get_output_paths(ins, "/out/o.zarr")] == g["plain"]
    assert [str(p) for p in paths.get_output_paths(ins, "/out/o.zarr", ensure_unique_positions=True)] == g["unique"]
    f = tmp_path / "s.sh"
    f.write_text("#!/bin/bash\n#SBATCH --mem-per-cpu=16G\n#SBATCH --time=1:00:00\n#LOCAL --cpus-per-task=1\n# comment\n")
    assert paths.sbatch_to_submitit(str(f)) == helpers_golden["sbatch"]


def test_matrix_builders(helpers_golden):
    g = helpers_golden["matrices"]
    np.testing.assert_allclose(register.get_3D_rescaling_matrix((10, 20, 30), (1, 2, 0.5), (10, 40, 15)), g["rescale"])
    np.testing.assert_allclose(register.get_3D_rotation_matrix((10, 20, 30), 30.0, (10, 25, 35)), g["rotate"], atol=1e-12)
    np.testing.assert_allclose(register.get_3D_fliplr_matrix((10, 20, 30), (10, 20, 40)), g["fliplr"])
    # the reference's own known answers: tests/test_cli/test_register_cli.py:42-72
    ones = np.array([1, 1, 1])
    np.testing.assert_allclose(register.rescale_voxel_size(np.diag([2, 3, 4]), ones), [2, 3, 4])
    np.testing.assert_allclose(register.rescale_voxel_size(np.diag([2, -3, 4]), ones), [2, 3, 4])
    np.testing.assert_allclose(register.rescale_voxel_size(np.array([[0, 2, 0], [1, 0, 0], [0, 0, 3]]), ones), [2, 1, 3])
    p = register.convert_transform_to_ants(np.arange(16.0).reshape(4, 4))
    np.testing.assert_array_equal(p, [0, 1, 2, 4, 5, 6, 8, 9, 10, 3, 7, 11])
    np.testing.assert_allclose(register.convert_transform_to_numpy(p)[:3], np.arange(16.0).reshape(4, 4)[:3])


def test_find_lir_reference_known_answer():
    # tests/test_cli/test_register_cli.py:75-86 of the reference
    data = np.zeros((10, 10, 10))
    data[2:8, 0:9, 3:10] = 1
    z_slice, y_slice, x_slice = register.find_lir(data)
    assert (z_slice, y_slice, x_slice) == (slice(2, 8), slice(0, 9), slice(3, 10))
    assert register.largest_interior_rectangle(np.ones((4, 7), bool)) == (0, 0, 7, 4)
    m = np.zeros((6, 8), bool)
    m[1:5, 2:7] = True
    m[1, 2] = False  # a notch: the best rectangle avoids it
    x, y, w, h = register.largest_interior_rectangle(m)
    assert w * h == 16 and m[y:y + h, x:x + w].all()
    assert register.largest_interior_rectangle(np.zeros((3, 3), bool))[2:] == (0, 0)
