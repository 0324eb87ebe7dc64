"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/bhcore.h declares.

No compute call is made here (there is no GPU); host-only entry points (bh_deskew_shape, error
strings, argument validation that fails before touching the device) are exercised.
"""

import ctypes as C
import json
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def _declared_symbols():
    text = (ROOT / "include" / "bhcore.h").read_text()
    return sorted(set(re.findall(r"\b(bh_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported(lib_built):
    lib = C.CDLL(str(lib_built))
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/bhcore.h but not exported by libbhcore.so"


def test_python_binding_covers_header(lib_built):
    from biahub_amd import _lib

    assert sorted(_lib.SIGNATURES) == _declared_symbols()
    lib = _lib.load()
    assert lib.bh_abi_version() == 1


def test_deskew_shape_through_abi(lib_built):
    from biahub_amd.deskew import get_deskewed_data_shape

    tab = json.load(open(GOLDEN / "deskew_shapes.json"))
    for r in tab["rows"]:
        out, vox = get_deskewed_data_shape(r["shape"], r["angle"], r["ratio"], r["keep_overhang"], r["n"], r["pixel"])
        assert list(out) == r["out"], r
        np.testing.assert_allclose(vox, r["voxel"], rtol=1e-14)
    e = tab["error_case"]
    with pytest.raises(ValueError, match="Dataset contains only overhang") as ei:
        get_deskewed_data_shape(e["shape"], e["angle"], e["ratio"], keep_overhang=False)
    assert str(ei.value) == e["message"]  # same text as biahub/deskew.py:263-267


def test_code_object_targets_gfx950(lib_built):
    data = open(lib_built, "rb").read()
    assert b"gfx950" in data
    assert b"gfx942" not in data and b"sm_" not in data


def test_ops_fail_loudly_without_gpu(lib_built):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from biahub_amd.deskew import _fast_deskew_czyx

    from biahub_amd.register import apply_affine_transform

    with pytest.raises(RuntimeError, match="no CPU path|no GPU visible"):   # no operator but deskew has a host path
        apply_affine_transform(np.zeros((4, 4, 4), np.float32), np.eye(4), (4, 4, 4))
    with pytest.raises(RuntimeError, match="no GPU visible"):
        _fast_deskew_czyx(np.zeros((1, 4, 4, 4), np.float32), device="cuda", ls_angle_deg=30, px_to_scan_ratio=0.3,
                          keep_overhang=True)


def test_product_never_imports_oracle():
    # the oracle is test infrastructure; nothing under biahub_amd/ may reference it
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|importlib.*oracle|__import__\(.*oracle", re.M)
    for p in (ROOT / "biahub_amd").rglob("*.py"):
        assert not pat.search(p.read_text()), f"{p} imports the oracle"


def _build_c_host(tmp_path):
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    exe = tmp_path / "c_host"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(ROOT / "examples" / "c_host.c"),
                    f"-L{ROOT / 'biahub_amd'}", "-lbhcore", f"-Wl,-rpath,{ROOT / 'biahub_amd'}", "-lm", "-o", str(exe)],
                   check=True)
    return exe


def test_plain_c_host_links_and_queries_geometry(tmp_path):
    """The boundary is a C-ABI: a C99 program with only include/bhcore.h links the library and runs the host-only calls."""
    import subprocess

    r = subprocess.run([str(_build_c_host(tmp_path))], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "deskewed shape (86, 256, 380)" in r.stdout and "invalid geometry refused" in r.stdout


@pytest.mark.gpu
def test_plain_c_host_runs_deskew_on_gpu(tmp_path):
    import subprocess

    r = subprocess.run([str(_build_c_host(tmp_path)), "gpu"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "fill value 499.9" in r.stdout or "fill value 500.0" in r.stdout


def test_invalid_arguments_return_status_not_crash(lib_built):
    """Every entry point validates its arguments before touching the device: BH_ERR_INVALID + a message, on any host."""
    from biahub_amd import _lib

    lib = _lib.load()
    I64x3, F64x3, Ix3 = C.c_int64 * 3, C.c_double * 3, C.c_int * 3
    bad = [
        lib.bh_deskew(None, None, _lib.DT_F32, 4, 4, 4, 30.0, 0.5, 1, 1, 0, 0.0, None, None),
        lib.bh_overhang_fill(None, None, 4, 4, 4, 1, 0.0, 3, None),
        lib.bh_richardson_lucy(None, None, None, 3, 3, 3, 8, 8, 8, 10, 1e-6, None),
        lib.bh_tikhonov(None, None, None, 8, 8, 8, 1e-3, None),
        lib.bh_transfer_function(None, None, 3, 3, 3, 8, 8, 8, None),
        lib.bh_phase_cross_corr(None, None, None, 8, 8, 8, 0, None, None),
        lib.bh_affine(None, None, _lib.DT_F32, 4, 4, 4, None, 1, 0, 0.0, None, 4, 4, 4, None),
        lib.bh_crop_flip(None, None, 4, 1, 4, 4, 4, None, 2, 2, 2, 0, 0, 0, None),
        lib.bh_flat_field(None, None, _lib.DT_U16, 4, 4, 4, None, None, None),
        lib.bh_median_z(None, None, _lib.DT_U16, 4, 4, 4, None),
        lib.bh_image_stats(None, None, 4, 4, 4, None),
        lib.bh_mattes_mi(None, None, 4, 4, 4, None, 4, 4, 4, None, None, 32, 1, 0, None, None, None),
        lib.bh_sobel(None, None, 4, 4, 4, None),
        lib.bh_patch_peaks(None, None, 4, 4, 4, None, 0, None, 2.0, None),
        lib.bh_average_patches(None, None, 4, 4, 4, None, 1, None, 1, None),
        lib.bh_smooth_shrink(None, None, 0, 4, 4, F64x3(0, 0, 0), Ix3(1, 1, 1), None, I64x3(), I64x3()),
        lib.bh_smooth_shrink(None, None, 4, 4, 4, F64x3(0, 0, 0), Ix3(0, 1, 1), None, I64x3(), I64x3()),
        lib.bh_block_peaks(None, None, 4, 4, 4, 2, Ix3(2, 2, 2), None, None, I64x3()),
        lib.bh_ctx_destroy(None) if False else _lib.BH_ERR_INVALID,
    ]
    assert all(s == _lib.BH_ERR_INVALID for s in bad), bad
    assert _lib.last_error()  # the thread-local message is set
    # host-only geometry queries succeed without a context
    shape, off = I64x3(), I64x3()
    assert lib.bh_smooth_shrink(None, None, 13, 20, 31, F64x3(2, 1, 0), Ix3(3, 2, 1), None, shape, off) == _lib.BH_OK
    assert tuple(shape) == (4, 10, 31) and tuple(off) == (1, 0, 0)
    nb = I64x3()
    assert lib.bh_block_peaks(None, None, 24, 40, 36, 3, Ix3(8, 8, 8), None, None, nb) == _lib.BH_OK
    assert tuple(nb) == (4, 6, 5)  # torch: floor((N + 2*(b//2) - b) / b) + 1


def test_richardson_lucy_plan_boxes(monkeypatch):
    """The back-end choice is host logic (bh_richardson_lucy_plan): power-of-two and 3 * 2^k shapes run the fused engine as
    they are, awkward ones at a wrap-padded box when that is cheaper than the 7-smooth library box."""
    from biahub_amd.deconvolve import richardson_lucy_plan as plan

    assert plan((33, 17, 17), (512, 2048, 2048)) == ((512, 2048, 2048), "engine")
    assert plan((33, 17, 17), (384, 1024, 1024)) == ((384, 1024, 1024), "engine")
    assert plan((33, 17, 17), (342, 1024, 1517)) == ((384, 1024, 1536), "engine-padded")       # 342 + 32 <= 3 * 128, 1517 + 16 <= 3 * 512
    assert plan((33, 17, 17), (384, 1024, 1536)) == ((384, 1024, 1536), "engine")              # 3 * 2^k on z and x
    assert plan((33, 17, 17), (1068, 256, 1664)) == ((1280, 256, 2048), "engine-padded")       # a mantis position: z -> 5 * 256
    assert plan((33, 17, 17), (683, 2048, 3034)) == ((768, 2048, 3072), "engine-padded")       # deskewed config 2: 8-row X passes
    assert plan((33, 17, 17), (683, 2048, 3100)) == ((720, 2048, 3125), "library")             # x beyond the engine's 3072
    assert plan((5, 5, 5), (15, 42, 50)) == ((15, 42, 50), "library")                          # 7-smooth and small
    # rows that would fit 5 * 2^k take the next 3 * 2^k instead: wave-private X passes and no fold pass beat the smaller box
    assert plan((33, 17, 17), (342, 1024, 2100)) == ((384, 1024, 3072), "engine-padded")
    assert plan((33, 17, 17), (342, 1024, 1100)) == ((384, 1024, 1536), "engine-padded")
    monkeypatch.setenv("BH_RL_X5", "1")
    assert plan((33, 17, 17), (342, 1024, 2100)) == ((384, 1024, 2560), "engine-padded")
    monkeypatch.delenv("BH_RL_X5")
    monkeypatch.setenv("BH_RL_ENGINE_PAD", "0")
    assert plan((33, 17, 17), (342, 1024, 1517)) == ((375, 1024, 1536), "library")
    monkeypatch.setenv("BH_RL_ENGINE_PAD", "1")
    assert plan((7, 5, 9), (21, 64, 150)) == ((32, 64, 192), "engine-padded")
    assert plan((9, 9, 3), (40, 70, 64)) == ((40, 96, 64), "engine-padded")                    # z = 5 * 8 runs as it is
    assert plan((5, 5, 5), (70, 150, 300)) == ((80, 160, 320), "engine-padded")
    assert plan((5, 3, 11), (19, 40, 134)) == ((24, 64, 192), "engine-padded")
    monkeypatch.setenv("BH_FC_NORADIX3", "1")
    assert plan((5, 3, 11), (19, 40, 134)) == ((32, 64, 256), "engine-padded")
    with pytest.raises(ValueError):
        plan((9, 9, 9), (4, 64, 64))
