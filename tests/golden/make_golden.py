#!/usr/bin/env python3
"""Generate golden input/output vectors by running the *reference* (czbiohub-sf/biahub).

Run ONLY in the build container, where the reference is mounted at /root/reference:

    python tests/golden/make_golden.py

It imports the reference's own functions (through oracle/ref_import.py, which stubs the
absent orchestration packages), evaluates them on small seeded inputs and stores inputs,
parameters and outputs as data files next to this script.  No reference source is copied;
on the GPU box this script is inert (no /root/reference) and the tests read the fixtures.

Fixtures written:
  deskew_shapes.json      get_deskewed_data_shape table (+ the ValueError case)
  deskew_cases.npz        fast_deskew_zyx / _fast_deskew_czyx inputs+outputs
  average_n_slices.npz    _average_n_slices known answers
  transfer_function.npz   compute_tranfser_function (odd/even psf x odd/even volume)
  transform_scipy.npz     core.transform.Transform.apply (SciPy), orders 0/1
  transform_spline.npz    Transform.apply(order=3) (3-D and 2-D, cval, reference grid, integer dtypes) and the raw
                          register.apply_affine_transform(method="scipy") call (cubic spline, output = input shape)
  phase_cross_corr.npz    estimate_stabilization.phase_cross_corr (three normalisations)
  estimate_crop.npz       estimate_crop.estimate_crop_one_position on in-memory arrays (LIR from biahub_amd's restatement)
  legacy_fill.npz         deskew._fill_overhang_with_mean (legacy 6-connected SciPy dilation)
  deskew_transform_matrix.json  deskew._get_transform_matrix over a grid of (angle, ratio)
  concatenate.json        biahub.concatenate slicing/channel-layout helpers, ConcatenateSettings validation
  helpers.json            settings dumps, fingerprints, estimate_resources, output paths,
                          sbatch parsing, matrix builders
"""

from __future__ import annotations

import json
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))

from oracle.ref_import import load_reference, reference_available  # noqa: E402


def flat_field_vectors():
    """biahub/flat_field.py:56-155 — np.median pattern, flat_field_zyx, _flat_field_czyx (own generator: can run alone)."""
    import biahub.flat_field as FF

    rng = np.random.default_rng(20261004)
    ff = {}
    cases = [((8, 6, 4), np.uint16), ((9, 6, 4), np.uint16), ((2, 3, 5), np.uint16), ((1, 4, 4), np.uint16),
             ((8, 6, 4), np.float32), ((9, 7, 5), np.float32), ((16, 12, 9), np.uint16), ((33, 5, 70), np.uint16),
             ((40, 3, 130), np.float32), ((12, 4, 66), np.int16), ((7, 5, 9), np.uint8)]
    for j, (shape, dt) in enumerate(cases):
        lo = -2000 if dt == np.int16 else 1
        hi = 250 if dt == np.uint8 else 4095
        data = (rng.random(shape) * (hi - lo) + lo).astype(dt)
        if j in (6, 7):  # ties: many equal samples per pixel, as a 12-bit camera gives
            data = (data // 64 * 64 + 1).astype(dt)
        ff[f"in{j}"] = data
        ff[f"median{j}"] = FF._median_tiled(data, axis=0, tile_bytes=1)
        ff[f"flat{j}"] = FF.flat_field_zyx(data)
    czyx = (rng.random((3, 8, 6, 4)) * 4095 + 1).astype(np.uint16)
    ff["czyx_in"] = czyx
    ff["czyx_out"] = FF._flat_field_czyx(czyx, target_indices=[1])
    ax = (rng.random((7, 8, 9)) * 4095 + 1).astype(np.uint16)
    ff["axes_in"] = ax
    for a in range(3):
        ff[f"axes_median{a}"] = FF._median_tiled(ax, axis=a, tile_bytes=1)
    np.savez_compressed(HERE / "flat_field.npz", **ff)
    print("flat_field.npz written")


def detect_peaks_vectors():
    """biahub/characterize_psf.py:562-711 detect_peaks on seeded bead volumes (torch CPU), plus the raw pooling outputs."""
    import torch
    import torch.nn.functional as F

    import biahub.characterize_psf as CP

    rng = np.random.default_rng(20261005)
    dp = {}

    def beads(shape, n, seed):
        r = np.random.default_rng(seed)
        v = r.normal(110.0, 3.0, shape).astype(np.float32)
        zz, yy, xx = np.ogrid[: shape[0], : shape[1], : shape[2]]
        for _ in range(n):
            c = r.uniform(0, 1, 3) * np.array(shape)
            v += (r.uniform(400, 3000) * np.exp(-0.5 * (((zz - c[0]) / 1.5) ** 2 + ((yy - c[1]) / 1.2) ** 2
                                                         + ((xx - c[2]) / 1.2) ** 2))).astype(np.float32)
        return np.rint(v).astype(np.float32)

    cases = [((24, 40, 36), 12, dict(block_size=(8, 8, 8), nms_distance=3, min_distance=6, threshold_abs=200.0,
                                      max_num_peaks=50, exclude_border=(2, 3, 3), blur_kernel_size=3)),
             ((33, 50, 47), 25, dict(block_size=(16, 12, 8), nms_distance=4, min_distance=10, threshold_abs=300.0,
                                      max_num_peaks=30, exclude_border=None, blur_kernel_size=3)),
             ((20, 31, 29), 8, dict(block_size=(4, 4, 4), nms_distance=3, min_distance=0, threshold_abs=150.0,
                                     max_num_peaks=500, exclude_border=(1, 1, 1), blur_kernel_size=5))]
    for j, (shape, n, kw) in enumerate(cases):
        vol = beads(shape, n, 100 + j)
        dp[f"vol{j}"] = vol
        dp[f"kw{j}"] = np.array(json.dumps(kw))
        dp[f"peaks{j}"] = CP.detect_peaks(vol, **kw, device="cpu")
        t = torch.from_numpy(vol)[None, None]
        k, b = kw["blur_kernel_size"], kw["block_size"]
        sm = F.avg_pool3d(t, kernel_size=k, stride=1, padding=k // 2, count_include_pad=False)
        val, idx = F.max_pool3d(sm, kernel_size=b, stride=b, padding=tuple(x // 2 for x in b), return_indices=True)
        dp[f"pool_val{j}"], dp[f"pool_idx{j}"] = val.flatten().numpy(), idx.flatten().numpy()
    np.savez_compressed(HERE / "detect_peaks.npz", **dp)
    print("detect_peaks.npz written")


def pcc_chain_vectors():
    """estimate_stabilization.py:129-196 (phase_cross_corr_padding) and :259-310 (get_tform_from_pcc)."""
    import biahub.estimate_stabilization as ES

    rng = np.random.default_rng(20261006)
    pc = {}
    for j, (shape, roll) in enumerate([((10, 14, 12), (1, -2, 3)), ((9, 11, 13), (-2, 1, 0)), ((16, 20, 18), (3, 0, -4))]):
        ref = rng.random(shape, dtype=np.float32)
        mov = np.roll(ref, roll, axis=(0, 1, 2)) + 0.02 * rng.random(shape, dtype=np.float32)
        pc[f"ref{j}"], pc[f"mov{j}"] = ref, mov
        for norm in (None, "magnitude"):
            peak, corr = ES.phase_cross_corr_padding(ref, mov, normalization=norm)
            pc[f"peak{j}_{norm}"], pc[f"corr{j}_{norm}"] = np.asarray(peak), np.asarray(corr)
    stack = np.stack([pc["ref2"], pc["mov2"], np.roll(pc["ref2"], (0, 2, 1), axis=(0, 1, 2))])
    first = np.broadcast_to(stack[0], stack.shape).copy()
    pc["stack"] = stack
    for t in (1, 2):
        for ft in ("custom", "custom_padding"):
            tr, sh, _ = ES.get_tform_from_pcc(t, stack, first, function_type=ft, normalization="magnitude")
            pc[f"tform{t}_{ft}"], pc[f"tshift{t}_{ft}"] = np.asarray(tr), np.asarray(sh, dtype=np.float64)
    np.savez_compressed(HERE / "pcc_chain.npz", **pc)
    print("pcc_chain.npz written")


def binning_vectors():
    """biahub/process_data.py:29-105 binning_czyx."""
    import biahub.process_data as PD

    rng = np.random.default_rng(20261007)
    bn = {}
    cases = [((2, 4, 12, 16), np.uint16, (1, 2, 2), "sum"), ((2, 4, 12, 16), np.uint16, (2, 3, 4), "mean"),
             ((1, 6, 8, 10), np.uint8, (3, 2, 5), "sum"), ((3, 4, 6, 8), np.int16, (1, 2, 2), "mean"),
             ((2, 4, 8, 12), np.float32, (2, 2, 3), "sum"), ((2, 4, 8, 12), np.float32, (1, 4, 4), "mean"),
             ((1, 2, 4, 4), np.uint16, (1, 2, 2), "sum")]
    for j, (shape, dt, f, mode) in enumerate(cases):
        if dt == np.int16:
            data = (rng.random(shape) * 3000 + 5).astype(dt)
        elif dt == np.float32:
            data = (rng.random(shape) * 1000).astype(dt)
        else:
            data = (rng.random(shape) * (250 if dt == np.uint8 else 60000)).astype(dt)
        if j == 6:
            data[:] = 0  # all-zero channel: the stretch is skipped
        bn[f"in{j}"] = data
        bn[f"kw{j}"] = np.array(json.dumps({"binning_factor_zyx": list(f), "mode": mode}))
        bn[f"out{j}"] = PD.binning_czyx(data, binning_factor_zyx=f, mode=mode)
    np.savez_compressed(HERE / "binning.npz", **bn)
    print("binning.npz written")


def estimate_crop_vectors():
    """biahub.estimate_crop.estimate_crop_one_position -> estimate_crop.npz.  The reference reads stores through iohub/dask
    and takes the largest interior rectangle from a library this image lacks: it is pointed at in-memory arrays (an ndarray
    subclass with .compute()) and at biahub_amd.register's restatement of `lir`, so the fixture pins everything but that."""
    import biahub.estimate_crop as rc
    import biahub.register as rr
    from biahub_amd.register import largest_interior_rectangle

    class Lazy(np.ndarray):
        def compute(self):
            return np.asarray(self)

    class _Lir:
        @staticmethod
        def lir(mask):
            return np.array(largest_interior_rectangle(mask))

    rr.lir = _Lir
    rc.da.isnan = lambda a: np.isnan(a).view(Lazy)
    store = {}

    class _Data:
        def __init__(self, a):
            self._a = a

        def dask_array(self):
            return self._a.view(Lazy)

    class _Pos:
        def __init__(self, path):
            self.data = _Data(store[str(path)])

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    rc.open_ome_zarr = _Pos
    rng = np.random.default_rng(17)

    def dataset(T, C, shape, box, dtype, nan=False):
        a = np.zeros((T, C) + shape, dtype=dtype)
        (z0, z1), (y0, y1), (x0, x1) = box
        a[:, :, z0:z1, y0:y1, x0:x1] = (rng.random((T, C, z1 - z0, y1 - y0, x1 - x0)) * 100 + 1).astype(dtype)
        if nan:
            a[:, :, z0:z0 + 1, y0:y1, x0:x1] = np.nan
        return a

    cases = []
    # same shapes would hit the reference's undefined `_max_zyx_dims` when a mask radius is given, so every case with a
    # radius has different shapes; equal shapes are covered without radius
    lf = dataset(3, 2, (10, 30, 34), ((1, 9), (3, 27), (4, 31)), np.float32)
    ls = dataset(3, 1, (10, 30, 34), ((2, 10), (0, 25), (6, 34)), np.float32, nan=True)
    ls[1] = 0  # an empty time point: its count is far from the median and it is left out
    cases.append((lf, ls, None))
    lf2 = dataset(2, 1, (8, 40, 40), ((0, 8), (0, 40), (0, 40)), np.float32)
    ls2 = dataset(2, 2, (9, 38, 44), ((1, 8), (2, 36), (5, 40)), np.uint16)
    cases.append((lf2, ls2, 0.9))
    cases.append((lf2, ls2, None))
    lf3 = dataset(2, 1, (6, 24, 36), ((0, 6), (0, 24), (0, 36)), np.uint16)
    ls3 = dataset(2, 1, (7, 26, 36), ((0, 7), (1, 25), (2, 33)), np.uint16)
    cases.append((lf3, ls3, 0.8))
    out = {}
    for j, (a, b, radius) in enumerate(cases):
        store["p/A/1/0"], store["q/A/1/0"] = a, b
        res = rc.estimate_crop_one_position(Path("p/A/1/0"), Path("q/A/1/0"), lf_mask_radius=radius, output_dir=None)
        out[f"lf{j}"], out[f"ls{j}"] = a, b
        out[f"radius{j}"] = np.array(np.nan if radius is None else radius)
        out[f"crop{j}"] = np.array(res)
    np.savez_compressed(HERE / "estimate_crop.npz", **out)
    print("estimate_crop.npz written", [out[f"crop{j}"].tolist() for j in range(len(cases))])


def legacy_fill_vectors():
    """biahub.deskew._fill_overhang_with_mean (legacy path, SciPy 6-connected dilation) -> legacy_fill.npz."""
    from biahub.deskew import _fill_overhang_with_mean

    rng = np.random.default_rng(31)
    out = {}
    for j, (shape, it) in enumerate((((12, 20, 70), 3), ((9, 33, 41), 2), ((20, 16, 130), 3), ((6, 6, 6), 1), ((10, 12, 14), 5))):
        v = (rng.random(shape) * 200 + 50).astype(np.float32)
        # overhang-like wedges of exact zeros at both ends of x, plus isolated zeros inside the signal
        for z in range(shape[0]):
            w = int(shape[2] * 0.3 * z / max(1, shape[0] - 1))
            v[z, :, :w] = 0
            v[shape[0] - 1 - z, :, shape[2] - w:] = 0
        idx = tuple(rng.integers(0, n, 5) for n in shape)
        v[idx] = 0
        out[f"in{j}"] = v
        out[f"it{j}"] = np.array(it)
        out[f"out{j}"] = _fill_overhang_with_mean(v, dilation_iterations=it)
    np.savez_compressed(HERE / "legacy_fill.npz", **out)
    print("legacy_fill.npz written")


def concatenate_vectors():
    """biahub.concatenate helpers + ConcatenateSettings validation -> concatenate.json.  The one reference function that
    opens stores (get_channel_combiner_metadata) is pointed at biahub_amd's reader through the module attribute it looks
    `open_ome_zarr` up by; the plates it reads are made here with biahub_amd.io and described in the fixture."""
    import biahub.concatenate as rc
    from biahub.settings import ConcatenateSettings
    from biahub_amd import io

    out = {}
    params = ["all", [2, 9], [[0, 4], [3, 7]], [[0, 4], "all", [1, 5]], [[[0, 2], [4, 6]], "all"], ["all", [1, 3]]]
    out["get_path_slice_param"] = [
        {"param": p_, "index": i, "total": n, "result": rc.get_path_slice_param(p_, i, n)}
        for p_ in params for n in (2, 4) for i in range(n)]
    out["get_slice"] = []
    for p_ in ("all", [0, 5], [3, 3], [7, 2], [1], [1, 2, 3], [0.5, 2], "x", None):
        try:
            s_ = rc.get_slice(p_, 11)
            out["get_slice"].append({"param": p_, "max": 11, "result": [s_.start, s_.stop]})
        except ValueError as e:
            out["get_slice"].append({"param": p_, "max": 11, "error": str(e)})
    out["create_path_slicing_params"] = []
    for z, y, x in (("all", "all", "all"), ([1, 3], "all", [0, 8]), ([0, 2], [2, 6], [4, 12])):
        r = rc.create_path_slicing_params(z, y, x, (3, 2, 5, 9, 13))
        out["create_path_slicing_params"].append({"z": z, "y": y, "x": x, "shape": [3, 2, 5, 9, 13], "result": [[q.start, q.stop] for q in r]})
    out["calculate_cropped_size"] = [
        {"slices": sl, "result": list(rc.calculate_cropped_size([slice(*q) for q in sl]))}
        for sl in ([[0, 4], [0, 6], [0, 8]], [[2, 4], [6, 1], [3, 3]])]
    out["validate_slicing_params_zyx"] = []
    for group in ([[[0, 4], [0, 6], [0, 8]], [[1, 5], [2, 8], [4, 12]]], [[[0, 4], [0, 6], [0, 8]], [[0, 4], [0, 5], [0, 8]]]):
        try:
            rc.validate_slicing_params_zyx([[slice(*q) for q in sl] for sl in group])
            out["validate_slicing_params_zyx"].append({"slices": group, "error": None})
        except ValueError as e:
            out["validate_slicing_params_zyx"].append({"slices": group, "error": str(e)})
    # settings: accepted / rejected configurations
    base = {"concat_data_paths": ["a/*/*/*", "b/*/*/*"], "channel_names": ["all", ["GFP"]]}
    trials = [{}, {"X_slice": [0, 10]}, {"X_slice": [[0, 10], [5, 15]]}, {"X_slice": [[0, 10], "all"]}, {"X_slice": "half"},
              {"X_slice": [-1, 5]}, {"X_slice": [[0, 10]]}, {"X_slice": [[0, 10], [1, 2], [3, 4]]}, {"Y_slice": [[0, 1, 2], [3, 4]]},
              {"Z_slice": [[[0, 2], [4, 6]], "all"]}, {"Z_slice": [[[0, 2], [4, -6]], "all"]}, {"Z_slice": [[[0, 2], "x"], "all"]},
              {"Z_slice": [[[0, 2]], 5]}, {"Z_slice": ["all", "all"]}, {"Z_slice": 3},
              {"chunks_czyx": [1, 4, 8, 8]}, {"chunks_czyx": [4, 8, 8]}, {"shards_ratio": [1, 1, 2, 2, 2]},
              {"time_indices": [0, 2]}, {"time_indices": 1}, {"time_indices": "some"}, {"output_ome_zarr_version": "0.4"},
              {"output_ome_zarr_version": None}, {"output_ome_zarr_version": "0.3"}, {"channel_names": "all"},
              {"concat_data_paths": "a"}, {"unknown": 1}, {"ensure_unique_positions": True}]
    out["settings"] = []
    for t_ in trials:
        cfg = {**base, **t_}
        try:
            out["settings"].append({"config": cfg, "dump": ConcatenateSettings(**cfg).model_dump()})
        except Exception as e:  # pydantic ValidationError
            msgs = sorted({err["msg"] for err in e.errors()}) if hasattr(e, "errors") else [str(e)]
            out["settings"].append({"config": cfg, "errors": msgs})
    # channel layout over real (small) plates
    rc.open_ome_zarr = io.open_ome_zarr
    rc.natsorted = lambda seq: sorted(seq, key=lambda s_: [int(k) if k.isdigit() else k for k in __import__("re").split(r"(\d+)", s_)])
    out["channel_combiner"] = []
    with tempfile.TemporaryDirectory() as tmp:
        tmp = Path(tmp)
        plates = {"p1": (["DAPI", "Cy5"], [("A", "1", "0"), ("A", "2", "0"), ("A", "10", "0")]),
                  "p2": (["GFP", "RFP", "DAPI"], [("A", "1", "0"), ("A", "2", "0"), ("A", "10", "0")]),
                  "p3": (["BF"], [("B", "1", "0")])}
        for name, (chans, poss) in plates.items():
            io.create_empty_plate(tmp / f"{name}.zarr", poss, chans, (2, len(chans), 4, 6, 8), dtype=np.uint16)
        out["channel_combiner_plates"] = {k: {"channels": v[0], "positions": ["/".join(p_) for p_ in v[1]], "shape": [2, len(v[0]), 4, 6, 8]} for k, v in plates.items()}
        cases = [(["p1.zarr/*/*/*", "p2.zarr/*/*/*"], ["all", "all"], ["all", "all", "all"]),
                 (["p1.zarr/*/*/*", "p2.zarr/*/*/*"], [["Cy5"], ["RFP", "GFP", "nope"]], [[0, 2], [[0, 3], [3, 6]], "all"]),
                 (["p2.zarr/A/1/0", "p1.zarr/A/*/0", "p3.zarr/*/*/*"], ["all", ["DAPI"], "all"], ["all", [1, 5], [[0, 4], [2, 6], [4, 8]]]),
                 (["p1.zarr/*/*/*", "p2.zarr/*/*/*"], ["all", "all"], [[[0, 2], [0, 3]], "all", "all"])]
        for globs, chans, slicing in cases:
            rec = {"globs": globs, "channels": chans, "slicing_zyx": slicing}
            try:
                paths, names, cin, cout, sl = rc.get_channel_combiner_metadata([str(tmp / g) for g in globs], chans, slicing)
                rec.update(paths=[str(Path(p_).relative_to(tmp)) for p_ in paths], names=names, input_idx=cin, output_idx=cout,
                           slices=[[[q.start, q.stop] for q in s_] for s_ in sl])
            except ValueError as e:
                rec["error"] = str(e)
            out["channel_combiner"].append(rec)
    (HERE / "concatenate.json").write_text(json.dumps(out, indent=1))
    print("concatenate.json written")


def spline_vectors():
    """core/transform.py:374-396 Transform.apply with order=3 (and the integer-dtype rounding of orders 1 / 3), and
    register.py:256-272 apply_affine_transform(method="scipy"): SciPy's cubic B-spline resampling -> transform_spline.npz."""
    import biahub.register as R
    from biahub.core.transform import Transform

    rng = np.random.default_rng(20261005)
    sp = {}

    def rot(deg_z, deg_y, scale, shift):
        a, b = np.deg2rad(deg_z), np.deg2rad(deg_y)
        Rz = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
        Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
        M = np.eye(4)
        M[:3, :3] = scale * (Rz @ Ry)
        M[:3, 3] = shift
        return M

    # 3-D, two shapes, rotation + fractional shift
    for j, (shape, M) in enumerate([((12, 16, 20), rot(7.0, 0.0, 1.02, (0.4, 1.75, -2.25))),
                                    ((9, 33, 14), rot(-11.0, 4.0, 0.97, (-0.6, 2.3, 1.15))),
                                    ((5, 6, 7), rot(3.0, 0.0, 1.0, (0.25, -0.5, 0.75)))]):
        mov = rng.random(shape, dtype=np.float32) * 1000
        t = Transform(M)
        sp[f"mov{j}"], sp[f"M{j}"] = mov, M
        sp[f"o3_{j}"] = t.apply(mov, order=3)
        sp[f"o3_cval_{j}"] = t.apply(mov, order=3, cval=37.5)
    ref = np.zeros((10, 20, 18), dtype=np.float32)
    sp["o3_ref_0"] = Transform(sp["M0"]).apply(sp["mov0"], reference=ref, order=3, cval=-2.0)
    # identity and integer shift: the spline interpolates the samples
    sp["o3_identity"] = Transform(np.eye(4)).apply(sp["mov0"], order=3)
    sp["o3_shift_int"] = Transform.from_translation([-3.0, 1.0, 4.0]).apply(sp["mov0"], order=3)
    # integer dtypes: SciPy rounds (and clamps) the float64 result into the input dtype
    u16 = (rng.random((12, 16, 20)) * 4000 + 100).astype(np.uint16)
    u16[3:5, 4:9, 6:12] = 65535  # overshoot above the type's range
    u16[8:10, 2:5, 3:7] = 0      # and below it
    sp["u16"] = u16
    sp["u16_o3"] = Transform(sp["M0"]).apply(u16, order=3)
    sp["u16_o1"] = Transform(sp["M0"]).apply(u16, order=1)
    i16 = (rng.random((12, 16, 20)) * 6000 - 3000).astype(np.int16)
    sp["i16"] = i16
    sp["i16_o3"] = Transform(sp["M0"]).apply(i16, order=3)
    sp["i16_o1"] = Transform(sp["M0"]).apply(i16, order=1)
    # 2-D images
    th = np.deg2rad(9.0)
    M2 = np.array([[np.cos(th), -np.sin(th), 1.3], [np.sin(th), np.cos(th), -2.7], [0, 0, 1.0]])
    img = rng.random((24, 31), dtype=np.float32) * 100
    sp["img2d"], sp["M2d"] = img, M2
    sp["img2d_o3"] = Transform(M2).apply(img, order=3, cval=5.0)
    sp["img2d_o1"] = Transform(M2).apply(img, order=1)
    # the raw register.py:271-272 call: pull matrix, order 3, output shape = input shape whatever output_shape_zyx says
    Mp = rot(5.0, 0.0, 1.01, (0.3, -1.2, 2.6))
    vol = rng.random((10, 18, 22), dtype=np.float32) * 500
    vol[2, 3, 4] = np.nan  # np.nan_to_num(nan=0) comes first (:254)
    sp["reg_vol"], sp["reg_M"] = vol, Mp
    sp["reg_out"] = R.apply_affine_transform(vol, Mp, (12, 20, 24), method="scipy")
    sp["reg_out_crop"] = R.apply_affine_transform(vol, Mp, (10, 18, 22), method="scipy",
                                                  crop_output_slicing=(slice(1, 9), slice(2, 15), slice(3, 20)))
    sp["reg_u16"] = u16
    sp["reg_u16_out"] = R.apply_affine_transform(u16, sp["M0"], u16.shape, method="scipy")
    np.savez_compressed(HERE / "transform_spline.npz", **sp)


def deskew_transform_matrix_vectors():
    """biahub/deskew.py:180-210 — _get_transform_matrix over a grid of (angle, ratio), incl. the example settings."""
    import biahub.deskew as D

    cases = [(36.17, 0.371), (30.0, 0.25), (45.0, 1.0), (0.5, 0.05), (22.5, 0.6543), (36.0, 0.375)]
    out = [{"ls_angle_deg": a, "px_to_scan_ratio": r, "matrix": np.asarray(D._get_transform_matrix(a, r), dtype=np.float64).tolist()}
           for a, r in cases]
    (HERE / "deskew_transform_matrix.json").write_text(json.dumps(out, indent=1))
    print("deskew_transform_matrix.json written")


def main():
    if sys.argv[1:] == ["deskew_transform_matrix"]:
        load_reference()
        deskew_transform_matrix_vectors()
        return 0
    if sys.argv[1:] == ["estimate_crop"]:
        load_reference()
        estimate_crop_vectors()
        return 0
    if sys.argv[1:] == ["spline"]:
        load_reference()
        spline_vectors()
        return 0
    if sys.argv[1:] == ["legacy_fill"]:
        load_reference()
        legacy_fill_vectors()
        return 0
    if sys.argv[1:] == ["concatenate"]:
        load_reference()
        concatenate_vectors()
        return 0
    if sys.argv[1:] == ["binning"]:
        load_reference()
        binning_vectors()
        return 0
    if sys.argv[1:] == ["pcc_chain"]:
        load_reference()
        pcc_chain_vectors()
        return 0
    if sys.argv[1:] == ["detect_peaks"]:
        load_reference()
        detect_peaks_vectors()
        return 0
    if sys.argv[1:] == ["flat_field"]:
        load_reference()
        flat_field_vectors()
        return 0
    if not reference_available():
        print("reference checkout not present; nothing to do")
        return 0
    load_reference()
    import torch

    import biahub.deskew as D
    from biahub.core.transform import Transform
    from biahub.deconvolve import compute_tranfser_function

    torch.set_num_threads(4)
    rng = np.random.default_rng(20261003)

    # ---- 1. shape table ---------------------------------------------------------------
    rows = []
    grid = [
        ((64, 256, 256), 36.17, 0.371, True, 3, 0.116),
        ((512, 2048, 2048), 36.17, 0.371, True, 3, 0.116),
        ((256, 1024, 1024), 36.17, 0.371, True, 3, 0.116),
        ((2, 3, 4), 36, 0.386, True, 1, 1.0),
        ((20, 14, 9), 36.17, 0.371, True, 1, 1.0),
        ((20, 14, 9), 36.17, 0.371, True, 2, 1.0),
        ((33, 17, 12), 30.0, 0.25, True, 3, 0.2),
        ((40, 31, 24), 36.17, 0.371, False, 3, 0.116),
        ((40, 31, 24), 45.0, 0.8, False, 2, 0.116),
        ((100, 50, 7), 10.0, 0.5, False, 4, 1.0),
        ((10, 500, 100), 30, 0.1, True, 1, 1.0),
    ]
    for shape, ang, r, ko, n, px in grid:
        out, vox = D.get_deskewed_data_shape(shape, ang, r, ko, n, px)
        rows.append(
            dict(shape=list(shape), angle=ang, ratio=r, keep_overhang=ko, n=n, pixel=px,
                 out=[int(v) for v in out], voxel=[float(v) for v in vox])
        )
    err = None
    try:
        D.get_deskewed_data_shape((10, 500, 100), 30, 0.1, keep_overhang=False)
    except ValueError as e:  # tests/test_cli/test_deskew_cli.py:189-197
        err = str(e)
    json.dump({"rows": rows, "error_case": {"shape": [10, 500, 100], "angle": 30, "ratio": 0.1,
                                            "message": err}},
              open(HERE / "deskew_shapes.json", "w"), indent=1)

    # ---- 2/3. fast_deskew_zyx and the CZYX adapter ------------------------------------
    store = {}
    meta = []

    def add_case(name, vol, ang, r, ko, n, fill, splits=None):
        if splits is None:
            out = D.fast_deskew_zyx(torch.from_numpy(vol.astype(np.float32)), ang, r, ko, n, fill).numpy()
        else:
            out = D._fast_deskew_czyx(vol[None], device="cpu", num_splits=splits, ls_angle_deg=ang,
                                      px_to_scan_ratio=r, keep_overhang=ko, average_n_slices=n,
                                      overhang_fill=fill)
        store[name + "__in"] = vol
        store[name + "__out"] = out.astype(np.float32)
        meta.append(dict(name=name, angle=ang, ratio=r, keep_overhang=ko, n=n,
                         fill=fill, splits=splits))

    vols = {
        "a": rng.random((20, 14, 9), dtype=np.float32),
        "b": rng.random((33, 17, 12), dtype=np.float32),
        "c": rng.random((40, 31, 24), dtype=np.float32),
        "u": rng.integers(90, 4000, (24, 16, 10)).astype(np.uint16),
    }
    # sprinkle exact zeros so the zero-mask of the overhang fill sees interior zeros too
    vols["b"][5, 3, 2] = 0.0
    vols["c"][20:22, 10:12, 5] = 0.0
    i = 0
    for key in ("a", "b", "c"):
        for n in (1, 2, 3):
            for fill in (0, 100.0, "mean"):
                if key == "c" and fill == 100.0:
                    continue
                add_case(f"d{i:02d}_{key}_n{n}", vols[key], 36.17, 0.371, True, n, fill)
                i += 1
    add_case("k0_c_n3_nooverhang", vols["c"], 36.17, 0.371, False, 3, 0)
    add_case("k1_c_n2_nooverhang", vols["c"], 45.0, 0.8, False, 2, 0)
    add_case("p0_b_n3_angle30", vols["b"], 30.0, 0.25, True, 3, 0)
    add_case("p1_a_n4", vols["a"], 20.0, 0.5, True, 4, "mean")
    add_case("u0_uint16_n3", vols["u"], 36.17, 0.371, True, 3, 0, splits=1)
    add_case("u1_uint16_n3_mean", vols["u"], 36.17, 0.371, True, 3, "mean", splits=1)
    for s in (1, 2, 3):
        add_case(f"s{s}_c_split", vols["c"], 36.17, 0.371, True, 3, 0, splits=s)
    store["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(HERE / "deskew_cases.npz", **store)

    # ---- 4. _average_n_slices ----------------------------------------------------------
    data = np.arange(1, 17).reshape(4, 2, 2)
    np.savez(HERE / "average_n_slices.npz", data=data,
             w3=D._average_n_slices(data, 3), w2=D._average_n_slices(data, 2),
             w1=D._average_n_slices(data, 1))

    # ---- 5. transfer function ----------------------------------------------------------
    tf = {}
    for j, (ps, vs) in enumerate([((5, 5, 5), (8, 9, 10)), ((4, 6, 5), (9, 12, 8)),
                                  ((7, 3, 3), (16, 8, 8)), ((9, 9, 9), (9, 9, 9))]):
        psf = rng.random(ps, dtype=np.float32)
        tf[f"psf{j}"] = psf
        tf[f"shape{j}"] = np.array(vs)
        tf[f"tf{j}"] = compute_tranfser_function(psf, vs)
    np.savez_compressed(HERE / "transfer_function.npz", **tf)

    # ---- 6. Transform.apply (SciPy path) ----------------------------------------------
    tr = {}
    mov = rng.random((12, 16, 20), dtype=np.float32)
    th = np.deg2rad(7.0)
    M = np.array([[1.02, 0.0, 0.0, 0.4],
                  [0.0, np.cos(th), -np.sin(th), 1.75],
                  [0.0, np.sin(th), np.cos(th), -2.25],
                  [0, 0, 0, 1.0]])
    t = Transform(M)
    tr["moving"] = mov
    tr["matrix"] = M
    tr["order1"] = t.apply(mov, order=1)
    tr["order0"] = t.apply(mov, order=0)
    ref_shape = np.zeros((10, 20, 18), dtype=np.float32)
    tr["order1_ref"] = t.apply(mov, reference=ref_shape, order=1, cval=3.0)
    tr["inv"] = t.invert().matrix
    tr["compose"] = (t @ Transform.from_translation([1, 2, 3])).matrix
    pts = rng.random((5, 3)) * 10
    tr["points"] = pts
    tr["points_out"] = t.apply_points(pts)
    tr["shift_int"] = Transform.from_translation([-3.0, 1.0, 4.0]).apply(np.ones((10, 10, 10), np.float32))
    np.savez_compressed(HERE / "transform_scipy.npz", **tr)
    spline_vectors()

    # ---- 6b. phase cross-correlation (estimate_stabilization.py:199-256) ---------------
    import biahub.estimate_stabilization as ES

    pc = {}
    for j, (shape, roll) in enumerate([((8, 12, 10), (1, -2, 3)), ((9, 7, 11), (-3, 2, 0)), ((16, 16, 16), (5, 0, -7))]):
        ref = rng.random(shape, dtype=np.float32)
        mov = np.roll(ref, roll, axis=(0, 1, 2)) + 0.05 * rng.random(shape, dtype=np.float32)
        pc[f"ref{j}"], pc[f"mov{j}"] = ref, mov
        for norm in (None, "magnitude", "classic"):
            sh, corr = ES.phase_cross_corr(ref, mov, normalization=norm)
            pc[f"shift{j}_{norm}"], pc[f"corr{j}_{norm}"] = np.asarray(sh), np.asarray(corr)
    np.savez_compressed(HERE / "phase_cross_corr.npz", **pc)

    # ---- 7. settings / helpers ---------------------------------------------------------
    import biahub.register as R
    import biahub.settings as S
    from biahub.cli.parsing import sbatch_to_submitit
    from biahub.utils.cluster import estimate_resources, get_submitit_cluster
    from biahub.utils.config import settings_fingerprint, yaml_to_model
    from biahub.utils.ngff import get_output_paths

    helpers = {}
    sdir = Path("/root/reference/settings")
    for fname, model in [("example_deskew_settings.yml", S.DeskewSettings),
                         ("example_registration_settings.yml", S.RegistrationSettings),
                         ("example_stabilize_timelapse_settings.yml", S.StabilizationSettings)]:
        m = yaml_to_model(sdir / fname, model)
        helpers[fname] = {"yaml": (sdir / fname).read_text(), "dump": m.model_dump(mode="json"),
                          "fingerprint": settings_fingerprint(m)}
    m = S.DeconvolveSettings()
    helpers["deconvolve_default"] = {"dump": m.model_dump(mode="json"),
                                     "fingerprint": settings_fingerprint(m)}
    m = S.DeskewSettings(pixel_size_um=0.116, ls_angle_deg=36.1749, scan_step_um=0.3125)
    helpers["deskew_derived_ratio"] = m.model_dump(mode="json")
    res = []
    for ci in ("true", None):
        if ci:
            os.environ["CI"] = ci
        else:
            os.environ.pop("CI", None)
        for shape, kw in [((3, 6, 4, 5, 6), dict(ram_multiplier=8, time_multiplier=0.5, max_num_cpus=16)),
                          ((4, 2, 256, 1024, 1024), dict(ram_multiplier=8, time_multiplier=0.5, max_num_cpus=16)),
                          ((100, 3, 512, 2048, 2048), dict(ram_multiplier=16, max_num_cpus=16)),
                          ((1, 1, 64, 256, 256), dict())]:
            res.append({"ci": ci, "shape": list(shape), "kw": kw,
                        "out": [int(v) for v in estimate_resources(shape, **kw)]})
        res.append({"ci": ci, "cluster": [get_submitit_cluster(False, None), get_submitit_cluster(True, None),
                                          get_submitit_cluster(False, "debug")]})
    os.environ.pop("CI", None)
    helpers["estimate_resources"] = res
    ins = [Path("/data/in.zarr/A/1/0"), Path("/data/in.zarr/B/2/0"), Path("/data/other.zarr/A/1/0")]
    helpers["output_paths"] = {
        "plain": [str(p) for p in get_output_paths(ins, Path("/out/o.zarr"))],
        "unique": [str(p) for p in get_output_paths(ins, Path("/out/o.zarr"), ensure_unique_positions=True)],
    }
    with tempfile.NamedTemporaryFile("w", suffix=".sh", delete=False) as f:
        f.write("#!/bin/bash\n#SBATCH --mem-per-cpu=16G\n#SBATCH --time=1:00:00\n#LOCAL --cpus-per-task=1\n# comment\n")
    helpers["sbatch"] = sbatch_to_submitit(f.name)
    os.unlink(f.name)
    helpers["matrices"] = {
        "rescale": R.get_3D_rescaling_matrix((10, 20, 30), (1, 2, 0.5), (10, 40, 15)).tolist(),
        "rotate": R.get_3D_rotation_matrix((10, 20, 30), 30.0, (10, 25, 35)).tolist(),
        "fliplr": R.get_3D_fliplr_matrix((10, 20, 30), (10, 20, 40)).tolist(),
        "rescale_voxel": R.rescale_voxel_size(np.array(M[:3, :3]), np.array([0.2, 0.1, 0.1])).tolist(),
    }
    json.dump(helpers, open(HERE / "helpers.json", "w"), indent=1, default=str)
    flat_field_vectors()
    detect_peaks_vectors()
    pcc_chain_vectors()
    binning_vectors()
    deskew_transform_matrix_vectors()
    total = sum(p.stat().st_size for p in HERE.glob("*.np*")) + sum(p.stat().st_size for p in HERE.glob("*.json"))
    print(f"golden fixtures written to {HERE} ({total/1e6:.2f} MB)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
