"""CPU: pin the oracle (oracle/oracle_np.py) against vectors captured from the reference.

The fixtures under tests/golden/ were produced by tests/golden/make_golden.py, which imports the
reference's own functions (czbiohub-sf/biahub) in the build container.
"""

import json

import numpy as np
import pytest

from conftest import GOLDEN, rel_err
from oracle import oracle_np as O


def test_deskew_shape_table():
    tab = json.load(open(GOLDEN / "deskew_shapes.json"))
    for r in tab["rows"]:
        out, vox = O.get_deskewed_data_shape(r["shape"], r["angle"], r["ratio"], r["keep_overhang"], r["n"], r["pixel"])
        assert list(out) == r["out"], r
        np.testing.assert_allclose(vox, r["voxel"], rtol=1e-15)
    e = tab["error_case"]
    with pytest.raises(ValueError, match="Dataset contains only overhang") as ei:
        O.get_deskewed_data_shape(e["shape"], e["angle"], e["ratio"], False)
    assert str(ei.value) == e["message"]


def test_average_n_slices_known_answers():
    # the reference's own known-answer test: tests/test_cli/test_deskew_cli.py:11-30
    z = np.load(GOLDEN / "average_n_slices.npz")
    data = z["data"]
    assert np.array_equal(O.average_n_slices(data, 3), np.array([[[5, 6], [7, 8]], [[13, 14], [15, 16]]]))
    assert np.array_equal(O.average_n_slices(data, 2), np.array([[[3, 4], [5, 6]], [[11, 12], [13, 14]]]))
    assert np.array_equal(O.average_n_slices(data, 1), data)
    for k, w in (("w3", 3), ("w2", 2), ("w1", 1)):
        assert np.array_equal(O.average_n_slices(data, w), z[k])


def test_deskew_oracle_matches_reference(deskew_cases):
    z, meta = deskew_cases
    assert len(meta) >= 30
    for m in meta:
        vol = z[m["name"] + "__in"]
        ref = z[m["name"] + "__out"]
        kw = dict(ls_angle_deg=m["angle"], px_to_scan_ratio=m["ratio"], keep_overhang=m["keep_overhang"],
                  average_n_slices=m["n"], overhang_fill=m["fill"])
        if m["splits"] is None:
            got = O.fast_deskew_zyx(vol.astype(np.float32), **kw)
        else:
            got = O.fast_deskew_czyx(vol[None], num_splits=m["splits"], **kw)
        assert got.shape == ref.shape, m
        # coordinates are restated bit-exactly; what is left is summation order (<= 2 ulp)
        assert rel_err(got, ref) <= 5e-7, (m, rel_err(got, ref))


def test_transfer_function_oracle_matches_reference():
    z = np.load(GOLDEN / "transfer_function.npz")
    for j in range(4):
        tf = O.compute_transfer_function(z[f"psf{j}"], tuple(z[f"shape{j}"]))
        assert tf.shape == z[f"tf{j}"].shape
        assert rel_err(tf, z[f"tf{j}"]) <= 2e-6


def test_scipy_affine_oracle_matches_reference():
    z = np.load(GOLDEN / "transform_scipy.npz")
    M, mov = z["matrix"], z["moving"]
    assert rel_err(O.transform_apply_scipy(mov, M, order=1), z["order1"]) <= 2e-6
    assert np.array_equal(O.transform_apply_scipy(mov, M, order=0), z["order0"])
    got = O.transform_apply_scipy(mov, M, output_shape=(10, 20, 18), order=1, cval=3.0)
    assert rel_err(got, z["order1_ref"]) <= 2e-6
    np.testing.assert_allclose(np.linalg.inv(M), z["inv"], rtol=1e-12, atol=1e-12)
    # integer translation, the reference's own test (tests/test_affine.py:43-59) through SciPy
    assert np.array_equal(O.transform_apply_scipy(np.ones((10, 10, 10), np.float32),
                                                  np.array([[1, 0, 0, -3.0], [0, 1, 0, 1.0], [0, 0, 1, 4.0], [0, 0, 0, 1]])),
                          z["shift_int"])


def test_spline_oracle_matches_reference_and_scipy():
    """SciPy's cubic-spline resampling restated (oracle spline_*): the reference's own Transform.apply(order=3) and
    apply_affine_transform(method="scipy") outputs (core/transform.py:374-396, register.py:271-272), and scipy itself."""
    import scipy.ndimage as ndi

    z = np.load(GOLDEN / "transform_spline.npz")
    for j in range(3):
        mov, M = z[f"mov{j}"], z[f"M{j}"]
        assert rel_err(O.transform_apply_spline(mov, M), z[f"o3_{j}"]) <= 1e-6
        assert rel_err(O.transform_apply_spline(mov, M, cval=37.5), z[f"o3_cval_{j}"]) <= 1e-6
    assert rel_err(O.transform_apply_spline(z["mov0"], z["M0"], (10, 20, 18), -2.0), z["o3_ref_0"]) <= 1e-6
    assert rel_err(O.transform_apply_spline(z["mov0"], np.eye(4)), z["o3_identity"]) <= 1e-6
    sh = np.array([[1, 0, 0, -3.0], [0, 1, 0, 1.0], [0, 0, 1, 4.0], [0, 0, 0, 1]])
    assert rel_err(O.transform_apply_spline(z["mov0"], sh), z["o3_shift_int"]) <= 1e-6
    for k in ("u16", "i16"):  # integers: round half away from zero, saturate
        assert np.array_equal(O.transform_apply_spline(z[k], z["M0"]), z[k + "_o3"])
    assert rel_err(O.transform_apply_spline(z["img2d"], z["M2d"], cval=5.0), z["img2d_o3"]) <= 1e-6
    assert rel_err(O.apply_affine_transform_scipy(z["reg_vol"], z["reg_M"], (12, 20, 24)), z["reg_out"]) <= 1e-6
    crop = (slice(1, 9), slice(2, 15), slice(3, 20))
    assert rel_err(O.apply_affine_transform_scipy(z["reg_vol"], z["reg_M"], None, crop), z["reg_out_crop"]) <= 1e-6
    assert np.array_equal(O.apply_affine_transform_scipy(z["reg_u16"], z["M0"]), z["reg_u16_out"])
    # the prefilter against scipy's, incl. axes of length 1 and 2
    rng = np.random.default_rng(3)
    for shape in [(5,), (2,), (7, 3), (6, 9, 11), (1, 5, 4), (40, 3, 2)]:
        v = rng.random(shape)
        np.testing.assert_allclose(O.spline_prefilter(v), ndi.spline_filter(v, 3, output=np.float64, mode="mirror"),
                                   rtol=0, atol=1e-13)


def test_itk_mode_reference_tests():
    # tests/test_affine.py:26-59 restated for the ITK boundary mode of the oracle
    ones = np.ones((10, 10, 10))
    for interp in ("linear", "nearestneighbor"):
        r = O.apply_affine_transform(ones, np.eye(4), (10, 10, 10), interpolation=interp)
        assert r.shape == (10, 10, 10) and np.all(r == 1)
    m = np.eye(4)
    m[:3, -1] = [-3, 1, 4]
    r = O.apply_affine_transform(ones, m, (10, 10, 10))
    assert np.all(r[3:10, 0:9, 0:6] == 1)
    assert np.all(r[0:3] == 0) and np.all(r[:, 9:] == 0) and np.all(r[:, :, 6:] == 0)


def test_rl_and_tikhonov_oracle_sanity():
    # parity unpinned (no reference arithmetic available): check defining properties instead
    rng = np.random.default_rng(0)
    psf = O.gaussian_psf((5, 5, 5), (1.0, 1.0, 1.0))
    truth = np.zeros((16, 20, 24), np.float32)
    truth[8, 10, 12] = 100.0
    truth[4, 5, 6] = 50.0
    otf = O.rl_otf(psf, truth.shape)
    blurred = np.fft.irfftn(otf * np.fft.rfftn(truth), s=truth.shape).astype(np.float32)
    assert abs(blurred.sum() - truth.sum()) / truth.sum() < 1e-5  # unit-sum PSF keeps flux
    assert np.unravel_index(blurred.argmax(), blurred.shape) == (8, 10, 12)  # PSF centred at the origin
    est = O.richardson_lucy_zyx(blurred + 1.0, psf, iterations=20)
    assert est.min() >= 0
    assert est[8, 10, 12] > (blurred + 1.0)[8, 10, 12] * 1.5  # sharpened
    assert abs(est.sum() - (blurred + 1).sum()) / (blurred + 1).sum() < 1e-3  # RL conserves flux
    tf = O.compute_transfer_function(psf, truth.shape)
    dec = O.tikhonov_zyx(blurred, tf, 1e-3)
    assert dec[8, 10, 12] > blurred[8, 10, 12]
    # reg -> infinity kills the output, reg = 0 with H == 1 is the identity
    assert rel_err(O.tikhonov_zyx(blurred, np.ones_like(tf), 0.0), blurred) < 1e-5
    x = rng.random(truth.shape).astype(np.float32)
    a = O.tikhonov_zyx(x, tf, 1e-2)
    b = O.tikhonov_zyx(2 * x, tf, 1e-2)
    assert rel_err(b, 2 * a) < 1e-5  # linear


def test_crop_flip_oracle():
    a = np.arange(2 * 3 * 4 * 5, dtype=np.uint16).reshape(2, 3, 4, 5)
    s = [slice(1, 3), slice(0, 2), slice(2, 5)]
    assert np.array_equal(O.copy_n_paste_czyx(a, s), a[:, 1:3, 0:2, 2:5])
    f = a[0].astype(np.float32)
    f[1, 1, 1] = np.nan
    out = O.copy_n_paste(f, s)
    assert not np.isnan(out).any()
    assert np.array_equal(O.flip_zyx(a[0], x=True, y=True), a[0][:, ::-1, ::-1])


def test_phase_cross_corr_oracle_matches_reference():
    z = np.load(GOLDEN / "phase_cross_corr.npz")
    for j in range(3):
        for norm in (None, "magnitude", "classic"):
            sh, corr = O.phase_cross_corr(z[f"ref{j}"], z[f"mov{j}"], norm)
            assert np.array_equal(sh, z[f"shift{j}_{norm}"]), (j, norm)
            assert rel_err(corr, z[f"corr{j}_{norm}"]) <= 1e-5, (j, norm)


def test_flat_field_golden():
    """biahub/flat_field.py:56-155 through the oracle; values captured from the reference import."""
    z = np.load(GOLDEN / "flat_field.npz")
    for j in range(11):
        data = z[f"in{j}"]
        med = O.median_axis(data, 0)
        assert med.dtype == z[f"median{j}"].dtype and np.array_equal(med, z[f"median{j}"])
        got = O.flat_field_zyx(data)
        assert got.dtype == z[f"flat{j}"].dtype and np.array_equal(got, z[f"flat{j}"])
    assert np.array_equal(O.flat_field_czyx(z["czyx_in"], [1]), z["czyx_out"])
    for a in range(3):
        assert np.array_equal(O.median_axis(z["axes_in"], a), z[f"axes_median{a}"])


def test_detect_peaks_golden():
    """characterize_psf.detect_peaks through the oracle: pooling outputs bit-equal to torch, peaks equal to the reference."""
    import json

    z = np.load(GOLDEN / "detect_peaks.npz")
    for j in range(3):
        kw = json.loads(str(z[f"kw{j}"]))
        kw["block_size"] = tuple(kw["block_size"])
        kw["exclude_border"] = tuple(kw["exclude_border"]) if kw["exclude_border"] else None
        val, idx = O.block_peaks(z[f"vol{j}"], kw["blur_kernel_size"], kw["block_size"])
        assert np.array_equal(val, z[f"pool_val{j}"]) and np.array_equal(idx, z[f"pool_idx{j}"])
        assert np.array_equal(O.detect_peaks(z[f"vol{j}"], **kw), z[f"peaks{j}"])


def test_pcc_chain_golden():
    """phase_cross_corr_padding / get_tform_from_pcc through the oracle (estimate_stabilization.py:129-196, 259-310)."""
    z = np.load(GOLDEN / "pcc_chain.npz")
    for j in range(3):
        for norm in (None, "magnitude"):
            peak, corr = O.phase_cross_corr_padding(z[f"ref{j}"], z[f"mov{j}"], normalization=norm)
            assert tuple(peak) == tuple(z[f"peak{j}_{norm}"]) and corr.shape == z[f"corr{j}_{norm}"].shape
            assert rel_err(corr, z[f"corr{j}_{norm}"]) <= 1e-5
    stack = z["stack"]
    first = np.broadcast_to(stack[0], stack.shape)
    for t in (1, 2):
        for ft in ("custom", "custom_padding"):
            tr, sh, _ = O.get_tform_from_pcc(t, stack, first, ft, "magnitude")
            assert np.array_equal(tr, z[f"tform{t}_{ft}"]) and np.array_equal(np.asarray(sh, float), z[f"tshift{t}_{ft}"])


def test_binning_golden():
    """process_data.binning_czyx through the oracle (values captured from the reference import)."""
    import json

    z = np.load(GOLDEN / "binning.npz")
    for j in range(7):
        kw = json.loads(str(z[f"kw{j}"]))
        got = O.binning_czyx(z[f"in{j}"], tuple(kw["binning_factor_zyx"]), kw["mode"])
        assert got.dtype == z[f"out{j}"].dtype and np.array_equal(got, z[f"out{j}"]), j


def test_legacy_fill_overhang_with_mean_matches_reference():
    """oracle fill_overhang_with_mean (SciPy 6-connected dilation) against the reference's _fill_overhang_with_mean."""
    z = np.load(GOLDEN / "legacy_fill.npz")
    for j in range(5):
        got = O.fill_overhang_with_mean(z[f"in{j}"], int(z[f"it{j}"]))
        assert np.array_equal(got, z[f"out{j}"]), j


def test_transform_from_skimage_and_text():
    """Transform.from_skimage / repr / str as biahub/core/transform.py:169-227,530-538 (checked against the reference's
    own class while this fixture was written: identical matrices and strings)."""
    from biahub_amd.core.transform import Transform

    class SimilarityTransform:  # stands for skimage.transform.SimilarityTransform: only `.params` and the class name matter
        params = np.array([[0.9, -0.1, 10.0], [0.1, 0.9, 20.0], [0.0, 0.0, 1.0]])

    t = Transform.from_skimage(SimilarityTransform(), ndim=3)
    assert t.transform_type == "similarity" and np.array_equal(t.matrix[1:3, 1:3], SimilarityTransform.params[:2, :2])
    assert np.array_equal(t.matrix[:, 0], [1, 0, 0, 0]) and np.array_equal(t.translation, [0, 10, 20])
    assert repr(t) == "Transform(ndim=3, type='similarity', translation=[0.0, 10.0, 20.0])"
    assert str(t).splitlines()[0] == "Transform(similarity, 3D)" and "0.9 -0.1 10." in str(t)
    assert Transform.from_skimage(SimilarityTransform(), ndim=2).ndim == 2

    class Odd:
        params = np.eye(4)

    assert Transform.from_skimage(Odd(), ndim=3).transform_type == "affine"
    with pytest.raises(ValueError, match="Cannot convert 3D"):
        Transform.from_skimage(Odd(), ndim=2)
    Odd.params = np.eye(5)
    with pytest.raises(ValueError, match="Unexpected skimage transform shape"):
        Transform.from_skimage(Odd(), ndim=3)


def test_estimate_crop_oracle_matches_reference():
    """oracle estimate_crop_arrays against the reference's estimate_crop_one_position (fixture made with the same LIR)."""
    from biahub_amd.register import find_lir

    z = np.load(GOLDEN / "estimate_crop.npz")
    for j in range(4):
        radius = None if np.isnan(z[f"radius{j}"]) else float(z[f"radius{j}"])
        got = O.estimate_crop_arrays(z[f"lf{j}"], z[f"ls{j}"], radius, find_lir)
        assert np.array_equal(np.array(got), z[f"crop{j}"]), (j, got)



def test_wave_private_x_pass_model():
    """tools/xw_model.py is the executable specification of csrc/fftconv_xw.inc (index maps, LDS addresses, twiddles,
    untangle pairing, stored column order): its own checks against numpy's FFT must hold for both row lengths in use."""
    import importlib.util
    from pathlib import Path

    spec = importlib.util.spec_from_file_location("xw_model", Path(__file__).resolve().parent.parent / "tools" / "xw_model.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    for logm in (8, 9, 10):
        m.check(logm)
        x = m.XW(logm)
        # lane 0 owns the two self-mirrored groups, every other lane a group and the group of the mirrored frequencies
        for lam in range(1, x.lg):
            g0, g1 = x.group_of(lam, 0), x.group_of(lam, 1)
            assert all(x.mirror(8 * g0 + n) == 8 * g1 + 7 - n for n in range(8))
        assert (x.group_of(0, 0), x.group_of(0, 1)) == (0, 1)


def test_radix3_wave_private_x_pass_model():
    """tools/x3_model.py is the executable specification of csrc/fftconv_x3.inc (rows of 1536 / 3072 voxels: radix-3 across
    the thirds of a row, three register stages per third, LDS addresses and swizzles, untangle pairing of third 0 across
    neighbouring lanes and of thirds 1 <-> 2 inside a lane, stored column order): checked against numpy's FFT."""
    import importlib.util
    import sys
    from pathlib import Path

    tools = Path(__file__).resolve().parent.parent / "tools"
    sys.path.insert(0, str(tools))
    try:
        spec = importlib.util.spec_from_file_location("x3_model", tools / "x3_model.py")
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
    finally:
        sys.path.remove(str(tools))
    for logl in (8, 9):
        m.check(logl, verbose=False)
        x = m.X3(logl)
        cyc, _ = x.lds_cycles()
        for name, (actual, ideal) in cyc.items():   # every exchange access within 2x of conflict-free, DA / DB conflict-free
            assert actual <= 2 * ideal, (name, actual, ideal)
            assert not name.startswith(("DA", "DB")) or actual == ideal, (name, actual, ideal)
        # third 0: the mirrored group sits in the neighbouring lane; thirds 1 <-> 2: complementary groups in one lane
        for l in range(2, x.lg):
            assert all(x.mirror_pos(x.dc(0, l, n)) == x.dc(0, l ^ 1, 7 - n) for n in range(8))
        for l in range(x.lg):
            assert x.group(1, l) + x.group(2, l) == x.lg - 1


def test_get_transform_matrix_golden():
    """D2: the deskew pull matrix of the product's host mirror, against the reference's own outputs
    (biahub/deskew.py:180-210; fixture: tests/golden/make_golden.py deskew_transform_matrix)."""
    import json

    from biahub_amd.deskew import _get_transform_matrix

    cases = json.load(open(GOLDEN / "deskew_transform_matrix.json"))
    assert len(cases) >= 6
    for c in cases:
        got = _get_transform_matrix(c["ls_angle_deg"], c["px_to_scan_ratio"])
        assert got.shape == (4, 4) and np.array_equal(got, np.asarray(c["matrix"])), c


def test_product_average_n_slices_golden():
    """D7: the product's `_average_n_slices` host helper (not only the oracle's) on the reference's known answers
    (biahub/deskew.py:43-68, tests/test_cli/test_deskew_cli.py:11-30)."""
    from biahub_amd.deskew import _average_n_slices

    z = np.load(GOLDEN / "average_n_slices.npz")
    for w in (1, 2, 3):
        got = _average_n_slices(z["data"], w)
        assert got.shape == z[f"w{w}"].shape and np.array_equal(got, z[f"w{w}"]), w


def test_register_stage_column_pass_model():
    """tools/colw_model.py is the executable specification of csrc/fftconv_colw.inc: index maps, the three-stage column
    transform against numpy's FFT, and conflict-free LDS exchanges with the kernel's padding rule (128 B per 32 rows)."""
    import importlib.util
    import sys
    from pathlib import Path

    tools = Path(__file__).resolve().parent.parent / "tools"
    sys.path.insert(0, str(tools))
    try:
        spec = importlib.util.spec_from_file_location("colw_model", tools / "colw_model.py")
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        for logn in (8, 9, 10):
            tot = m.check(logn, lambda row: (row >> 5) * 8, verbose=False)
            assert all(c == i for c, i in tot.values()), (logn, tot)
    finally:
        sys.path.remove(str(tools))


def test_oracle_inverse_filter_z_padding_mirrors_edge_planes():
    """waveorder's util.pad_zyx_along_z (recalled; parity unpinned): pad planes are the flipped first / last z_padding planes
    when z_padding < Z, zeros otherwise.  With H = 1 and reg = 0 the filter is the identity, so the cropped output is the
    input whatever the padding; a delta transfer function along z that shifts by one plane exposes the pad planes."""
    rng = np.random.default_rng(3)
    x = rng.random((4, 3, 5)).astype(np.float32)
    for pad in (2, 6):
        Zp = 4 + 2 * pad
        shift = np.zeros((Zp, 3, 5))
        shift[1, 0, 0] = 1.0                                  # convolution kernel: out[z] = in[z - 1]
        H = np.fft.fftn(shift)
        # inverse filter conj(H) / |H|^2 undoes the shift: out[z] = in[z + 1]; the last kept plane shows the first pad plane
        got = O.wo_apply_inverse_transfer_function(x, H, pad, 0.0, False)
        assert np.allclose(got[:-1], x[1:], atol=1e-5)
        want_last = x[-1] if pad < 4 else np.zeros_like(x[-1])   # mirrored edge plane, or the zero fallback
        assert np.allclose(got[-1], want_last, atol=1e-5), pad


def test_oracle_richardson_lucy_against_spatial_domain_restatement():
    """Richardson-Lucy has no reference implementation (parity unpinned by construction); what the oracle must be is the
    DEFINITION in DESIGN.md 2.3.  This restates that definition in the spatial domain — circular convolution and correlation as
    explicit sums of rolled copies, no FFT anywhere — and holds the FFT oracle to it, for odd and even PSF extents (the centre
    convention: tap k of an axis of extent K sits at offset k - K // 2)."""
    rng = np.random.default_rng(4)
    for shape, pshape in (((6, 7, 9), (3, 3, 5)), ((5, 8, 6), (2, 4, 3)), ((4, 4, 4), (3, 1, 2))):
        d = (rng.random(shape) * 50 + 5).astype(np.float64)
        d[0, 0, 0] = -3.0                                   # clipped by e0 = max(d, 0)
        psf = rng.random(pshape) + 0.1
        h = psf / psf.sum()

        def conv(x, flip):
            out = np.zeros_like(x)
            for k in np.ndindex(*pshape):
                off = tuple(ki - K // 2 for ki, K in zip(k, pshape))
                # (h * x)[n] = sum_k h[k] x[n - off_k];  correlation: x[n + off_k]
                out += h[k] * np.roll(x, tuple(-o if flip else o for o in off), axis=(0, 1, 2))
            return out

        est = np.maximum(d, 0.0)
        for _ in range(4):
            ratio = d / np.maximum(conv(est, False), 1e-6)
            est = np.maximum(est * conv(ratio, True), 0.0)
        got = O.richardson_lucy_zyx(d.astype(np.float32), psf.astype(np.float32), iterations=4, eps=1e-6)
        assert np.abs(got - est).max() <= 2e-5 * np.abs(est).max(), (shape, pshape)
