"""GPU: parity of the HIP path (through the C-ABI) against the reference's golden vectors and the oracle.

Tolerances (relative to the volume's max |value|):
  deskew            <= 1e-5  (coordinates restated bit-exactly; only summation order differs)
  overhang fill     <= 1e-5  (mean computed in float64 here, cascade float32 in torch)
  transfer function <= 1e-5 ; Tikhonov / Richardson-Lucy <= 1e-4 (FFT ordering)
  affine linear     <= 1e-5 ; nearest / crop / flip / NaN->0 bit-exact
"""

import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT, rel_err
from oracle import oracle_np as O

pytestmark = pytest.mark.gpu

DESKEW_TOL = 1e-5
FFT_TOL = 1e-4


# ----------------------------------------------------------------------------- deskew
def test_deskew_golden_vectors(gpu, deskew_cases):
    from biahub_amd.deskew import _fast_deskew_czyx, fast_deskew_zyx

    z, meta = deskew_cases
    for m in meta:
        vol = z[m["name"] + "__in"]
        ref = z[m["name"] + "__out"]
        kw = dict(ls_angle_deg=m["angle"], px_to_scan_ratio=m["ratio"], keep_overhang=m["keep_overhang"],
                  average_n_slices=m["n"], overhang_fill=m["fill"])
        if m["splits"] is None:
            got = fast_deskew_zyx(torch.from_numpy(vol.astype(np.float32)).to(gpu), **kw).cpu().numpy()
        else:
            got = _fast_deskew_czyx(vol[None], device="cuda", num_splits=m["splits"], **kw)
        assert got.shape == ref.shape, m
        assert got.dtype == np.float32
        assert rel_err(got, ref) <= DESKEW_TOL, (m, rel_err(got, ref))


@pytest.mark.parametrize("shape,n,angle,ratio", [
    ((64, 256, 256), 3, 36.17, 0.371),   # BASELINE config 1
    ((48, 100, 70), 3, 36.17, 0.371),    # Y % N != 0 (last-slab rule), X not a tile multiple
    ((96, 64, 200), 2, 30.0, 0.25),
    ((40, 33, 65), 5, 20.0, 0.5),        # generic-N path
    ((31, 50, 33), 1, 45.0, 0.9),
    ((40, 30, 128), 2, 30.0, 0.25),      # X a multiple of 64: the persistent double-buffered kernel (two x tiles)
    ((31, 50, 64), 1, 45.0, 0.9),        # ... with a ratio at which a lane's four outputs span more than three z rows
    ((33, 17, 192), 4, 36.17, 0.371),    # ... N = 4, ragged last slab
    ((200, 40, 320), 3, 36.17, 0.371),   # ... several x' chunks, windows leaving the volume on both sides
])
def test_deskew_vs_oracle(gpu, shape, n, angle, ratio):
    from biahub_amd.deskew import fast_deskew_zyx

    rng = np.random.default_rng(sum(shape))
    vol = rng.random(shape, dtype=np.float32) * 1000
    want = O.fast_deskew_zyx(vol, angle, ratio, True, n, 0)
    got = fast_deskew_zyx(torch.from_numpy(vol).to(gpu), angle, ratio, True, n, 0).cpu().numpy()
    assert got.shape == want.shape
    assert rel_err(got, want) <= DESKEW_TOL


def test_deskew_uint16_equals_float32_input(gpu):
    from biahub_amd.deskew import _fast_deskew_czyx

    rng = np.random.default_rng(5)
    vol = rng.integers(0, 65535, (40, 60, 90)).astype(np.uint16)
    kw = dict(ls_angle_deg=36.17, px_to_scan_ratio=0.371, keep_overhang=True, average_n_slices=3, overhang_fill=0)
    a = _fast_deskew_czyx(vol[None], device="cuda", **kw)
    b = _fast_deskew_czyx(vol[None].astype(np.float32), device="cuda", **kw)
    assert np.array_equal(a, b)  # widening on load is exact


def test_deskew_fill_modes_vs_oracle(gpu):
    from biahub_amd.deskew import fast_deskew_zyx

    rng = np.random.default_rng(11)
    vol = (rng.random((48, 70, 50), dtype=np.float32) * 500 + 100).astype(np.float32)
    vol[10:14, 20:30, 10:20] = 0  # true zeros inside the signal are masked too (deskew.py:361)
    for fill in ("mean", 100.0):
        want = O.fast_deskew_zyx(vol, 36.17, 0.371, True, 3, fill)
        got = fast_deskew_zyx(torch.from_numpy(vol).to(gpu), 36.17, 0.371, True, 3, fill).cpu().numpy()
        assert rel_err(got, want) <= DESKEW_TOL, fill
    # keep_overhang=False never fills (deskew.py:538)
    a = fast_deskew_zyx(torch.from_numpy(vol).to(gpu), 36.17, 0.371, False, 3, "mean").cpu().numpy()
    b = fast_deskew_zyx(torch.from_numpy(vol).to(gpu), 36.17, 0.371, False, 3, 0).cpu().numpy()
    assert np.array_equal(a, b)


def test_deskew_properties_large(gpu):
    """Size-independent properties at a size the oracle would not finish quickly."""
    from biahub_amd.deskew import fast_deskew_zyx, get_deskewed_data_shape

    Z, Y, X = 256, 512, 512
    g = torch.Generator(device=gpu).manual_seed(1)
    a = torch.rand((Z, Y, X), generator=g, device=gpu)
    b = torch.rand((Z, Y, X), generator=g, device=gpu)
    kw = dict(ls_angle_deg=36.17, px_to_scan_ratio=0.371, keep_overhang=True, average_n_slices=3)
    da, db, dab = fast_deskew_zyx(a, **kw), fast_deskew_zyx(b, **kw), fast_deskew_zyx(2 * a + b, **kw)
    assert tuple(da.shape) == get_deskewed_data_shape((Z, Y, X), 36.17, 0.371, True, 3)[0]
    assert float((dab - (2 * da + db)).abs().max()) <= 1e-5 * float(dab.abs().max())  # linearity
    ones = fast_deskew_zyx(torch.ones((Z, Y, X), device=gpu), **kw)
    assert float(ones.max()) <= 1.0 + 1e-6 and float(ones.min()) >= 0.0  # partition of unity, zero overhang
    # the Y (coverslip) axis is independent: deskewing a slab equals the slab of the deskew (flipped index)
    sub = fast_deskew_zyx(a[:, :, 100:164].contiguous(), **kw)
    assert torch.equal(sub, da[:, X - 164:X - 100, :])
    # splits along X reproduce the unsplit result exactly
    assert torch.equal(fast_deskew_zyx(a, **kw), da)  # deterministic


def test_deskew_persistent_kernel_equals_tile_kernel(gpu, monkeypatch):
    """float32 volumes whose rows are whole 64-column tiles take the persistent double-buffered kernel (loader / sampler
    wavefronts, a lane owns four consecutive x'); BH_DESKEW_PERS=0 keeps them on the one-tile-per-workgroup kernel.  Same sample
    positions, same operation order per output: bit-identical without fill; with a fill the two differ only in the summation
    order of the mean.  True zeros inside the signal and a row length that is not a multiple of 4 (unaligned 16-byte stores)."""
    from biahub_amd.deskew import fast_deskew_zyx

    for shape, n, ratio in (((96, 100, 256), 3, 0.371), ((64, 33, 128), 2, 0.29), ((50, 21, 64), 1, 0.8)):
        g = torch.Generator(device=gpu).manual_seed(sum(shape))
        vol = (torch.rand(shape, generator=g, device=gpu) * 500 + 10).round_()
        vol[10:20, 5:15, 20:40] = 0.0
        for fill in (0, "mean", 7.5):
            kw = dict(ls_angle_deg=36.17, px_to_scan_ratio=ratio, keep_overhang=True, average_n_slices=n, overhang_fill=fill)
            monkeypatch.setenv("BH_DESKEW_PERS", "1")   # also without a fill (the default takes it only with one)
            new = fast_deskew_zyx(vol, **kw)
            monkeypatch.setenv("BH_DESKEW_PERS", "0")
            old = fast_deskew_zyx(vol, **kw)
            monkeypatch.delenv("BH_DESKEW_PERS")
            if fill == "mean":
                assert float((new - old).abs().max()) <= 1e-6 * float(old.abs().max()), (shape, fill)
            else:
                assert torch.equal(new, old), (shape, fill)


@pytest.mark.parametrize("shape,n,angle,ratio", [
    ((64, 256, 256), 3, 36.17, 0.371),   # BASELINE config 1
    ((200, 40, 320), 3, 36.17, 0.371),   # several x' chunks, windows leaving the volume on both sides
    ((33, 17, 192), 4, 36.17, 0.371),    # N = 4, ragged last slab
    ((31, 50, 64), 1, 45.0, 0.9),
    ((40, 30, 128), 2, 30.0, 0.25),
    ((48, 100, 70), 3, 36.17, 0.371),    # X not a tile multiple, Y % N != 0
    ((40, 33, 65), 5, 20.0, 0.5),        # generic-N kernel
])
def test_deskew_one_pass_fill(gpu, monkeypatch, shape, n, angle, ratio):
    """The overhang is filled in ONE pass (csrc/deskew_rows.inc): the zero pattern and
    its dilation from geometry, the mean from row sums of the input, whole rows written by the resampling kernel.  Against the
    oracle (the reference's mask / dilate / mean / where on the finished volume, deskew.py:339-368), against the mask pipeline
    (BH_DESKEW_ONEPASS=0: same voxels bit for bit outside the fill, the fill value to float32 rounding), with row sums handed
    in, and with data zeros, where the conditional mask pipeline behind the kernel must take over (path 2)."""
    from biahub_amd.deskew import deskew_fill_path, fast_deskew_zyx

    rng = np.random.default_rng(sum(shape) + n)
    vol = (rng.random(shape, dtype=np.float32) * 900 + 50).astype(np.float32)
    t = torch.from_numpy(vol).to(gpu)
    for fill in ("mean", 123.5):
        kw = dict(ls_angle_deg=angle, px_to_scan_ratio=ratio, keep_overhang=True, average_n_slices=n, overhang_fill=fill)
        want = O.fast_deskew_zyx(vol, angle, ratio, True, n, fill)
        got = fast_deskew_zyx(t, **kw)
        assert deskew_fill_path(gpu) == 1, (shape, fill)
        assert rel_err(got.cpu().numpy(), want) <= DESKEW_TOL, (shape, fill)
        monkeypatch.setenv("BH_DESKEW_ONEPASS", "0")
        old = fast_deskew_zyx(t, **kw)
        assert deskew_fill_path(gpu) == 0
        monkeypatch.delenv("BH_DESKEW_ONEPASS")
        same = got == old
        fv_new, fv_old = got[~same], old[~same]  # differ only where the fill value went, by its rounding
        if fill == "mean":
            assert fv_new.numel() == 0 or (fv_new.unique().numel() == 1 and fv_old.unique().numel() == 1)
            assert float((got - old).abs().max()) <= 1e-6 * float(old.abs().max()), (shape, fill)
        else:
            assert bool(same.all()), (shape, fill)
    # the persistent kernel's one-pass form (float32, whole 64-column tiles, N <= 4) against the tile kernel's
    if shape[2] % 64 == 0 and n <= 4:
        kwm = dict(ls_angle_deg=angle, px_to_scan_ratio=ratio, keep_overhang=True, average_n_slices=n, overhang_fill="mean")
        a = fast_deskew_zyx(t, **kwm)
        monkeypatch.setenv("BH_DESKEW_ROWS_KERNEL", "pers")
        b = fast_deskew_zyx(t, **kwm)
        monkeypatch.delenv("BH_DESKEW_ROWS_KERNEL")
        assert deskew_fill_path(gpu) == 1 and torch.equal(a, b), shape
    # uint16 camera counts take the same path (widened on load, row sums reduced from the integers)
    u16 = torch.from_numpy((vol * 40).astype(np.uint16)).to(gpu)
    got16 = fast_deskew_zyx(u16, angle, ratio, True, n, "mean")
    assert deskew_fill_path(gpu) == 1
    want16 = O.fast_deskew_zyx((vol * 40).astype(np.uint16).astype(np.float32), angle, ratio, True, n, "mean")
    assert rel_err(got16.cpu().numpy(), want16) <= DESKEW_TOL, shape
    # row sums handed in by the caller: same result as reducing them here
    rs = t.to(torch.float64).sum(dim=2).contiguous()
    kw = dict(ls_angle_deg=angle, px_to_scan_ratio=ratio, keep_overhang=True, average_n_slices=n, overhang_fill="mean")
    a, b = fast_deskew_zyx(t, **kw), fast_deskew_zyx(t, row_sums=rs, **kw)
    assert float((a - b).abs().max()) <= 1e-6 * float(a.abs().max())
    # data zeros (a dark block): part of the reference's mask -> the kernel raises its flag, the mask pipeline redoes the volume
    vz = vol.copy()
    vz[shape[0] // 3: shape[0] // 3 + 6, : min(20, shape[1]), 8:40] = 0.0
    for fill in ("mean", 123.5):
        want = O.fast_deskew_zyx(vz, angle, ratio, True, n, fill)
        got = fast_deskew_zyx(torch.from_numpy(vz).to(gpu), angle, ratio, True, n, fill)
        path = deskew_fill_path(gpu)
        zeros_in_signal = bool(((O.fast_deskew_zyx(vz, angle, ratio, True, n, 0) == 0) & (O.fast_deskew_zyx(vol, angle, ratio, True, n, 0) != 0)).any())
        assert path == (2 if zeros_in_signal else 1), (shape, fill, path)
        assert rel_err(got.cpu().numpy(), want) <= DESKEW_TOL, (shape, fill)
    # and a clean volume afterwards takes the one-pass path again (the flag is re-armed per call)
    got = fast_deskew_zyx(t, **kw)
    assert deskew_fill_path(gpu) == 1


def test_richardson_lucy_hands_row_sums_to_deskew(gpu):
    """bh_richardson_lucy_apply_rows: the last update pass of the fused engine (rows of 512 / 1024 / 2048 voxels) also reduces the
    row sums of the estimate it stores; bh_deskew_rows derives the mean fill from them.  Other back-ends report "not produced"."""
    from biahub_amd.deconvolve import PreparedRichardsonLucy
    from biahub_amd.deskew import deskew_fill_path, fast_deskew_zyx

    psf = torch.from_numpy(O.gaussian_psf((9, 5, 5), (1.5, 1.0, 1.0))).to(gpu)
    for shape, expect in (((32, 64, 512), True), ((16, 32, 1024), True), ((32, 64, 128), False), ((20, 36, 100), False)):
        vol = torch.from_numpy(O.synthetic_volume(shape, seed=sum(shape), n_blobs=12)).to(gpu)
        rs = torch.full(shape[:2], -1.0, dtype=torch.float64, device=gpu)
        with PreparedRichardsonLucy(psf, shape, gpu) as prep:
            plain = prep(vol, 3, 1e-6)
            est, got = prep(vol, 3, 1e-6, row_sums=rs)
        assert torch.equal(plain, est)
        assert (got is not None) == expect, (shape, prep.backend)
        if got is None:
            assert bool((rs == -1.0).all())
            continue
        want = est.to(torch.float64).sum(dim=2)
        assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max()), shape
        if shape[2] % 64 == 0:
            kw = dict(ls_angle_deg=36.17, px_to_scan_ratio=0.371, keep_overhang=True, average_n_slices=3, overhang_fill="mean")
            a, b = fast_deskew_zyx(est, **kw), fast_deskew_zyx(est, row_sums=got, **kw)
            assert deskew_fill_path(gpu) == 1
            assert float((a - b).abs().max()) <= 1e-6 * float(a.abs().max())


def test_host_deskew_equals_gpu_deskew(gpu):
    """libbhcore's host implementation of the operator (bh_host_deskew, what `device: cpu` runs without a GPU) restates the
    kernels' float32 arithmetic operation by operation: bit-identical without a fill, the fill value to float32 rounding."""
    from biahub_amd.deskew import _fast_deskew_czyx

    rng = np.random.default_rng(17)
    for shape, n, dtype in (((40, 50, 64), 3, np.float32), ((33, 21, 70), 2, np.uint16), ((25, 16, 128), 1, np.float32)):
        vol = (rng.random(shape) * 900 + 20).astype(dtype)
        vol[5:9, 3:8, 10:30] = 0
        for fill in (0, "mean"):
            kw = dict(ls_angle_deg=36.17, px_to_scan_ratio=0.371, keep_overhang=True, average_n_slices=n, overhang_fill=fill)
            host = _fast_deskew_czyx(vol[None], device="cpu", **kw)
            dev = _fast_deskew_czyx(vol[None], device="cuda", **kw)
            if fill == 0:
                assert np.array_equal(host, dev), (shape, n)
            else:
                assert rel_err(host, dev) <= 1e-6, (shape, n)


def test_deskew_errors(gpu):
    from biahub_amd.deskew import fast_deskew_zyx

    x = torch.zeros((10, 500, 100), device=gpu)
    with pytest.raises(ValueError, match="Dataset contains only overhang"):
        fast_deskew_zyx(x, 30, 0.1, keep_overhang=False)
    with pytest.raises(ValueError):
        fast_deskew_zyx(torch.zeros((4, 4), device=gpu), 30, 0.3, True)
    with pytest.raises(ValueError):
        fast_deskew_zyx(x, 30, 0.1, True, 1, "median")


def test_fill_overhang_standalone(gpu):
    from biahub_amd.deskew import fill_overhang

    rng = np.random.default_rng(3)
    vol = rng.random((20, 37, 150), dtype=np.float32) + 0.5
    vol[:, :, :40] = 0
    vol[5, 5, 100] = 0
    vol[19, 36, 149] = 0
    for it in (0, 1, 3):
        want = O.fill_overhang(vol, None, it)
        got = fill_overhang(torch.from_numpy(vol).to(gpu), None, it).cpu().numpy()
        assert rel_err(got, want) <= 1e-6, it
    want = O.fill_overhang(vol, 7.5, 3)
    got = fill_overhang(torch.from_numpy(vol).to(gpu), 7.5, 3).cpu().numpy()
    assert np.array_equal(got, want)
    allzero = fill_overhang(torch.zeros((4, 5, 6), device=gpu), None).cpu().numpy()
    assert np.isnan(allzero).all()  # mean of an empty selection is NaN in the reference too


# ----------------------------------------------------------------------------- deconvolution
def test_transfer_function_golden(gpu):
    from biahub_amd.deconvolve import compute_tranfser_function

    z = np.load(GOLDEN / "transfer_function.npz")
    for j in range(4):
        got = compute_tranfser_function(z[f"psf{j}"], tuple(int(v) for v in z[f"shape{j}"]))
        assert got.shape == z[f"tf{j}"].shape and got.dtype == np.float32
        assert rel_err(got, z[f"tf{j}"]) <= 1e-5, j


@pytest.mark.parametrize("shape", [(16, 20, 24), (15, 21, 25), (32, 64, 48)])
def test_tikhonov_vs_oracle(gpu, shape):
    from biahub_amd.deconvolve import compute_tranfser_function, deconvolve

    rng = np.random.default_rng(7)
    psf = O.gaussian_psf((7, 5, 5), (1.5, 1.0, 1.0))
    czyx = rng.random((2,) + shape, dtype=np.float32) * 100
    tf = compute_tranfser_function(psf, shape)
    assert rel_err(tf, O.compute_transfer_function(psf, shape)) <= 1e-5
    got = deconvolve(czyx, transfer_function=tf, regularization_strength=1e-3)
    want = O.deconvolve_czyx(czyx, tf, 1e-3)
    assert got.shape == want.shape and got.dtype == np.float32
    assert rel_err(got, want) <= FFT_TOL
    # the staged filter is kept per transfer-function object: a second call reuses it, an in-place edit or another
    # regularisation must not
    from biahub_amd import deconvolve as D

    n0 = len(D._PREPARED)
    assert rel_err(deconvolve(czyx[:1], transfer_function=tf, regularization_strength=1e-3), want[:1]) <= FFT_TOL
    assert len(D._PREPARED) == n0
    assert rel_err(deconvolve(czyx[:1], transfer_function=tf, regularization_strength=1e-2),
                   O.deconvolve_czyx(czyx[:1], tf, 1e-2)) <= FFT_TOL
    tf *= 0.5
    assert rel_err(deconvolve(czyx[:1], transfer_function=tf, regularization_strength=1e-3),
                   O.deconvolve_czyx(czyx[:1], tf, 1e-3)) <= FFT_TOL
    assert len(D._PREPARED) <= D._PREPARED_MAX
    with pytest.raises(ValueError):
        deconvolve(czyx[:, 1:], transfer_function=tf)


@pytest.mark.parametrize("shape,pshape", [((16, 20, 24), (5, 5, 5)), ((15, 21, 25), (7, 5, 3)), ((32, 48, 64), (9, 7, 7))])
def test_richardson_lucy_vs_oracle(gpu, shape, pshape):
    from biahub_amd.deconvolve import richardson_lucy, richardson_lucy_czyx

    vol = O.synthetic_volume(shape, seed=3, n_blobs=6)
    psf = O.gaussian_psf(pshape, tuple(max(p / 5.0, 0.8) for p in pshape))
    want = O.richardson_lucy_zyx(vol, psf, iterations=10, eps=1e-6)
    got = richardson_lucy(torch.from_numpy(vol).to(gpu), torch.from_numpy(psf).to(gpu), 10, 1e-6).cpu().numpy()
    assert rel_err(got, want) <= FFT_TOL
    assert got.min() >= 0
    assert abs(got.sum(dtype=np.float64) - vol.sum(dtype=np.float64)) / vol.sum(dtype=np.float64) < 1e-4  # flux kept
    got0 = richardson_lucy(torch.from_numpy(vol).to(gpu), torch.from_numpy(psf).to(gpu), 0).cpu().numpy()
    assert np.array_equal(got0, np.maximum(vol, 0))
    c = richardson_lucy_czyx(np.stack([vol, vol]), psf, iterations=3)
    assert c.shape == (2,) + shape and np.array_equal(c[0], c[1])


@pytest.mark.parametrize("shape,pshape", [
    ((37, 53, 71), (7, 5, 9)),      # every axis has a large prime factor: all three padded and folded
    ((37, 53, 71), (4, 6, 8)),      # even PSF extents: the kernel reaches further below 0 than above N-1
    ((32, 53, 64), (9, 7, 5)),      # only Y is awkward; Z and X wrap on their own
    ((19, 40, 134), (5, 3, 11)),    # 134 = 2 * 67
    ((21, 64, 150), (7, 5, 9)),     # engine box: Y a power of two kept as it is, Z -> 32, X -> 256
    ((40, 70, 64), (9, 9, 3)),      # X stays, Z -> 48 = 3 * 16 (radix-3 columns), Y -> 96 (columns of 48 = 3 * 16)
    ((80, 150, 64), (5, 5, 5)),     # Z -> 96 (odd log2 of the power-of-two part), Y -> 192
    ((170, 64, 64), (9, 3, 3)),     # Z -> 192 = 3 * 64
    ((6, 40, 2300), (3, 3, 9)),     # X -> 3072: rows beyond 2048 voxels run the 8-row X passes
    ((70, 150, 300), (5, 5, 5)),    # -> (80, 160, 320): radix-5 first steps on all three axes
])
def test_richardson_lucy_awkward_sizes_pad_fold(gpu, shape, pshape, monkeypatch):
    """Axes with a large prime factor are zero-padded to a 7-smooth FFT size and the wrapped part of the linear
    convolution folded back: the result is still the circular R-L at the true size (oracle), and equals the plain
    hipFFT path at the awkward size (BH_RL_NOPAD=1)."""
    from biahub_amd.deconvolve import richardson_lucy

    vol = O.synthetic_volume(shape, seed=9, n_blobs=8)
    vol[0, 0, :] += 500.0   # structure on the faces: a wrong fold shows up as a wrap-around error
    vol[-1, :, -1] += 300.0
    psf = O.gaussian_psf(pshape, tuple(max(p / 4.0, 0.8) for p in pshape))
    psf[0, 0, 0] += 0.02    # asymmetric: convolution and correlation differ
    want = O.richardson_lucy_zyx(vol, psf, iterations=6, eps=1e-6)
    v, pt = torch.from_numpy(vol).to(gpu), torch.from_numpy(psf).to(gpu)
    got = richardson_lucy(v, pt, 6, 1e-6).cpu().numpy()          # whichever back-end the cost model picks
    assert rel_err(got, want) <= FFT_TOL, rel_err(got, want)
    monkeypatch.setenv("BH_RL_ENGINE_PAD", "0")                   # library transforms at the 7-smooth box
    lib = richardson_lucy(v, pt, 6, 1e-6).cpu().numpy()
    assert rel_err(lib, want) <= FFT_TOL, rel_err(lib, want)
    monkeypatch.setenv("BH_RL_NOPAD", "1")
    plain = richardson_lucy(v, pt, 6, 1e-6).cpu().numpy()
    assert rel_err(plain, want) <= FFT_TOL and rel_err(lib, plain) <= FFT_TOL
    monkeypatch.delenv("BH_RL_NOPAD")
    monkeypatch.setenv("BH_RL_ENGINE_PAD", "1")                   # fused engine at the wrap-padded power-of-two box
    eng = richardson_lucy(v, pt, 6, 1e-6).cpu().numpy()
    assert rel_err(eng, want) <= FFT_TOL, rel_err(eng, want)
    assert np.array_equal(richardson_lucy(v, pt, 0, 1e-6).cpu().numpy(), np.maximum(vol, 0))


@pytest.mark.parametrize("shape,pshape", [((48, 96, 64), (5, 7, 3)), ((24, 32, 128), (3, 3, 9)), ((64, 192, 64), (9, 5, 5)),
                                          ((96, 64, 256), (7, 3, 3)),
                                          ((16, 32, 192), (5, 5, 9)), ((8, 32, 384), (3, 5, 7)), ((24, 96, 768), (5, 3, 11)),
                                          ((4, 32, 1536), (1, 3, 17)),    # the last four: rows of 3 * 2^k
                                          ((4, 32, 3072), (1, 3, 9)),     # 3072-voxel rows: the 8-row instantiation of the X passes
                                          ((40, 160, 64), (5, 5, 3)), ((16, 32, 320), (3, 3, 5)), ((80, 32, 640), (3, 3, 7)),
                                          ((8, 32, 2560), (3, 3, 3))])    # radix-5 first steps: columns, rows, 8-row rows
def test_richardson_lucy_radix3_columns_fused(gpu, shape, pshape, monkeypatch):
    """Axes of 3 * 2^k: the fused 8-pass iteration runs at the volume's own shape, the transforms of those axes starting with a
    radix-3 step (csrc/fftconv.hip radix3_step; for rows the real-transform untangle pairs thirds 1 and 2 with each other);
    oracle parity and agreement with the library-FFT path."""
    from biahub_amd.deconvolve import richardson_lucy, richardson_lucy_plan

    assert richardson_lucy_plan(pshape, shape) == (shape, "engine")
    vol = O.synthetic_volume(shape, seed=12, n_blobs=10)
    vol[0, :, 0] += 400.0
    psf = O.gaussian_psf(pshape, tuple(max(p / 4.0, 0.8) for p in pshape))
    psf[0, 0, 0] += 0.02
    want = O.richardson_lucy_zyx(vol, psf, iterations=5, eps=1e-6)
    v, pt = torch.from_numpy(vol).to(gpu), torch.from_numpy(psf).to(gpu)
    got = richardson_lucy(v, pt, 5, 1e-6).cpu().numpy()
    assert rel_err(got, want) <= FFT_TOL, rel_err(got, want)
    monkeypatch.setenv("BH_FFT_BACKEND", "hipfft")
    assert richardson_lucy_plan(pshape, shape) == (shape, "library")
    lib = richardson_lucy(v, pt, 5, 1e-6).cpu().numpy()
    assert rel_err(got, lib) <= FFT_TOL


def test_legacy_fill_overhang_with_mean(gpu):
    """deskew._fill_overhang_with_mean (legacy path: exact zeros grown by SciPy's 6-connected structuring element) against
    the reference's outputs: the same voxels are filled (bit-exact mask), the mean agrees to float32 rounding."""
    from biahub_amd.deskew import _fill_overhang_with_mean, fill_overhang

    z = np.load(GOLDEN / "legacy_fill.npz")
    for j in range(5):
        vol, it, want = z[f"in{j}"], int(z[f"it{j}"]), z[f"out{j}"]
        got = _fill_overhang_with_mean(vol, dilation_iterations=it)
        changed_w, changed_g = want != vol, got != vol
        assert np.array_equal(changed_w, changed_g), j                      # same dilated mask
        assert np.array_equal(got[~changed_g], vol[~changed_g])
        fill_w, fill_g = want[changed_w], got[changed_g]
        assert np.all(fill_g == fill_g[0]) and abs(float(fill_g[0]) - float(fill_w[0])) <= 2e-6 * abs(float(fill_w[0])), j
        # the production structuring element (26-connected) masks more: the two are different on purpose
        prod = fill_overhang(torch.from_numpy(vol).to(gpu), None, it).cpu().numpy()
        assert (prod != vol).sum() >= changed_g.sum()
    with pytest.raises(ValueError):
        _fill_overhang_with_mean(z["in0"], dilation_iterations=0)
    # the legacy entry point composes the production resampler with the legacy fill (biahub/deskew.py:445-450)
    from biahub_amd.deskew import deskew_zyx

    raw = O.synthetic_volume((20, 14, 9), seed=3, n_blobs=4) + 50
    zero = deskew_zyx(raw, 36.17, 0.371, True, average_n_slices=2, overhang_fill="zero")
    mean = deskew_zyx(raw, 36.17, 0.371, True, average_n_slices=2, overhang_fill="mean")
    assert np.array_equal(zero, O.fast_deskew_zyx(raw, 36.17, 0.371, True, 2, 0).astype(np.float32)) or rel_err(zero, O.fast_deskew_zyx(raw, 36.17, 0.371, True, 2, 0)) <= 1e-5
    assert rel_err(mean, O.fill_overhang_with_mean(zero, 3)) <= 1e-6


def test_valid_mask_kernels_bit_exact(gpu):
    """bh_valid_mask / bh_bits_and / bh_bits_unpack against NumPy: (v != 0) & ~isnan(v), popcount, AND."""
    from biahub_amd.estimate_crop import _unpack, valid_mask_device
    from biahub_amd.device import get_context, ptr

    rng = np.random.default_rng(8)
    ctx = get_context(gpu)
    for dt, n in ((np.float32, 100_003), (np.uint16, 64 * 500), (np.uint8, 77), (np.int16, 4099), (np.float64, 1000)):
        a = (rng.random(n) * 10 - 3).astype(dt)
        a[rng.random(n) < 0.3] = 0
        if np.dtype(dt).kind == "f":
            a[rng.random(n) < 0.1] = np.nan
            a[5] = -0.0
            if dt == np.float64:
                a[6] = 1e-60  # not zero, though a float32 cast would make it one
        want = (a != 0) & ~np.isnan(a) if np.dtype(dt).kind == "f" else a != 0
        bits, count = valid_mask_device(a, gpu)
        assert count == int(want.sum()), dt
        assert np.array_equal(_unpack(bits, (n,), gpu).cpu().numpy().astype(bool), want), dt
        other = rng.random(n) < 0.5
        obits, _ = valid_mask_device(other.astype(np.uint8), gpu)
        _lib_and = ctx.lib.bh_bits_and(ctx.handle, ptr(bits), ptr(obits), bits.numel())
        assert _lib_and == 0
        assert np.array_equal(_unpack(bits, (n,), gpu).cpu().numpy().astype(bool), want & other), dt


def test_estimate_crop_golden_and_cli(gpu, tmp_path):
    """estimate_crop_one_position on stores holding the fixture's arrays gives the reference's crops; the command merges
    positions (largest start, smallest stop) into the concatenate configuration it writes."""
    import yaml
    from click.testing import CliRunner

    from biahub_amd import io
    from biahub_amd.cli import cli
    from biahub_amd.estimate_crop import estimate_crop_one_position, standardize_ranges

    z = np.load(GOLDEN / "estimate_crop.npz")
    crops = []
    for j in range(4):
        radius = None if np.isnan(z[f"radius{j}"]) else float(z[f"radius{j}"])
        for name, arr in (("lf", z[f"lf{j}"]), ("ls", z[f"ls{j}"])):
            dt = np.float32 if arr.dtype.kind == "f" else arr.dtype
            io.create_empty_plate(tmp_path / f"{name}{j}.zarr", [("A", "1", "0")], [f"c{k}" for k in range(arr.shape[1])], arr.shape,
                                  dtype=dt, compressor="blosc")
            pos = io.open_ome_zarr(tmp_path / f"{name}{j}.zarr/A/1/0")
            for t in range(arr.shape[0]):
                for c in range(arr.shape[1]):
                    pos.data[t, c] = arr[t, c]
        got = estimate_crop_one_position(tmp_path / f"lf{j}.zarr/A/1/0", tmp_path / f"ls{j}.zarr/A/1/0", lf_mask_radius=radius)
        assert np.array_equal(np.array(got), z[f"crop{j}"]), (j, got)
        crops.append(got)
    assert standardize_ranges([crops[1], crops[2]]).tolist() == [[1, 8, 7], [8, 33, 32]]
    # the command: two positions per dataset (cases 1 and 2 share their arrays, case 2 has no radius -> run with one radius)
    for name in ("lf", "ls"):
        arr = z[f"{name}1"]
        io.create_empty_plate(tmp_path / f"{name}.zarr", [("A", "1", "0"), ("B", "1", "0")], [f"c{k}" for k in range(arr.shape[1])],
                              arr.shape, dtype=np.float32 if arr.dtype.kind == "f" else arr.dtype)
        for key, shift in (("A/1/0", 0), ("B/1/0", 3)):
            pos = io.open_ome_zarr(tmp_path / f"{name}.zarr" / key)
            for t in range(arr.shape[0]):
                for c in range(arr.shape[1]):
                    v = arr[t, c].copy()
                    if name == "lf" and shift:
                        v[:2, :, :] = 0  # the second position's phase volume starts two planes later
                    pos.data[t, c] = v
    cfg = tmp_path / "concat.yml"
    cfg.write_text(yaml.safe_dump({"concat_data_paths": ["lf.zarr/*/*/*", "ls.zarr/*/*/*"], "channel_names": ["all", "all"]}))
    out = tmp_path / "out" / "concat_cropped.yml"
    out.parent.mkdir()
    r = CliRunner().invoke(cli, ["estimate-crop", "-c", str(cfg), "-o", str(out), "--lf-mask-radius", "0.9"])
    assert r.exit_code == 0, (r.output, r.exception)
    res = yaml.safe_load(out.read_text())
    a = O.estimate_crop_arrays(z["lf1"], z["ls1"], 0.9, __import__("biahub_amd.register", fromlist=["find_lir"]).find_lir)
    lf_b = z["lf1"].copy()
    lf_b[:, :, :2] = 0
    b = O.estimate_crop_arrays(lf_b, z["ls1"], 0.9, __import__("biahub_amd.register", fromlist=["find_lir"]).find_lir)
    assert b[0][0] == 2 and a[0][0] == 1
    want = standardize_ranges([a, b])
    assert (res["Z_slice"], res["Y_slice"], res["X_slice"]) == (want[:, 0].tolist(), want[:, 1].tolist(), want[:, 2].tolist())
    assert res["concat_data_paths"] == ["lf.zarr/*/*/*", "ls.zarr/*/*/*"] and (out.parent / "crop_slices.csv").exists()
    assert not (out.parent / "crop_estimates").exists()


# ----------------------------------------------------------------------------- affine
def test_affine_reference_tests(gpu):
    """tests/test_affine.py:26-59 of the reference, verbatim expectations."""
    from biahub_amd.register import apply_affine_transform

    ones = np.ones((10, 10, 10))
    for interp in ("linear", "nearestneighbor"):
        r = apply_affine_transform(ones, np.eye(4), (10, 10, 10), interpolation=interp)
        assert isinstance(r, np.ndarray) and r.shape == (10, 10, 10)
        assert np.all(r == 1)
    m = np.eye(4)
    m[:3, -1] = np.array([-3, 1, 4])
    r = apply_affine_transform(ones, m, (10, 10, 10))
    assert r.shape == (10, 10, 10)
    assert np.all(r[3:10, 0:9, 0:6] == 1)
    assert np.all(r[0:3] == 0)
    with pytest.raises(ValueError, match="Unknown method"):
        apply_affine_transform(ones, m, (10, 10, 10), method="cupy")


def _similarity(angle_deg, scale, t):
    th = np.deg2rad(angle_deg)
    return np.array([[scale, 0, 0, t[0]], [0, scale * np.cos(th), -scale * np.sin(th), t[1]],
                     [0, scale * np.sin(th), scale * np.cos(th), t[2]], [0, 0, 0, 1.0]])


@pytest.mark.parametrize("interp", ["linear", "nearestneighbor"])
@pytest.mark.parametrize("M", [
    _similarity(2.0, 1.02, (3.5, -12.25, 20.75)),   # BASELINE config 3's ground-truth transform
    _similarity(30.0, 0.7, (1.0, 20.0, -5.0)),
    _similarity(90.0, 1.0, (0.0, 0.0, 39.0)),        # LDS box does not fit -> global gather path
    np.array([[0.5, 0.1, 0.0, 2.0], [0.0, 1.5, 0.2, -3.0], [0.1, 0.0, 2.0, 1.0], [0, 0, 0, 1.0]]),
])
def test_affine_itk_mode_vs_oracle(gpu, M, interp):
    from biahub_amd.register import apply_affine_transform

    rng = np.random.default_rng(9)
    vol = rng.random((24, 40, 72), dtype=np.float32) * 1000
    vol[3, 4, 5] = np.nan
    out_shape = (20, 44, 80)
    want = O.apply_affine_transform(vol, M, out_shape, interp)
    got = apply_affine_transform(vol, M, out_shape, interpolation=interp)
    assert got.shape == want.shape and got.dtype == np.float32
    if interp == "linear":
        assert rel_err(got, want) <= 1e-5
    else:
        assert np.array_equal(got, want)
    crop = (slice(2, 18), slice(5, 40), slice(7, 70))
    got_c = apply_affine_transform(vol, M, out_shape, interpolation=interp, crop_output_slicing=crop)
    assert np.array_equal(got_c, got[crop])  # the cropped launch equals slicing the full warp
    got4 = apply_affine_transform(np.stack([vol, vol * 2]), M, out_shape, interpolation=interp)
    assert got4.shape == (2,) + out_shape and np.array_equal(got4[0], got)


def _rotation_about(axis, angle_deg, centre):
    ax = np.asarray(axis, float) / np.linalg.norm(axis)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    th = np.deg2rad(angle_deg)
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    M = np.eye(4)
    M[:3, :3] = R
    M[:3, 3] = np.asarray(centre) - R @ np.asarray(centre)
    return M


@pytest.mark.parametrize("interp", ["linear", "nearestneighbor"])
@pytest.mark.parametrize("axis,angle", [((0, 1.0, 0), 20.0), ((0, 1.0, 0), 45.0), ((1.0, 0.4, 0.3), 30.0), ((1.0, 0.4, 0.3), 90.0),
                                        ((0, 0.2, 1.0), 60.0)])
def test_affine_strongly_coupled_rotations(gpu, axis, angle, interp, monkeypatch):
    """Rotations that mix z with x / y by more than a few degrees: a full output tile's source box does not fit LDS and the
    warp runs on compact 8 x 8 x 16 blocks whose source box is staged in LDS (affine_gather_kernel).  Against the oracle, against
    the tile kernel's own gather fallback (BH_AFFINE_GATHER=0), for float32 and uint16 input, with a cropped launch."""
    from biahub_amd.register import apply_affine_transform

    rng = np.random.default_rng(17)
    shape = (40, 72, 136)
    vol = (rng.random(shape, dtype=np.float32) * 1000).round()
    M = _rotation_about(axis, angle, [(n - 1) / 2 for n in shape])
    out_shape = (44, 70, 150)
    want = O.apply_affine_transform(vol, M, out_shape, interp)
    got = apply_affine_transform(vol, M, out_shape, interpolation=interp)
    if interp == "linear":
        assert rel_err(got, want) <= 1e-5
    else:
        assert np.array_equal(got, want)
    monkeypatch.setenv("BH_AFFINE_GATHER", "0")
    tile = apply_affine_transform(vol, M, out_shape, interpolation=interp)
    monkeypatch.delenv("BH_AFFINE_GATHER")
    assert np.array_equal(got, tile)  # same arithmetic voxel for voxel: which kernel ran does not show
    got16 = apply_affine_transform(vol.astype(np.uint16), M, out_shape, interpolation=interp)
    assert np.array_equal(got16, got)
    crop = (slice(3, 40), slice(5, 66), slice(9, 140))
    assert np.array_equal(apply_affine_transform(vol, M, out_shape, interpolation=interp, crop_output_slicing=crop), got[crop])


@pytest.mark.parametrize("X", [200, 198])  # 16-B LDS-DMA staging / dword staging (rows not 16-B aligned)
def test_affine_interior_tiles_nonfinite(gpu, X):
    """Volume large enough that most tiles take the interior (branch-free) loop; NaN / +-inf taps inside them must
    come out as np.nan_to_num'd values (register.py:254), whichever tile samples them."""
    from biahub_amd.register import apply_affine_transform

    rng = np.random.default_rng(21)
    vol = rng.random((40, 48, X), dtype=np.float32) * 100
    vol[20, 24, 100] = np.nan
    vol[21, 30, 64] = np.inf      # on a tile seam in x
    vol[16, 16, 128] = -np.inf    # on tile seams in z, y and x
    M = _similarity(2.0, 1.02, (0.5, -1.25, 2.75))
    for interp in ("linear", "nearestneighbor"):
        want = O.apply_affine_transform(vol, M, vol.shape, interp)
        got = apply_affine_transform(vol, M, vol.shape, interpolation=interp)
        assert np.isfinite(got).all()
        if interp == "linear":  # per element: the +-FLT_MAX taps would swamp a max-norm error
            assert np.allclose(got, want, rtol=2e-5, atol=2e-3)
        else:
            assert np.array_equal(got, want)
    u16 = rng.integers(0, 60000, vol.shape).astype(np.uint16)   # X = 200: 16-B group staging; X = 198: per-sample loads
    assert rel_err(apply_affine_transform(u16, M, u16.shape), O.apply_affine_transform(u16, M, u16.shape, "linear")) <= 1e-5
    i16 = (rng.integers(0, 60000, vol.shape) - 30000).astype(np.int16)
    assert rel_err(apply_affine_transform(i16, M, i16.shape), O.apply_affine_transform(i16, M, i16.shape, "linear")) <= 1e-5
    assert np.array_equal(apply_affine_transform(u16, np.eye(4), u16.shape), u16.astype(np.float32))


def test_affine_z_walk_equals_staged_tiles(gpu, monkeypatch):
    """Z-separable warps (linear and nearest) take the wave-private z walk (csrc/affine_zwalk.inc); every voxel must be BIT-identical to
    the staged-tile kernel's (BH_AFFINE_NOZWALK=1), which the oracle / golden tests above pin: interior and every volume
    face, both edge rules, crops, ragged extents, every input dtype, z scales from 0 to > 2, NaN / inf taps."""
    from biahub_amd import _lib
    from biahub_amd.register import affine_device

    rng = np.random.default_rng(77)
    vol = rng.random((37, 50, 203), dtype=np.float32) * 1000 - 200
    vol[20, 24, 100] = np.nan
    vol[21, 30, 64] = np.inf
    vol[5, 7, 128] = -np.inf
    vol[0, 0, 0] = np.inf       # a clamped corner tap
    vol[36, 49, 202] = np.nan

    def zsep(az, ang, s, t):
        m = _similarity(ang, s, t)
        m[0, 0] = az
        return m

    mats = [np.eye(4), zsep(1.0, 0, 1, (2.5, -7.5, 11.125)), zsep(1.02, 2.0, 1.02, (3.5, -12.25, 20.75)),
            zsep(0.4, 30.0, 0.7, (1.0, 20.0, -5.0)), zsep(2.6, -5.0, 1.3, (-4.0, 3.0, 6.5)), zsep(0.0, 0.0, 1.0, (7.25, 0.5, -0.5)),
            zsep(1.0, 90.0, 1.0, (0.0, 0.0, 49.0)), zsep(1.0, 0, 1, (-0.5, -0.5, -0.5)), zsep(1.0, 0, 1, (0.5, 0.5, 0.5)),
            zsep(1.0, 180.0, 1.0, (0.0, 49.0, 202.0))]
    vol4 = rng.random((21, 70, 264), dtype=np.float32) * 1000 - 200   # 16-B aligned rows: interior waves take the LDS-DMA ring
    vol4[10, 35, 130] = np.nan
    vol4[11, 36, 131] = -np.inf
    cases = [(vol, _lib.DT_F32), (vol4, None)]
    cases.append((rng.integers(0, 60000, vol.shape).astype(np.uint16), None))
    cases.append(((rng.integers(0, 60000, vol.shape) - 30000).astype(np.int16), None))
    cases.append((rng.integers(0, 255, vol.shape).astype(np.uint8), None))
    cases.append((rng.integers(0, 60000, vol4.shape).astype(np.uint16), None))             # 16-B aligned 16-bit rows: LDS-DMA ring
    cases.append(((rng.integers(0, 60000, vol4.shape) - 30000).astype(np.int16), None))
    n_walk = 0
    for src, _ in cases:
        t = torch.from_numpy(src).to(gpu)
        for M in mats:
            for boundary in (_lib.BOUNDARY_ITK, _lib.BOUNDARY_SCIPY_CONSTANT):
                for shape, lo, cs in (((37, 50, 203), (0, 0, 0), None), ((41, 57, 130), (3, 5, 66), (30, 46, 63))):
                    for interp in ("linear", "nearestneighbor"):
                        monkeypatch.delenv("BH_AFFINE_NOZWALK", raising=False)
                        got = affine_device(t, M, shape, interp, boundary, -3.5, lo, cs)
                        monkeypatch.setenv("BH_AFFINE_NOZWALK", "1")
                        want = affine_device(t, M, shape, interp, boundary, -3.5, lo, cs)
                        assert torch.isfinite(got).all()
                        assert torch.equal(got, want), (src.dtype, M.tolist(), boundary, shape, interp)
                        n_walk += 1
    monkeypatch.delenv("BH_AFFINE_NOZWALK", raising=False)
    assert n_walk == 7 * len(mats) * 4 * 2
    # and against the oracle directly on the float volume (the walk is what the registration / stabilisation calls hit)
    from biahub_amd.register import apply_affine_transform

    for M in mats[:5]:
        assert np.allclose(apply_affine_transform(vol, M, vol.shape), O.apply_affine_transform(vol, M, vol.shape, "linear"),
                           rtol=2e-5, atol=2e-3)


def test_affine_oblique_walk_equals_staged_tiles(gpu, monkeypatch):
    """Linear warps with a weak z coupling (a small out-of-plane rotation) take the z walk with per-lane source planes
    (csrc/affine_zoblique.inc: a ring of four LDS plane slots per wave); stronger couplings stay on the staged tiles.  Every
    voxel must be BIT-identical to the staged-tile kernel's (BH_AFFINE_NOZWALK=1): interior planes, planes whose window leaves
    the volume, waves whose box does, crops, both edge rules, NaN / inf taps."""
    from biahub_amd import _lib
    from biahub_amd.register import affine_device

    rng = np.random.default_rng(78)
    vol = rng.random((70, 90, 264), dtype=np.float32) * 1000 - 200
    vol[30, 44, 100] = np.nan
    vol[31, 50, 64] = np.inf
    vol[5, 7, 128] = -np.inf
    vol[0, 0, 0] = np.inf
    vol[69, 89, 263] = np.nan

    def rot(axis, deg, s, t):
        ax = np.asarray(axis, dtype=np.float64)
        ax /= np.linalg.norm(ax)
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        th = np.deg2rad(deg)
        m = np.eye(4)
        m[:3, :3] = s * (np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K)
        m[:3, 3] = t
        return m

    mats = [rot((1.0, 0.4, 0.3), 2.0, 1.02, (3.5, -12.25, 20.75)), rot((1.0, 0.2, -0.1), -3.0, 0.97, (-2.0, 4.5, 1.25)),
            rot((1.0, 0.0, 0.05), 5.0, 1.0, (0.5, 0.5, 0.5)), rot((0.3, 1.0, 0.0), 0.5, 1.1, (6.0, -3.0, 2.0)),
            rot((0.0, 1.0, 0.0), 2.0, 1.0, (0.0, 0.0, 0.0)),          # z couples with x too strongly: staged tiles
            np.array([[1.0, 0.004, 0.002, 1.5], [0.03, 1.0, 0.0, -2.0], [-0.02, 0.0, 1.0, 3.0], [0, 0, 0, 1.0]])]
    srcs = [vol, rng.integers(0, 60000, vol.shape).astype(np.uint16), (rng.integers(0, 60000, vol.shape) - 30000).astype(np.int16)]
    for src in srcs:  # 16-bit volumes: the same ring with 8 samples per quad
        t = torch.from_numpy(src).to(gpu)
        for M in mats:
            for boundary in (_lib.BOUNDARY_ITK, _lib.BOUNDARY_SCIPY_CONSTANT):
                for shape, lo, cs in (((70, 90, 264), (0, 0, 0), None), ((75, 100, 200), (3, 5, 66), (60, 80, 130))):
                    monkeypatch.delenv("BH_AFFINE_NOZWALK", raising=False)
                    got = affine_device(t, M, shape, "linear", boundary, -3.5, lo, cs)
                    monkeypatch.setenv("BH_AFFINE_NOZWALK", "1")
                    want = affine_device(t, M, shape, "linear", boundary, -3.5, lo, cs)
                    assert torch.isfinite(got).all()
                    assert torch.equal(got, want), (src.dtype, M.tolist(), boundary, shape)
    monkeypatch.delenv("BH_AFFINE_NOZWALK", raising=False)


def test_affine_scipy_mode_golden(gpu):
    from biahub_amd.core.transform import Transform

    z = np.load(GOLDEN / "transform_scipy.npz")
    t = Transform(z["matrix"])
    assert rel_err(t.apply(z["moving"], order=1), z["order1"]) <= 1e-5
    assert np.array_equal(t.apply(z["moving"], order=0), z["order0"])
    ref = np.zeros((10, 20, 18), np.float32)
    assert rel_err(t.apply(z["moving"], reference=ref, order=1, cval=3.0), z["order1_ref"]) <= 1e-5
    assert np.array_equal(Transform.from_translation([-3.0, 1.0, 4.0]).apply(np.ones((10, 10, 10), np.float32)),
                          z["shift_int"])


def test_affine_cubic_spline_golden(gpu):
    """Transform.apply(order=3) and apply_affine_transform(method="scipy") against the reference's own outputs
    (core/transform.py:374-396, register.py:271-272; tests/golden/transform_spline.npz): <= 1e-5 of the volume's maximum."""
    from biahub_amd.core.transform import Transform
    from biahub_amd.register import apply_affine_transform

    z = np.load(GOLDEN / "transform_spline.npz")
    for j in range(3):
        t = Transform(z[f"M{j}"])
        got = t.apply(z[f"mov{j}"], order=3)
        assert got.dtype == np.float32 and got.shape == z[f"o3_{j}"].shape
        assert rel_err(got, z[f"o3_{j}"]) <= 1e-5, j
        assert rel_err(t.apply(z[f"mov{j}"], order=3, cval=37.5), z[f"o3_cval_{j}"]) <= 1e-5, j
    ref = np.zeros((10, 20, 18), np.float32)
    assert rel_err(Transform(z["M0"]).apply(z["mov0"], reference=ref, order=3, cval=-2.0), z["o3_ref_0"]) <= 1e-5
    assert rel_err(Transform(np.eye(4)).apply(z["mov0"], order=3), z["o3_identity"]) <= 1e-5
    assert rel_err(Transform.from_translation([-3.0, 1.0, 4.0]).apply(z["mov0"], order=3), z["o3_shift_int"]) <= 1e-5
    # integer dtypes: SciPy rounds half away from zero and saturates; float32 accumulation here may flip a value that lies
    # within 1e-3 of a half-integer
    for k in ("u16", "i16"):
        for order in (1, 3):
            got = Transform(z["M0"]).apply(z[k], order=order)
            want = z[f"{k}_o{order}"]
            assert got.dtype == want.dtype
            d = np.abs(got.astype(np.int64) - want.astype(np.int64))
            assert d.max() <= 1 and (d != 0).mean() < 2e-3, (k, order, d.max(), (d != 0).mean())
    # 2-D images ride as the single plane of a volume
    assert rel_err(Transform(z["M2d"]).apply(z["img2d"], order=3, cval=5.0), z["img2d_o3"]) <= 1e-5
    assert rel_err(Transform(z["M2d"]).apply(z["img2d"], order=1), z["img2d_o1"]) <= 1e-5
    # the raw register.py:271-272 call: output = input shape whatever output_shape_zyx says; NaN -> 0 first; crop after
    got = apply_affine_transform(z["reg_vol"], z["reg_M"], (12, 20, 24), method="scipy")
    assert got.shape == z["reg_out"].shape and rel_err(got, z["reg_out"]) <= 1e-5
    crop = (slice(1, 9), slice(2, 15), slice(3, 20))
    got = apply_affine_transform(z["reg_vol"], z["reg_M"], (10, 18, 22), method="scipy", crop_output_slicing=crop)
    assert got.shape == z["reg_out_crop"].shape and rel_err(got, z["reg_out_crop"]) <= 1e-5
    got = apply_affine_transform(z["reg_u16"], z["M0"], z["reg_u16"].shape, method="scipy")
    d = np.abs(got.astype(np.int64) - z["reg_u16_out"].astype(np.int64))
    assert got.dtype == np.uint16 and d.max() <= 1 and (d != 0).mean() < 2e-3
    with pytest.raises(ValueError, match="Unknown method"):
        apply_affine_transform(z["reg_vol"], z["reg_M"], (10, 18, 22), method="cupy")


def test_affine_cubic_spline_vs_oracle_sizes(gpu):
    """Prefilter blocks, run-ins and row chunks at sizes beyond the goldens: axes shorter than the run-in, longer than one
    block, rows longer than one LDS chunk, a strongly oblique matrix; the prefilter alone against the oracle's float64 one."""
    from biahub_amd import _lib
    from biahub_amd.device import get_context, ptr
    from biahub_amd.register import affine_device

    rng = np.random.default_rng(11)
    for shape in [(3, 150, 70), (70, 9, 200), (2, 5, 4500), (130, 66, 40)]:
        vol = rng.random(shape, dtype=np.float32) * 100 + 5
        t = torch.from_numpy(vol).to(gpu)
        coef = torch.empty_like(t)
        ctx = get_context(gpu)
        _lib.check(ctx.lib.bh_spline_prefilter(ctx.handle, ptr(t), _lib.DT_F32, *shape, ptr(coef)))
        want = O.spline_prefilter(vol)
        assert rel_err(coef.cpu().numpy(), want) <= 2e-6, shape
    vol = rng.random((40, 50, 60), dtype=np.float32) * 1000
    th = np.deg2rad(25.0)
    M = np.array([[np.cos(th), 0, np.sin(th), 3.3], [0.05, 0.98, 0, -1.2], [-np.sin(th), 0.02, np.cos(th), 9.7], [0, 0, 0, 1.0]])
    got = affine_device(vol, M, (44, 48, 70), "cubic", _lib.BOUNDARY_SCIPY_CONSTANT, 12.5).cpu().numpy()
    want = O.spline_affine_pull(vol, M, (44, 48, 70), 12.5)
    assert rel_err(got, want) <= 1e-5
    lo, shp = (3, 5, 7), (20, 30, 40)
    sub = affine_device(vol, M, (44, 48, 70), "cubic", _lib.BOUNDARY_SCIPY_CONSTANT, 12.5, lo, shp).cpu().numpy()
    assert np.array_equal(sub, got[3:23, 5:35, 7:47])
    with pytest.raises(Exception, match="constant"):
        affine_device(vol, M, vol.shape, "cubic", _lib.BOUNDARY_ITK)


def test_affine_cubic_tile_gather(gpu, monkeypatch):
    """The LDS-tiled 64-tap gather (csrc/spline.hip gather_tile_kernel): interior tiles (no per-voxel tests), face tiles with
    mirrored taps, rows that are not 16-B aligned (dword staging), outputs partly outside the volume, a warp too strong for the
    staging (cache launch) — against the float64 oracle at a size with interior tiles, and bit-identical to the launch that takes
    every tap through the vector cache (the arithmetic is the same, only the source of the taps differs)."""
    from biahub_amd import _lib
    from biahub_amd.register import affine_device

    rng = np.random.default_rng(23)
    vol = rng.random((26, 44, 204), dtype=np.float32) * 500 + 3
    th = np.deg2rad(3.0)
    M = np.array([[1.01, 0.02, 0.0, -0.4], [0.0, np.cos(th), -np.sin(th), 2.6], [0.01, np.sin(th), np.cos(th), -1.3], [0, 0, 0, 1.0]])
    got = affine_device(vol, M, vol.shape, "cubic", _lib.BOUNDARY_SCIPY_CONSTANT, -7.0).cpu().numpy()
    assert rel_err(got, O.spline_affine_pull(vol, M, vol.shape, -7.0)) <= 1e-5
    mats = [M, np.eye(4), _similarity(0, 1, (0.0, 0.0, 0.5)), _similarity(0, 1, (-9.5, 20.25, 130.0)),
            _similarity(40.0, 0.7, (1.0, 2.0, 3.0)), np.diag([1.0, 1.0, -1.0, 1.0]) + np.array([[0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 202.0], [0, 0, 0, 0]])]
    for shape in [(26, 44, 204), (26, 44, 203), (9, 8, 64), (1, 70, 130)]:
        v = torch.from_numpy(rng.random(shape, dtype=np.float32) * 90).to(gpu)
        for j, m in enumerate(mats):
            for oshape in (shape, (shape[0] + 3, shape[1] - 2, shape[2] + 17)):
                monkeypatch.delenv("BH_SPLINE_GATHER", raising=False)
                monkeypatch.delenv("BH_SPLINE_ZUNI", raising=False)
                a = affine_device(v, m, oshape, "cubic", _lib.BOUNDARY_SCIPY_CONSTANT, 1.5)
                monkeypatch.setenv("BH_SPLINE_ZUNI", "0")   # the general 64-tap tile path for matrices that leave z alone too
                c = affine_device(v, m, oshape, "cubic", _lib.BOUNDARY_SCIPY_CONSTANT, 1.5)
                monkeypatch.setenv("BH_SPLINE_GATHER", "global")
                b = affine_device(v, m, oshape, "cubic", _lib.BOUNDARY_SCIPY_CONSTANT, 1.5)
                assert torch.equal(c, b), (shape, j, oshape)
                # z-uniform matrices combine the four source planes first (another order of the same float32 sums)
                assert float((a - b).abs().max()) <= 3e-6 * float(b.abs().max()), (shape, j, oshape)
    monkeypatch.delenv("BH_SPLINE_GATHER", raising=False)
    monkeypatch.delenv("BH_SPLINE_ZUNI", raising=False)
    # the plane-combining path against the float64 oracle: rotation about z + anisotropic scale + shift, volume faces included
    thz = np.deg2rad(7.0)
    Mz = np.array([[1.04, 0.0, 0.0, 1.7], [0.0, 1.05 * np.cos(thz), -np.sin(thz), 3.2], [0.0, np.sin(thz), 0.97 * np.cos(thz), -6.1], [0, 0, 0, 1.0]])
    got = affine_device(vol, Mz, (30, 40, 210), "cubic", _lib.BOUNDARY_SCIPY_CONSTANT, 4.0).cpu().numpy()
    assert rel_err(got, O.spline_affine_pull(vol, Mz, (30, 40, 210), 4.0)) <= 1e-5
    # identity on the integer grid returns the samples (prefilter and B-spline sampling are inverses)
    v = rng.random((20, 30, 140), dtype=np.float32)
    assert rel_err(affine_device(v, np.eye(4), v.shape, "cubic", _lib.BOUNDARY_SCIPY_CONSTANT).cpu().numpy(), v) <= 2e-6


def test_stabilization_transform(gpu):
    from biahub_amd.stabilize import apply_stabilization_transform

    rng = np.random.default_rng(2)
    vol = rng.random((2, 12, 30, 40), dtype=np.float32)
    shifts = [np.eye(4), _similarity(0, 1, (0.5, -2.25, 3.0)), _similarity(1.0, 1, (0, 1, 1))]
    for t in range(3):
        got = apply_stabilization_transform(vol, shifts, t)
        want = np.stack([O.apply_affine_transform(v, shifts[t], v.shape) for v in vol])
        assert got.shape == vol.shape and rel_err(got, want) <= 1e-5
    assert np.array_equal(apply_stabilization_transform(vol, shifts, 0), vol)  # identity is exact
    with pytest.raises(IndexError):
        apply_stabilization_transform(vol, shifts, 7)
    big = apply_stabilization_transform(vol[0], shifts, 1, output_shape=(14, 32, 44))
    assert big.shape == (14, 32, 44)


# ----------------------------------------------------------------------------- crop / flip
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.int16, np.float32, np.float64, np.int32])
def test_crop_flip_bit_exact(gpu, dtype):
    from biahub_amd.array_ops import copy_n_paste, copy_n_paste_czyx, flip_zyx

    rng = np.random.default_rng(4)
    a = (rng.random((3, 9, 17, 33)) * 200).astype(dtype)
    s = [slice(2, 7), slice(0, 17), slice(5, 30)]
    got = copy_n_paste_czyx(a, s)
    assert got.dtype == a.dtype and np.array_equal(got, a[:, 2:7, 0:17, 5:30])
    assert copy_n_paste_czyx(a, [slice(3, 3), slice(0, 5), slice(0, 5)]).shape == (3, 0, 5, 5)  # empty crop
    for fx in (False, True):
        for fy in (False, True):
            assert np.array_equal(flip_zyx(a[0], x=fx, y=fy), O.flip_zyx(a[0], x=fx, y=fy))
    if np.dtype(dtype).kind == "f":
        f = a[0].copy()
        f[1, 2, 3] = np.nan
        f[4, 5, 6] = np.inf
        f[4, 5, 7] = -np.inf
        want = O.copy_n_paste(f, s)
        got = copy_n_paste(f, s)
        assert got.dtype == f.dtype and np.array_equal(got, want)  # nan_to_num incl. its +-inf handling
    else:
        assert np.array_equal(copy_n_paste(a[0], s), a[0][2:7, 0:17, 5:30])


def test_flip_involution_large(gpu):
    from biahub_amd.array_ops import crop_flip_device

    t = torch.randint(0, 65535, (1, 64, 512, 1000), device=gpu, dtype=torch.int32).to(torch.int16)
    once = crop_flip_device(t, (0, 0, 0), t.shape[1:], True, True)
    assert torch.equal(once, t.flip([2, 3]))
    assert torch.equal(crop_flip_device(once, (0, 0, 0), t.shape[1:], True, True), t)


# ----------------------------------------------------------------------------- fused FFT-convolution engine
@pytest.mark.parametrize("shape,pshape", [
    ((16, 32, 64), (5, 5, 5)),       # smallest supported tile shapes
    ((8, 64, 128), (3, 7, 9)),       # log2(Z) odd, log2(Y/2) odd
    ((64, 64, 64), (9, 9, 9)),
    ((32, 128, 256), (9, 7, 7)),
    ((4, 32, 512), (3, 5, 17)),
    ((128, 32, 1024), (17, 5, 9)),
])
def test_richardson_lucy_fused_engine_vs_oracle(gpu, shape, pshape, monkeypatch):
    """Power-of-two volumes take the fused engine (csrc/fftconv.hip); it must agree with the oracle and with
    the hipFFT path (BH_FFT_BACKEND=hipfft) that the other shapes use."""
    from biahub_amd.deconvolve import richardson_lucy

    vol = O.synthetic_volume(shape, seed=5, n_blobs=10)
    psf = O.gaussian_psf(pshape, tuple(max(p / 5.0, 0.7) for p in pshape))
    psf[0, 0, 0] += 0.01  # break the PSF's symmetry so conj(OTF) != OTF is exercised
    want = O.richardson_lucy_zyx(vol, psf, iterations=5, eps=1e-6)
    v, pt = torch.from_numpy(vol).to(gpu), torch.from_numpy(psf).to(gpu)
    got = richardson_lucy(v, pt, 5, 1e-6).cpu().numpy()
    assert rel_err(got, want) <= FFT_TOL, rel_err(got, want)
    assert np.array_equal(richardson_lucy(v, pt, 0, 1e-6).cpu().numpy(), np.maximum(vol, 0))  # zero iterations: e0
    alias = v.clone()
    from biahub_amd.device import get_context, ptr
    from biahub_amd import _lib
    ctx = get_context(gpu)
    _lib.check(ctx.lib.bh_richardson_lucy(ctx.handle, ptr(alias), ptr(pt), *pshape, *shape, 5, 1e-6, ptr(alias)))
    assert np.array_equal(alias.cpu().numpy(), got)                                           # in == out is allowed
    monkeypatch.setenv("BH_FFT_BACKEND", "hipfft")
    got_hipfft = richardson_lucy(v, pt, 5, 1e-6).cpu().numpy()
    assert rel_err(got_hipfft, want) <= FFT_TOL
    assert rel_err(got, got_hipfft) <= FFT_TOL


@pytest.mark.parametrize("shape", [(16, 32, 64), (8, 64, 128), (32, 128, 256), (128, 32, 1024),
                                   (48, 96, 64), (24, 64, 128), (64, 192, 64),     # radix-3 columns
                                   (16, 32, 192), (8, 96, 384), (8, 32, 1536),     # radix-3 rows
                                   (40, 160, 64), (16, 32, 320), (8, 32, 2560)])   # radix-5 columns / rows
def test_tikhonov_fused_engine_vs_oracle(gpu, shape, monkeypatch):
    from biahub_amd.deconvolve import compute_tranfser_function, deconvolve

    rng = np.random.default_rng(21)
    psf = O.gaussian_psf((7, 5, 9), (1.5, 1.0, 2.0))
    czyx = rng.random((2,) + shape, dtype=np.float32) * 100
    tf = compute_tranfser_function(psf, shape)
    want = O.deconvolve_czyx(czyx, tf, 1e-3)
    got = deconvolve(czyx, transfer_function=tf, regularization_strength=1e-3)
    assert rel_err(got, want) <= FFT_TOL, rel_err(got, want)
    monkeypatch.setenv("BH_FFT_BACKEND", "hipfft")
    got2 = deconvolve(czyx, transfer_function=tf, regularization_strength=1e-3)
    assert rel_err(got, got2) <= FFT_TOL


def test_library_fft_plan_kinds_agree(gpu, monkeypatch):
    """csrc/context.hip plans mixed-radix shapes with hipfftPlan3d and all-power-of-two shapes (rocFFT defect)
    as x-rows + strided (z, y) transforms; BH_FFT_SEPARABLE=1 forces the latter: same results either way, also for the
    power-of-two pair that breaks hipfftPlan3d."""
    from biahub_amd.deconvolve import richardson_lucy
    from biahub_amd.device import get_context

    ctx = get_context(gpu)
    psf = O.gaussian_psf((5, 7, 9), (1.0, 1.5, 2.0))
    pt = torch.from_numpy(psf).to(gpu)
    for shape in ((15, 42, 50), (24, 36, 60)):  # 7-smooth: no pad-and-fold, straight to the library plans
        vol = O.synthetic_volume(shape, seed=9, n_blobs=6)
        want = O.richardson_lucy_zyx(vol, psf, iterations=4, eps=1e-6)
        v = torch.from_numpy(vol).to(gpu)
        ctx.release_workspace()
        a = richardson_lucy(v, pt, 4, 1e-6).cpu().numpy()
        ctx.release_workspace()
        monkeypatch.setenv("BH_FFT_SEPARABLE", "1")
        b = richardson_lucy(v, pt, 4, 1e-6).cpu().numpy()
        monkeypatch.delenv("BH_FFT_SEPARABLE")
        ctx.release_workspace()
        assert rel_err(a, want) <= FFT_TOL and rel_err(b, want) <= FFT_TOL, (rel_err(a, want), rel_err(b, want))
    monkeypatch.setenv("BH_FFT_BACKEND", "hipfft")
    for shape in ((8, 128, 64), (4, 32, 256), (64, 8, 64), (2, 64, 256)):  # plans stay alive across the loop
        vol = O.synthetic_volume(shape, seed=10, n_blobs=6)
        ps = O.gaussian_psf((1, 5, 5), (0.5, 1.0, 1.0))
        want = O.richardson_lucy_zyx(vol, ps, iterations=3, eps=1e-6)
        got = richardson_lucy(torch.from_numpy(vol).to(gpu), torch.from_numpy(ps).to(gpu), 3, 1e-6).cpu().numpy()
        assert rel_err(got, want) <= FFT_TOL, (shape, rel_err(got, want))


def test_find_overlapping_volume(gpu):
    from biahub_amd.register import apply_affine_transform, find_overlapping_volume

    m = np.eye(4)
    m[:3, 3] = [-3, 1, 4]  # the reference's integer-shift case: valid region [3:10, 0:9, 0:6]
    assert find_overlapping_volume((10, 10, 10), (10, 10, 10), m) == (slice(3, 10), slice(0, 9), slice(0, 6))
    M = _similarity(5.0, 1.0, (1.5, -3.0, 6.0))
    zs, ys, xs = find_overlapping_volume((24, 40, 72), (20, 44, 80), M)
    warped = apply_affine_transform(np.ones((24, 40, 72), np.float32), M, (20, 44, 80), crop_output_slicing=(zs, ys, xs))
    assert warped.size > 0 and np.all(warped > 0)  # the cuboid lies inside the covered region
    with pytest.raises(ValueError, match="Unknown method"):
        find_overlapping_volume((4, 4, 4), (4, 4, 4), m, method="hull")


def test_richardson_lucy_properties_large(gpu, monkeypatch):
    """Size-independent properties at a size the oracle would not finish quickly (BASELINE config 3/4 volume)."""
    from biahub_amd.deconvolve import richardson_lucy

    shape = (256, 1024, 1024)
    g = torch.Generator(device=gpu).manual_seed(7)
    vol = (torch.rand(shape, generator=g, device=gpu) * 300 + 100).round_()
    psf = torch.from_numpy(O.gaussian_psf((33, 17, 17), (3.0, 1.5, 1.5))).to(gpu)
    est = richardson_lucy(vol, psf, 5, 1e-6)
    assert float(est.min()) >= 0.0 and bool(torch.isfinite(est).all())
    flux_in, flux_out = float(vol.double().sum()), float(est.double().sum())
    assert abs(flux_out - flux_in) / flux_in < 1e-4                      # R-L with a unit-sum PSF conserves flux
    assert float(est.var()) > float(vol.var())                            # and sharpens
    flat = richardson_lucy(torch.full(shape, 42.0, device=gpu), psf, 3, 1e-6)
    assert float((flat - 42.0).abs().max()) < 1e-2                        # a constant image is a fixed point
    monkeypatch.setenv("BH_FFT_BACKEND", "hipfft")                        # the two FFT back-ends agree at full size
    est_ref = richardson_lucy(vol, psf, 5, 1e-6)
    assert float((est - est_ref).abs().max()) <= FFT_TOL * float(est_ref.abs().max())


def test_richardson_lucy_properties_large_awkward(gpu, monkeypatch):
    """The same properties on the deskewed config-3/4 volume (342, 1024, 1517), whose axes fit none of the engine's lengths:
    it runs at the padded box (384, 1024, 1536) with radix-3 first steps, and agrees with the library pad-and-fold path."""
    from biahub_amd.deconvolve import richardson_lucy, richardson_lucy_plan

    shape = (342, 1024, 1517)
    psf_h = O.gaussian_psf((33, 17, 17), (3.0, 1.5, 1.5))
    assert richardson_lucy_plan(psf_h.shape, shape) == ((384, 1024, 1536), "engine-padded")
    g = torch.Generator(device=gpu).manual_seed(8)
    vol = (torch.rand(shape, generator=g, device=gpu) * 300 + 100).round_()
    vol[0] += 500.0       # structure on the faces: a wrong wrap would show
    vol[:, :, -1] += 250.0
    psf = torch.from_numpy(psf_h).to(gpu)
    est = richardson_lucy(vol, psf, 4, 1e-6)
    assert float(est.min()) >= 0.0 and bool(torch.isfinite(est).all())
    flux_in, flux_out = float(vol.double().sum()), float(est.double().sum())
    assert abs(flux_out - flux_in) / flux_in < 1e-4
    flat = richardson_lucy(torch.full(shape, 42.0, device=gpu), psf, 3, 1e-6)
    assert float((flat - 42.0).abs().max()) < 1e-2
    monkeypatch.setenv("BH_RL_ENGINE_PAD", "0")
    assert richardson_lucy_plan(psf_h.shape, shape)[1] == "library"
    est_ref = richardson_lucy(vol, psf, 4, 1e-6)
    assert float((est - est_ref).abs().max()) <= FFT_TOL * float(est_ref.abs().max())


def test_phase_cross_corr_golden_and_oracle(gpu):
    """estimate_stabilization.phase_cross_corr: shifts exact, correlation volume within FFT tolerance."""
    from biahub_amd.estimate_stabilization import phase_cross_corr

    z = np.load(GOLDEN / "phase_cross_corr.npz")
    for j in range(3):
        for norm in (None, "magnitude", "classic"):
            sh, corr = phase_cross_corr(z[f"ref{j}"], z[f"mov{j}"], normalization=norm)
            assert sh.dtype == np.float32 and np.array_equal(sh, z[f"shift{j}_{norm}"]), (j, norm, sh)
            assert rel_err(corr, z[f"corr{j}_{norm}"]) <= FFT_TOL, (j, norm)
    rng = np.random.default_rng(1)
    ref = rng.random((32, 48, 40), dtype=np.float32)
    for roll in ((0, 0, 0), (15, -23, 19), (-16, 24, -20)):
        mov = np.roll(ref, roll, axis=(0, 1, 2))
        want, _ = O.phase_cross_corr(ref, mov, "magnitude")
        got, corr = phase_cross_corr(ref, mov, normalization="magnitude")
        assert np.array_equal(got, want) and corr.shape == ref.shape
    # power-of-two volumes run on the fused FFT engine (scrambled half-spectrum), everything else on hipFFT: same answers
    big = rng.random((32, 64, 128), dtype=np.float32)
    for roll in ((0, 0, 0), (5, -20, 33), (-16, 32, -64)):
        mov = np.roll(big, roll, axis=(0, 1, 2)) + 0.05 * rng.random(big.shape, dtype=np.float32)
        for norm in (None, "magnitude", "classic"):
            want, wcorr = O.phase_cross_corr(big, mov, norm)
            got, corr = phase_cross_corr(big, mov, normalization=norm)
            assert np.array_equal(got, want), (roll, norm, got, want)
            assert rel_err(corr, wcorr) <= FFT_TOL, (roll, norm)
    # z / y of 3 * 2^k: the engine's column passes start with a radix-3 step; the coefficient order changes, the answers do not
    for shape3, rolls in (((48, 96, 64), ((0, 0, 0), (7, -40, 21), (-24, 48, -32))), ((16, 32, 192), ((3, -9, 77), (-8, 16, -96))),
                          ((40, 160, 320), ((-20, 80, 160), (9, -70, 33)))):
        r3 = rng.random(shape3, dtype=np.float32)
        for roll in rolls:
            mov = np.roll(r3, roll, axis=(0, 1, 2)) + 0.05 * rng.random(r3.shape, dtype=np.float32)
            for norm in (None, "magnitude", "classic"):
                want, wcorr = O.phase_cross_corr(r3, mov, norm)
                got, corr = phase_cross_corr(r3, mov, normalization=norm)
                assert np.array_equal(got, want), (roll, norm, got, want)
                assert rel_err(corr, wcorr) <= FFT_TOL, (roll, norm)
    with pytest.raises(ValueError):
        phase_cross_corr(ref, ref[:-1], normalization=None)
    with pytest.raises(ValueError):
        phase_cross_corr(ref, ref, normalization="l2")


@pytest.mark.gpu
def test_phase_cross_corr_peak_only(gpu):
    """``want_corr=False`` on rows the wave-private X kernels take: the last inverse pass keeps the argmax candidates itself
    (xw::INV_ARGMAX) and the correlation volume is never stored — same shift as the search over the stored volume and as the
    oracle, including a tie between two equal peaks (np.argmax: the first one) and more row pairs than one launch round."""
    from biahub_amd.estimate_stabilization import phase_cross_corr_device

    rng = np.random.default_rng(21)
    for shape, rolls in (((16, 32, 512), ((0, 0, 0), (5, -11, 200), (-8, 16, -256))), ((8, 16, 1024), ((3, 7, -500),)),
                         ((4, 16, 2048), ((-2, 8, 1023),)), ((256, 128, 512), ((100, -50, 17),))):
        ref = rng.random(shape, dtype=np.float32)
        for roll in rolls:
            mov = np.roll(ref, roll, axis=(0, 1, 2)) + 0.05 * rng.random(shape, dtype=np.float32)
            for norm in (None, "magnitude", "classic"):
                stored, _ = phase_cross_corr_device(ref, mov, norm, want_corr=True)
                peak, corr = phase_cross_corr_device(ref, mov, norm, want_corr=False)
                assert corr is None and np.array_equal(peak, stored), (shape, roll, norm, peak, stored)
                if shape[0] <= 16:
                    want, _ = O.phase_cross_corr(ref, mov, norm)
                    assert np.array_equal(peak, want), (shape, roll, norm, peak, want)
    # two peaks of one height: the correlation of an impulse with two impulses
    ref = np.zeros((8, 16, 512), np.float32)
    mov = np.zeros_like(ref)
    ref[0, 0, 0] = 1.0
    mov[2, 5, 300] = mov[6, 3, 40] = 1.0
    for norm in (None, "magnitude"):
        stored, _ = phase_cross_corr_device(ref, mov, norm, want_corr=True)
        peak, _ = phase_cross_corr_device(ref, mov, norm, want_corr=False)
        assert np.array_equal(peak, stored), (norm, peak, stored)


@pytest.mark.gpu
def test_prepared_phase_cross_corr(gpu):
    """``PreparedPhaseCrossCorr`` (bh_phase_cross_corr_create / _apply): one image's spectrum kept, either as the product's first
    or as its second (conjugated) factor; ``roll`` makes the image of a call the stored one of the next.  Same shifts and the
    same correlation volumes as the one-shot call on every pair, on the fused engine (wave-private and tile X passes, register
    and LDS-stepped Z passes, a 3 * 2^k axis) and on the library path (odd X included)."""
    from biahub_amd.estimate_stabilization import PreparedPhaseCrossCorr, phase_cross_corr_device

    rng = np.random.default_rng(33)
    for shape in ((16, 32, 512), (32, 64, 128), (512, 16, 64), (48, 32, 64), (20, 30, 50), (9, 14, 31)):
        vols = [rng.random(shape, dtype=np.float32) for _ in range(2)]
        vols.append(np.roll(vols[0], (3, -5, 7), axis=(0, 1, 2)) + 0.05 * rng.random(shape, dtype=np.float32))
        vols.append(np.roll(vols[2], (-2, 4, 9), axis=(0, 1, 2)))
        for norm in (None, "magnitude", "classic"):
            for second in (False, True):
                with PreparedPhaseCrossCorr(vols[0], fixed_is_second=second, device=gpu) as h:
                    for v in vols[1:]:
                        pair = (v, vols[0]) if second else (vols[0], v)
                        want_s, want_c = phase_cross_corr_device(*pair, norm, gpu, want_corr=True)
                        for want_corr in (True, False):
                            got_s, got_c = h(v, norm, want_corr=want_corr)
                            assert np.array_equal(got_s, want_s), (shape, norm, second, want_corr, got_s, want_s)
                            if want_corr:
                                assert rel_err(got_c.cpu().numpy(), want_c.cpu().numpy()) <= 1e-6, (shape, norm, second)
                # the "previous timepoint" chain: each image against the one before it
                with PreparedPhaseCrossCorr(vols[0], fixed_is_second=second, device=gpu) as h:
                    for k in range(1, len(vols)):
                        pair = (vols[k], vols[k - 1]) if second else (vols[k - 1], vols[k])
                        want_s, want_c = phase_cross_corr_device(*pair, norm, gpu, want_corr=True)
                        got_s, got_c = h(vols[k], norm, want_corr=(k % 2 == 0), roll=True)
                        assert np.array_equal(got_s, want_s), (shape, norm, second, k, got_s, want_s)
                        if got_c is not None:
                            assert rel_err(got_c.cpu().numpy(), want_c.cpu().numpy()) <= 1e-6, (shape, norm, second, k)
    with PreparedPhaseCrossCorr(vols[0], device=gpu) as h:
        with pytest.raises(ValueError):
            h(vols[0][:-1])
        with pytest.raises(ValueError):
            h(vols[0], "l2")


@pytest.mark.gpu
def test_volume_pool_layout(gpu, monkeypatch):
    """``device.volume_pool``: torch tensors allocated inside it come from the library's allocator (``bh_torch_alloc``: 2-MiB
    physical chunks mapped in a shuffled order for blocks of 2 GiB and more — ``BH_ALLOC_VMM_MIN_MB`` — ``hipMalloc`` below).  Re-entrant, tensors outlive the context, freed
    blocks are reused, and the operators compute the same bits on pooled and on plain tensors."""
    from biahub_amd.deconvolve import richardson_lucy
    from biahub_amd.device import empty, get_context, volume_pool

    ctx = get_context(gpu)
    lib = ctx.lib
    p = lib.bh_torch_alloc(96 << 20, gpu.index or 0, None)          # a shuffled block straight through the C entry points
    assert p
    lib.bh_torch_free(p, 96 << 20, gpu.index or 0, None)
    with volume_pool(gpu):
        a = empty((48, 1024, 1024), torch.float32, gpu)               # 192 MiB (hipMalloc under the default threshold)
        with volume_pool(gpu):                                        # nested: a no-op, not an error
            b = torch.full((16, 64, 64), 3.0, device=gpu)
    a.copy_((torch.arange(a.numel(), device=gpu) % 1021).reshape(a.shape))
    assert float(b.sum()) == 3.0 * b.numel()
    idx = (47, 1023, 1023)
    assert float(a[idx]) == float((a.numel() - 1) % 1021) and float(a[0, 0, 5]) == 5.0
    ptr_a = a.data_ptr()
    del a
    with volume_pool(gpu):
        c = empty((48, 1024, 1024), torch.float32, gpu)               # the cached block comes back
    assert c.data_ptr() == ptr_a
    del c
    vol = O.synthetic_volume((32, 64, 128), seed=9, n_blobs=6)
    psf = O.gaussian_psf((5, 5, 5), (1.0, 1.2, 1.2))
    got = richardson_lucy(torch.from_numpy(vol).to(gpu), torch.from_numpy(psf).to(gpu), 3, 1e-6).cpu().numpy()
    monkeypatch.setenv("BH_VOLUME_POOL", "0")
    plain = richardson_lucy(torch.from_numpy(vol).to(gpu), torch.from_numpy(psf).to(gpu), 3, 1e-6).cpu().numpy()
    assert np.array_equal(got, plain)


@pytest.mark.gpu
def test_allocation_layout_switches(gpu, tmp_path):
    """The physical layout of the workspace (csrc/context.hip ``dev_alloc``: 2-MiB chunks mapped in a shuffled order through the
    HIP virtual-memory API, or ``hipMalloc`` with ``BH_ALLOC_VMM_MB=0``) is read once per process, so each layout runs in a
    process of its own: same Richardson-Lucy bits, and ``bh_alloc_layout`` reports what was in force.  ``BH_ALLOC_VMM_MIN_MB=1``
    makes the small test buffers take the virtual-memory path."""
    import json
    import subprocess
    import sys

    script = tmp_path / "layout_case.py"
    script.write_text(
        "import sys, json, numpy as np, torch\n"
        f"sys.path.insert(0, {str(ROOT)!r})\n"
        "from biahub_amd.deconvolve import PreparedRichardsonLucy\n"
        "from biahub_amd.device import alloc_layout\n"
        "rng = np.random.default_rng(4)\n"
        "vol = (rng.random((64, 128, 256), dtype=np.float32) * 300 + 50)\n"
        "ax = [np.arange(n) - (n - 1) / 2 for n in (7, 5, 5)]\n"
        "g = [np.exp(-0.5 * (a / 1.2) ** 2) for a in ax]\n"
        "psf = (g[0][:, None, None] * g[1][None, :, None] * g[2][None, None, :]).astype(np.float32)\n"
        "psf /= psf.sum()\n"
        "dev = torch.device('cuda', 0)\n"
        "with PreparedRichardsonLucy(psf, vol.shape, dev) as h:\n"
        "    out = h(torch.from_numpy(vol).to(dev), 4, 1e-6).cpu().numpy()\n"
        "    lay = alloc_layout()\n"
        "np.save(sys.argv[1], out)\n"
        "print(json.dumps(lay))\n")
    outs, lays = [], []
    for k, env in enumerate(({"BH_ALLOC_VMM_MIN_MB": "1"}, {"BH_ALLOC_VMM_MB": "0", "BH_VOLUME_POOL": "0"},
                             {"BH_ALLOC_VMM_MIN_MB": "1", "BH_ALLOC_VMM_SHUFFLE": "0", "BH_ALLOC_VMM_MB": "4"})):
        f = tmp_path / f"out{k}.npy"
        r = subprocess.run([sys.executable, str(script), str(f)], capture_output=True, text=True, env={**os.environ, **env})
        assert r.returncode == 0, r.stderr[-2000:]
        lays.append(json.loads(r.stdout.strip().splitlines()[-1]))
        outs.append(np.load(f))
    assert lays[0]["chunk_kib"] == 2048 and lays[0]["shuffled"] and lays[0]["live_blocks"] >= 1
    assert lays[1]["chunk_kib"] == 0 and not lays[1]["shuffled"] and lays[1]["live_blocks"] == 0 and not lays[1]["volume_pool"]
    assert lays[2]["chunk_kib"] == 4096 and not lays[2]["shuffled"] and lays[2]["live_blocks"] >= 1
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


# ----------------------------------------------------------------------------- registration estimate (N1)
def test_registration_kernels_vs_oracle(gpu):
    """bh_image_stats / bh_smooth_shrink / bh_sobel / bh_mattes_mi against their NumPy restatements."""
    import torch
    from biahub_amd.registration import metric as R

    vol = O.synthetic_volume((37, 70, 93), seed=11, n_blobs=120)
    t = torch.from_numpy(vol).to(gpu)
    st, want = R.image_stats(t), O.image_stats(vol)
    assert st["min"] == want[0] and st["max"] == want[1]
    assert abs(st["sum"] - want[2]) <= 1e-9 * want[2]
    assert np.allclose(st["center_of_mass"], want[3:6] / want[2], rtol=1e-9)

    for sigma, factor in (((2, 2, 2), (6, 6, 6)), ((1, 1, 1), (3, 3, 3)), ((0, 1.5, 0.5), (1, 2, 5)), ((0, 0, 0), (1, 1, 1))):
        got, off = R.smooth_shrink(t, sigma, factor)
        w, woff = O.smooth_shrink(vol, sigma, factor)
        assert tuple(got.shape) == w.shape and off == woff
        assert R.smooth_shrink_geometry(vol.shape, factor) == (w.shape, woff)
        assert rel_err(got.cpu().numpy(), w) <= 1e-6
    assert rel_err(R.sobel(t).cpu().numpy(), O.sobel(vol)) <= 1e-6

    M = _similarity(3.0, 1.03, (0.4, 1.5, -1.2))
    mov = O.affine_pull(vol, M, vol.shape, 1, O.BOUNDARY_ITK)
    fx, mv = torch.from_numpy(mov).to(gpu), t
    rng_ = (float(mov.min()), float(mov.max()), float(vol.min()), float(vol.max()))
    for P, stride, offset in ((np.eye(4)[:3], 1, 0), (M[:3], 5, 0), (_similarity(-8.0, 0.9, (3, 4, -6))[:3], 3, 2)):
        v, g, n = R.mattes_mi(fx, mv, P, rng_, bins=32, stride=stride, offset=offset)
        wv, wg, wn = O.mattes_mi(mov, vol, P, rng_, 32, stride, offset)
        assert n == wn                                          # same samples inside the moving volume
        assert abs(v - wv) <= 2e-5 * max(1.0, abs(wv))          # 2^-20 fixed-point histogram, f32 interpolation
        assert np.abs(g - wg).max() <= 2e-3 * np.abs(wg).max()
        v2, g2, n2 = R.mattes_mi(fx, mv, P, rng_, bins=32, stride=stride, offset=offset)
        assert v2 == v and n2 == n and np.array_equal(g2, g)    # bit-reproducible
    with pytest.raises(ValueError):
        R.mattes_mi(fx, mv, np.eye(4)[:3], (1.0, 1.0, 0.0, 1.0))  # empty intensity range
    with pytest.raises(ValueError):
        R.mattes_mi(fx, mv, np.eye(4)[:3], rng_, bins=5)


def test_estimate_registration_recovers_similarity(gpu):
    """BASELINE config 3 in miniature: arm B = arm A warped by a known similarity; the estimate must find it."""
    from biahub_amd.register import apply_affine_transform
    from biahub_amd.registration.ants import estimate, estimate_czyx

    shape = (48, 160, 160)
    arm_a = O.synthetic_volume(shape, seed=21, n_blobs=400)
    M = _similarity(2.0, 1.02, (0.9, -3.25, 5.75))
    arm_b = apply_affine_transform(arm_a, M, shape)          # arm_b(p) = arm_a(M p)
    arm_b = np.where(arm_b == 0, 110.0, arm_b).astype(np.float32)  # camera offset outside the overlap
    fwd, inv = estimate(ref=arm_b, mov=arm_a)
    T = fwd.matrix
    centre = np.append((np.array(shape) - 1) / 2, 1)
    assert np.abs(T[:3, :3] - M[:3, :3]).max() < 2e-3
    assert np.linalg.norm((T @ centre - M @ centre)[:3]) < 0.1
    assert np.allclose(inv.matrix @ T, np.eye(4), atol=1e-9)
    # registering with the estimate reproduces arm B (up to interpolation) inside the overlap
    rereg = apply_affine_transform(arm_a, T, shape)
    core = (slice(8, 40), slice(20, 140), slice(20, 140))
    assert np.abs(rereg[core] - arm_b[core]).mean() < 0.02 * arm_b[core].mean()
    # estimate_czyx: start from a rough initial guess and let the estimate correct it (registration/ants.py:281-366)
    init = _similarity(1.5, 1.0, (0.0, -2.0, 4.0))
    composed = estimate_czyx(arm_a[None], arm_b[None], init, crop=True)
    assert np.abs(composed.matrix[:3, :3] - M[:3, :3]).max() < 3e-3
    assert np.linalg.norm((composed.matrix @ centre - M @ centre)[:3]) < 0.15
    # 2-D images are one-plane volumes with in-plane parameters only (registration/ants.py:82-88)
    img = O.synthetic_volume((1, 192, 192), seed=8, n_blobs=120)[0]
    M2 = _similarity(1.5, 1.0, (0.0, 2.5, -1.75))
    img_b = O.affine_pull(img[None], M2, (1,) + img.shape, 1, O.BOUNDARY_ITK)[0]
    img_b = np.where(img_b == 0, 110.0, img_b).astype(np.float32)
    f2, i2 = estimate(ref=img_b, mov=img, ants_kwargs={"type_of_transform": "Rigid", "aff_shrink_factors": (2, 1),
                                                       "aff_iterations": (300, 60), "aff_smoothing_sigmas": (1, 0)})
    assert f2.matrix.shape == (3, 3) and np.abs(f2.matrix[:2, :2] - M2[1:3, 1:3]).max() < 3e-3
    c2 = np.array([95.5, 95.5, 1.0])
    assert np.linalg.norm((f2.matrix @ c2)[:2] - (M2[1:3][:, 1:] @ c2)[:2]) < 0.15
    with pytest.raises(ValueError, match="Dimension mismatch"):
        estimate(ref=arm_b, mov=arm_a[0])
    with pytest.raises(ValueError, match="NaN or zeros"):
        estimate_czyx(np.zeros((1,) + shape, np.float32), arm_b[None], init)


# ----------------------------------------------------------------------------- flat field (N3)
@pytest.mark.parametrize("bitsearch", [False, True])
def test_flat_field_golden_and_oracle(gpu, monkeypatch, bitsearch):
    """bh_median_z is np.median exactly (both selection kernels); bh_flat_field matches the reference expression."""
    if bitsearch:
        monkeypatch.setenv("BH_FF_BITSEARCH", "1")  # 8/16-bit input through the bit-search kernel too
    from biahub_amd.flat_field import _flat_field_czyx, _median_tiled, flat_field_zyx, median_z_device

    z = np.load(GOLDEN / "flat_field.npz")
    for j in range(11):
        data = z[f"in{j}"]
        med = _median_tiled(data, axis=0)
        assert med.dtype == z[f"median{j}"].dtype and np.array_equal(med, z[f"median{j}"]), j
        got = flat_field_zyx(data)
        want = z[f"flat{j}"].astype(np.float32)
        assert got.dtype == np.float32 and got.shape == want.shape
        # the pattern mean is a device float64 tree, numpy's a pairwise sum: at most the last float32 bit can differ
        assert np.abs(got - want).max() <= 2e-7 * np.abs(want).max(), j
    assert np.abs(_flat_field_czyx(z["czyx_in"], [1]) - z["czyx_out"]).max() <= 2e-7 * z["czyx_out"].max()
    assert np.array_equal(_flat_field_czyx(z["czyx_in"], [1])[0], z["czyx_in"][0].astype(np.float32))
    for a in range(3):
        assert np.array_equal(_median_tiled(z["axes_in"], axis=a), z[f"axes_median{a}"])
    rng = np.random.default_rng(12)
    # every staging width: 64 columns (uint16, Z = 1068 like a mantis position), 32 (float32, Z = 700), 8 (float32, Z = 4000)
    for shape, dt in (((1068, 3, 200), np.uint16), ((700, 2, 150), np.float32), ((4000, 1, 40), np.float32),
                      ((512, 5, 64), np.int16), ((300, 7, 33), np.uint8), ((2, 9, 129), np.uint16),
                      ((200, 4, 130), np.uint16), ((129, 3, 66), np.int16), ((511, 2, 256), np.uint16), ((384, 2, 300), np.uint8)):
        lo, hi = (-3000, 3000) if dt == np.int16 else ((0, 255) if dt == np.uint8 else (1, 4096))
        data = (rng.random(shape) * (hi - lo) + lo).astype(dt)
        if dt == np.float32:
            data[::3] = -data[::3]                      # negative floats exercise the key transform
        if dt == np.uint16:
            data = (data // 16 * 16 + 1).astype(dt)     # ties
        want = np.median(data, axis=0)
        got = median_z_device(data).cpu().numpy()
        assert np.array_equal(got, want.astype(np.float64)), (shape, dt)
    vol = (rng.random((64, 40, 96)) * 4000 + 100).astype(np.uint16)
    assert np.abs(flat_field_zyx(vol) - O.flat_field_zyx(vol).astype(np.float32)).max() <= 2e-7 * 65535
    with pytest.raises(ValueError, match="broadcast"):   # as the reference expression does for axis != 0
        flat_field_zyx(vol, axis=1)


@pytest.mark.parametrize("dt", [np.uint16, np.int16])
def test_flat_field_median_mixed_ranges(gpu, dt):
    """The 16-bit median's range-adaptive histogram selection (csrc/flatfield.hip median_hist_kernel): noise-like pixels (one
    histogram pass: a bin is a value), pixels spanning thousands of levels in the same tile (shifted bins + refinement for the
    tile), sparse bright and dark outliers that widen a pixel's range, bimodal pixels, even and odd Z, ties, ranges clamped at
    both ends of the key space — np.median exactly."""
    from biahub_amd.flat_field import median_z_device

    rng = np.random.default_rng(41)
    off = 0 if dt == np.uint16 else -20000
    for Z in (512, 257, 130, 37):
        Y, X = 3, 300
        d = rng.normal(600.0, 6.0, (Z, Y, X))
        d[:, 0, 10:40] += rng.random((Z, 30)) * 3000                      # wide pixels inside a narrow tile
        d[rng.random(d.shape) < 0.01] += 20000                             # sparse bright outliers: over the window
        d[:, 1, 100:160][rng.random((Z, 60)) < 0.2] -= 500                 # a fifth of the samples under the window
        zs = max(1, Z // 128)
        miss = np.ones(Z, bool)
        miss[np.minimum(np.arange(128) * zs, Z - 1)] = False
        d[miss, 2, 200:230] += 5000                                        # bimodal pixels: three quarters of the samples far above the rest
        d[~miss, 2, 240:260] -= 550                                        # ... a quarter below
        data = (np.clip(np.rint(d), 0, 60000) + off).astype(dt)
        want = np.median(data, axis=0).astype(np.float64)
        got = median_z_device(data).cpu().numpy()
        assert np.array_equal(got, want), (Z, np.argwhere(got != want)[:5])
    flat = np.full((64, 2, 128), 7 + off, dt)                              # zero range; constant pixels
    assert np.array_equal(median_z_device(flat).cpu().numpy(), np.full((2, 128), 7.0 + off))
    edge = np.zeros((100, 1, 130), dt) + (np.iinfo(dt).max - 3)            # window clamped at the top of the key range
    edge[::7] -= 90
    assert np.array_equal(median_z_device(edge).cpu().numpy(), np.median(edge, axis=0).astype(np.float64))
    low = np.zeros((100, 1, 130), dt) + (np.iinfo(dt).min + 2)             # ... and at the bottom
    low[::5] += 100
    assert np.array_equal(median_z_device(low).cpu().numpy(), np.median(low, axis=0).astype(np.float64))


# ----------------------------------------------------------------------------- bead detection / PSF estimate (N4)
def _bead_volume(shape, n, seed, sigma=(1.5, 1.2, 1.2)):
    r = np.random.default_rng(seed)
    v = r.normal(110.0, 3.0, shape).astype(np.float32)
    zz, yy, xx = np.ogrid[: shape[0], : shape[1], : shape[2]]
    centres = []
    for _ in range(n):
        c = r.uniform(0.15, 0.85, 3) * np.array(shape)
        centres.append(c)
        v += (r.uniform(800, 3000) * np.exp(-0.5 * (((zz - c[0]) / sigma[0]) ** 2 + ((yy - c[1]) / sigma[1]) ** 2
                                                     + ((xx - c[2]) / sigma[2]) ** 2))).astype(np.float32)
    return np.rint(v).astype(np.float32), np.array(centres)


def test_detect_peaks_golden(gpu):
    """bh_block_peaks is torch's avg_pool3d + max_pool3d bit for bit; detect_peaks returns the reference's peaks."""
    import json
    import torch
    from biahub_amd.characterize_psf import block_peaks, detect_peaks

    z = np.load(GOLDEN / "detect_peaks.npz")
    for j in range(3):
        kw = json.loads(str(z[f"kw{j}"]))
        kw["block_size"] = tuple(kw["block_size"])
        kw["exclude_border"] = tuple(kw["exclude_border"]) if kw["exclude_border"] else None
        val, idx = block_peaks(torch.from_numpy(z[f"vol{j}"]).to(gpu), kw["blur_kernel_size"], kw["block_size"])
        assert np.array_equal(val, z[f"pool_val{j}"]) and np.array_equal(idx, z[f"pool_idx{j}"]), j
        assert np.array_equal(detect_peaks(z[f"vol{j}"], **kw), z[f"peaks{j}"]), j
    with pytest.raises(ValueError, match="odd"):
        detect_peaks(z["vol0"], blur_kernel_size=4)
    with pytest.raises(ValueError, match="exclude_border"):
        detect_peaks(z["vol0"], exclude_border=(1, 2))


def test_estimate_psf_vs_oracle(gpu):
    """detect -> recentre -> average on a synthetic bead volume: same beads and (to float tolerance) the same PSF as the
    NumPy/SciPy restatement; the estimated PSF has the beads' widths."""
    from biahub_amd.characterize_psf import detect_peaks, extract_beads
    from biahub_amd.estimate_psf import estimate_psf

    shape, patch = (48, 120, 110), (15, 21, 21)
    vol, truth = _bead_volume(shape, 14, 5)
    kw = dict(block_size=(16, 16, 16), blur_kernel_size=3, nms_distance=6, min_distance=12, threshold_abs=300.0,
              max_num_peaks=100, exclude_border=(3, 5, 5))
    peaks = detect_peaks(vol, **kw)
    assert np.array_equal(peaks, O.detect_peaks(vol, **kw)) and 8 <= len(peaks) <= 14
    want, centres = O.recentre_and_average(vol, peaks, patch)
    psf = estimate_psf([vol], (1.0, 1.0, 1.0), patch_size=patch, bead_detection_settings=kw)
    assert psf.shape == patch and psf.dtype == np.float32 and psf.min() == 0.0 and psf.max() == 1.0
    assert np.abs(psf - want).max() <= 1e-5
    assert np.unravel_index(int(psf.argmax()), patch) == tuple(p // 2 for p in patch)   # recentred on the peak
    beads, offsets = extract_beads(vol, peaks, (1.0, 1.0, 1.0), patch_size=patch)
    assert len(beads) == len(centres) and all(b.shape == patch for b in beads)
    assert np.array_equal(np.array(offsets) + np.array(patch) // 2, centres)
    two = estimate_psf([vol, vol], (1.0, 1.0, 1.0), patch_size=patch, bead_detection_settings=kw)  # positions pooled
    assert np.abs(two - psf).max() <= 1e-6
    with pytest.raises(ValueError, match="No beads"):
        estimate_psf([np.full(shape, 100.0, np.float32)], (1, 1, 1), patch_size=patch, bead_detection_settings=kw)


def test_pcc_chain_golden(gpu):
    """The phase-cross-correlation call chain around bh_phase_cross_corr equals the reference's (incl. its sign and
    index conventions): phase_cross_corr_padding, get_tform_from_pcc."""
    from biahub_amd.estimate_stabilization import get_tform_from_pcc, phase_cross_corr_padding

    z = np.load(GOLDEN / "pcc_chain.npz")
    for j in range(3):
        for norm in (None, "magnitude"):
            peak, corr = phase_cross_corr_padding(z[f"ref{j}"], z[f"mov{j}"], normalization=norm)
            assert tuple(peak) == tuple(z[f"peak{j}_{norm}"]), (j, norm, peak)
            assert corr.shape == z[f"corr{j}_{norm}"].shape and rel_err(corr, z[f"corr{j}_{norm}"]) <= FFT_TOL
    stack = z["stack"]
    first = np.broadcast_to(stack[0], stack.shape)
    for t in (1, 2):
        for ft in ("custom", "custom_padding"):
            tr, sh, _ = get_tform_from_pcc(t, stack, first, function_type=ft, normalization="magnitude")
            assert np.array_equal(tr, z[f"tform{t}_{ft}"]), (t, ft, tr)
            assert np.array_equal(np.asarray(sh, dtype=np.float64), z[f"tshift{t}_{ft}"])


# ----------------------------------------------------------------------------- BASELINE config 2 at full size
def test_full_size_properties_config2(gpu):
    """The bench workload's shape (512, 2048, 2048): properties that hold at any size, checked where the oracle cannot go.

    R-L: non-negative, finite, flux-conserving; a shifted delta image is deconvolved to the same shifted answer (circular
    shift equivariance).  Deskew: shape, linearity, partition of unity, determinism.  Flat field: a per-pixel pattern times
    a z profile gives the analytic median.  Affine: identity is exact, an integer translation is a shifted copy.
    """
    from biahub_amd.deconvolve import richardson_lucy
    from biahub_amd.deskew import fast_deskew_zyx, get_deskewed_data_shape
    from biahub_amd.flat_field import flat_field_device, median_z_device
    from biahub_amd.register import affine_device

    shape = (512, 2048, 2048)
    g = torch.Generator(device=gpu).manual_seed(11)
    vol = (torch.rand(shape, generator=g, device=gpu) * 300 + 100).round_()
    psf = torch.from_numpy(O.gaussian_psf((33, 17, 17), (3.0, 1.5, 1.5))).to(gpu)
    est = richardson_lucy(vol, psf, 3, 1e-6)
    assert float(est.min()) >= 0.0 and bool(torch.isfinite(est).all())
    assert abs(float(est.double().sum()) / float(vol.double().sum()) - 1.0) < 1e-4
    rolled = richardson_lucy(torch.roll(vol, (7, -33, 129), (0, 1, 2)), psf, 3, 1e-6)
    assert float((torch.roll(est, (7, -33, 129), (0, 1, 2)) - rolled).abs().max()) <= FFT_TOL * float(est.max())
    del rolled

    kw = dict(ls_angle_deg=36.17, px_to_scan_ratio=0.371, keep_overhang=True, average_n_slices=3)
    d_est = fast_deskew_zyx(est, **kw)
    assert tuple(d_est.shape) == get_deskewed_data_shape(shape, 36.17, 0.371, True, 3)[0] == (683, 2048, 3034)
    d_vol = fast_deskew_zyx(vol, **kw)
    mix = fast_deskew_zyx(2 * est + vol, **kw)
    mix -= d_vol
    mix -= 2 * d_est
    assert float(mix.abs().max()) <= 1e-5 * float(d_est.abs().max() * 3)          # linearity
    del mix, d_vol
    assert torch.equal(fast_deskew_zyx(est, **kw), d_est)                           # deterministic
    del d_est
    ones = fast_deskew_zyx(torch.ones(shape, device=gpu), **kw)
    assert float(ones.max()) <= 1.0 + 1e-6 and float(ones.min()) >= 0.0             # partition of unity
    del ones, est

    ident = affine_device(vol, np.eye(4), shape, "linear")
    assert torch.equal(ident, vol)                                                  # identity is exact
    m = np.eye(4)
    m[:3, 3] = (3, -5, 17)
    moved = affine_device(vol, m, shape, "linear")                                  # out(p) = in(p + t)
    assert torch.equal(moved[:500, 10:, :2000], vol[3:503, 5:2043, 17:2017])
    assert float(moved[509:].abs().max()) == 0.0 and float(moved[:, :5].abs().max()) == 0.0
    del ident, moved

    pat = (torch.rand(shape[1:], generator=g, device=gpu) * 3000 + 200).round_()
    prof = (torch.arange(shape[0], device=gpu) % 5).float()                         # 103, 103, 102, 102, 102 samples
    raw = (pat[None] + prof[:, None, None]).to(torch.uint16)
    del vol
    assert torch.equal(median_z_device(raw), (pat + 2).double())                    # both middle ranks carry profile 2
    flat = flat_field_device(raw)
    want_mean = float((pat + 2).double().mean())
    z7 = raw[7].double() / (pat + 2).double() * want_mean
    assert float((flat[7].double() - z7).abs().max()) <= 2e-7 * float(z7.max())


# ----------------------------------------------------------------------------- binning (process-with-config)
def test_binning_golden_and_oracle(gpu):
    """process_data.binning_czyx: integer inputs bit-exact (window sums are exact in float32, the stretch is evaluated
    in the reference's operation order), float32 inputs to summation-order tolerance."""
    import json
    from biahub_amd.process_data import binning_czyx, process_czyx
    from biahub_amd.settings import ProcessingFunctions

    z = np.load(GOLDEN / "binning.npz")
    for j in range(7):
        kw = json.loads(str(z[f"kw{j}"]))
        got = binning_czyx(z[f"in{j}"], **kw)
        want = z[f"out{j}"]
        assert got.dtype == want.dtype and got.shape == want.shape
        if np.issubdtype(want.dtype, np.integer):
            assert np.array_equal(got, want), j
        else:
            assert rel_err(got, want) <= 1e-6, j
    rng = np.random.default_rng(4)
    big = (rng.random((2, 32, 96, 160)) * 60000).astype(np.uint16)
    for f, mode in (((1, 2, 2), "sum"), ((2, 4, 4), "mean"), ((4, 3, 5), "sum")):
        assert np.array_equal(binning_czyx(big, f, mode), O.binning_czyx(big, f, mode)), (f, mode)
    proc = ProcessingFunctions(function="biahub.process_data.binning_czyx", input_channels=[0],
                               kwargs={"binning_factor_zyx": [1, 2, 2], "mode": "sum"})
    assert np.array_equal(process_czyx(big, [proc]), O.binning_czyx(big, (1, 2, 2), "sum"))
    with pytest.raises(ValueError, match="cannot reshape"):
        binning_czyx(big, (1, 5, 2))
    with pytest.raises(ValueError, match="Invalid mode"):
        binning_czyx(big, (1, 2, 2), "median")


# ---- chunk codec: Blosc permutations on the GPU and device-side volume I/O ---------------------
@pytest.mark.gpu
@pytest.mark.parametrize("typesize", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("mode", [1, 2])
def test_blosc_filters_device_bit_exact(gpu, typesize, mode):
    """bh_blosc_filter / bh_blosc_unfilter against the NumPy oracle (itself pinned to c-blosc streams): full blocks,
    ragged last blocks, blocks whose element count is not a multiple of 8 (bit shuffle then leaves them alone), tails
    shorter than one element."""
    from biahub_amd import codecs
    from oracle import codec_np

    rng = np.random.default_rng(typesize * 7 + mode)
    for nbytes, blocksize in ((1 << 20, 1 << 16), (1_000_003, 4096 * typesize), (70_001, 70_001), (5000 * typesize + 1, 2048 * typesize),
                              (typesize * 8 * 37 + typesize - 1, 1 << 20), (100, 64), (3, 1 << 10), (12 * typesize, 5 * typesize)):
        raw = rng.integers(0, 256, nbytes, dtype=np.uint8)
        want = codec_np.filter_blocks(raw, blocksize, typesize, mode)
        src = torch.from_numpy(raw).to(gpu)
        dst = torch.empty_like(src)
        codecs.filter_device(src, dst, blocksize, typesize, mode)
        assert np.array_equal(dst.cpu().numpy(), want), (nbytes, blocksize)
        back = torch.empty_like(src)
        codecs.unfilter_device(dst, back, blocksize, typesize, mode)
        assert np.array_equal(back.cpu().numpy(), raw), (nbytes, blocksize)
        assert np.array_equal(codec_np.unfilter(want, nbytes, blocksize, typesize, mode), raw)


@pytest.mark.gpu
def test_blosc_device_unfilter_on_c_blosc_streams(gpu):
    """Streams written by the real c-blosc 1.21.0: entropy-decode on the host, un-shuffle on the GPU."""
    from biahub_amd import codecs

    z = np.load(GOLDEN / "blosc_streams.npz")
    names = sorted(k[: -len("__blosc")] for k in z.files if k.endswith("__blosc"))
    done = 0
    for n in names:
        h, shuffled = codecs.blosc_decode_blocks(z[f"{n}__blosc"])
        if h.nbytes == 0:
            continue
        src = torch.from_numpy(shuffled.copy()).to(gpu)
        dst = torch.empty_like(src)
        codecs.unfilter_device(src, dst, h.blocksize, h.typesize, 0 if h.memcpyed else h.shuffle_mode)
        assert np.array_equal(dst.cpu().numpy(), z[f"{n}__raw"]), n
        done += 1
    assert done >= 60


@pytest.mark.gpu
def test_lz4_device_decodes_c_blosc_streams(gpu):
    """csrc/lz4.hip, decoder: every lz4 stream the real c-blosc 1.21.0 wrote (tests/golden/blosc_streams.npz: split and unsplit
    blocks, stored splits, short last blocks, match lengths of hundreds of kilobytes) decodes on the GPU to the bytes the host
    decoder gives, and un-shuffles to the original."""
    from biahub_amd import codecs

    z = np.load(GOLDEN / "blosc_streams.npz")
    names = sorted(k[: -len("__blosc")] for k in z.files if k.endswith("__blosc"))
    done = 0
    for n in names:
        stream = z[f"{n}__blosc"].tobytes()
        h = codecs.BloscHeader(stream)
        if h.codec != "lz4" or h.memcpyed or h.nbytes == 0:
            continue
        _, want = codecs.blosc_decode_blocks(stream)
        out = torch.empty(h.nbytes, dtype=torch.uint8, device=gpu)
        codecs.blosc_lz4_decode_blocks_device(stream, out)
        assert np.array_equal(out.cpu().numpy(), want), n
        raw = torch.empty_like(out)
        codecs.unfilter_device(out, raw, h.blocksize, h.typesize, h.shuffle_mode)
        assert np.array_equal(raw.cpu().numpy(), z[f"{n}__raw"]), n
        done += 1
    assert done >= 12
    # a damaged stream is reported, not decoded into garbage: the runs volume is a handful of long matches — zero their offsets
    bad = bytearray(z["lz4runs_u2_s2__blosc"].tobytes())
    h, soff, csize, _, dlen = codecs.blosc_lz4_stream_table(bytes(bad))
    i = int(np.argmax(csize != dlen))  # the first compressed stream
    for k in range(int(soff[i]), int(soff[i]) + int(csize[i])):
        bad[k] = 0x0F if bad[k] else 0  # every token now announces a match after no literals, every offset reads 0 or 0x0f0f
    with pytest.raises(Exception, match="corrupt"):
        codecs.blosc_lz4_decode_blocks_device(bytes(bad), torch.empty(h.nbytes, dtype=torch.uint8, device=gpu))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,mode", [("u2", 2), ("f4", 2), ("u2", 1), ("u1", 0)])
def test_lz4_device_compressor_frames(gpu, dtype, mode):
    """csrc/lz4.hip, compressor + frame assembly: Blosc-lz4 frames made ON the GPU (permute -> LZ4 blocks -> packed frames) are
    read back by the host decoder, by the device decoder and — where this image has it — by the real c-blosc; compressible
    chunks shrink, incompressible ones come out as stored blocks, a short last block and a last frame of pure zeros included."""
    import ctypes
    import os

    from biahub_amd import codecs

    rng = np.random.default_rng(12)
    ts = np.dtype(dtype).itemsize
    n_el = 190_000  # per chunk: 380 / 760 KB -> two or three 256-KiB blocks, the last one short
    x = np.arange(n_el)
    chunks = [
        (300 + 200 * np.sin(x / 37.0) + rng.poisson(2, n_el)).astype(dtype),     # camera-like: compressible
        rng.integers(0, 256, n_el * ts, dtype=np.uint8).view(dtype),                # noise in every bit: incompressible
        np.zeros(n_el, dtype),                                                      # one long run
        np.where(x % 5000 < 4000, 1234, x % 251).astype(dtype),                     # runs and ramps
    ]
    raw = np.concatenate([c.view(np.uint8) for c in chunks])
    cbytes = n_el * ts
    bsz = codecs.default_blocksize(ts)
    src = torch.from_numpy(raw).to(gpu)
    filt = torch.empty_like(src)
    for i in range(len(chunks)):
        codecs.filter_device(src[i * cbytes:(i + 1) * cbytes], filt[i * cbytes:(i + 1) * cbytes], bsz, ts, mode)
    packed, offs = codecs.blosc_lz4_compress_device(filt, len(chunks), cbytes, bsz, ts, mode)
    host = packed[: offs[-1]].cpu().numpy()
    lib = ctypes.CDLL("/opt/conda/lib/libblosc.so.1") if os.path.exists("/opt/conda/lib/libblosc.so.1") else None
    if lib is not None:
        lib.blosc_decompress_ctx.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    sizes = []
    for i, c in enumerate(chunks):
        frame = host[offs[i]: offs[i + 1]]
        h = codecs.BloscHeader(frame.tobytes())
        assert (h.nbytes, h.blocksize, h.typesize, h.codec, h.shuffle_mode) == (cbytes, min(bsz, cbytes), ts, "lz4", mode if ts > 1 or mode != 1 else 0)
        stream = frame[: h.cbytes].tobytes()
        assert np.array_equal(codecs.blosc_decompress(stream), c.view(np.uint8)), i      # host decoder (pyarrow's lz4)
        out = torch.empty(cbytes, dtype=torch.uint8, device=gpu)
        codecs.blosc_lz4_decode_blocks_device(stream, out)                                  # device decoder
        assert torch.equal(out, filt[i * cbytes:(i + 1) * cbytes]), i
        if lib is not None:                                                                 # the real library
            buf = np.frombuffer(stream, np.uint8).copy()
            back = np.empty(cbytes, np.uint8)
            assert lib.blosc_decompress_ctx(buf.ctypes.data, back.ctypes.data, back.size, 1) == cbytes, i
            assert np.array_equal(back, c.view(np.uint8)), i
        sizes.append(h.cbytes)
    # every frame of the "volume" in one upload and one launch (what read_volume_device does)
    allout = torch.zeros_like(filt)
    fr = [host[offs[i]: offs[i + 1]].tobytes() for i in range(len(chunks))]
    hs = codecs.blosc_lz4_decode_frames_device(fr, allout, [i * cbytes for i in range(len(chunks))])
    assert len(hs) == len(chunks) and torch.equal(allout, filt)
    with pytest.raises(ValueError, match="fit"):
        codecs.blosc_lz4_decode_frames_device(fr[:1], allout[: cbytes - 1], [0])
    # (a sine + Poisson noise wrapped into single bytes leaves LZ4 little to match: only the wider types must shrink there)
    assert sizes[0] < (0.8 if ts > 1 else 1.0) * cbytes and sizes[2] < 0.02 * cbytes and sizes[3] < 0.3 * cbytes, sizes
    nb = -(-cbytes // bsz)
    assert cbytes + 16 + 4 * nb <= sizes[1] <= cbytes + 16 + 8 * nb, sizes  # noise: every block stored


@pytest.mark.gpu
def test_zarr_device_volume_io_lz4(gpu, tmp_path, monkeypatch):
    """A Blosc-lz4 store written from the device carries compressed bytes only across PCIe (csrc/lz4.hip) and reads back through
    every reader: host, device (LZ4 on the GPU), and with BH_LZ4_DEVICE=0 the host entropy coder writes what the device reads."""
    from biahub_amd import io

    shape = (1, 2, 21, 256, 320)  # 21 planes in chunks of 8: the last chunk overhangs; 1.3-MB chunks = five 256-KiB blocks
    comp = {"id": "blosc", "cname": "lz4", "clevel": 1, "shuffle": 2, "blocksize": 0}
    io.create_empty_position(tmp_path / "p", ["a", "b"], shape, chunks=(1, 1, 8, 256, 320), dtype=np.uint16, version="0.4", compressor=comp)
    arr = io.open_ome_zarr(tmp_path / "p").data
    rng = np.random.default_rng(2)
    v0 = (rng.poisson(4, shape[2:]) + 100).astype(np.uint16)
    v1 = (rng.poisson(9, shape[2:]) + 300).astype(np.uint16)
    arr.write_volume_device(0, 0, torch.from_numpy(v0).to(gpu))   # device LZ4 -> host reader, device reader
    assert np.array_equal(arr.read_volume(0, 0), v0)
    assert np.array_equal(arr.read_volume_device(0, 0, gpu).cpu().numpy(), v0)
    monkeypatch.setenv("BH_LZ4_DEVICE", "0")
    arr.write_volume_device(0, 1, torch.from_numpy(v1).to(gpu))   # host LZ4 (pyarrow) ...
    monkeypatch.delenv("BH_LZ4_DEVICE")
    assert np.array_equal(arr.read_volume_device(0, 1, gpu).cpu().numpy(), v1)   # ... -> device decoder
    f = tmp_path / "p" / "0" / "0/0/0/0/0"
    assert f.stat().st_size < 8 * 256 * 320 * 2 // 2  # it really is compressed


@pytest.mark.gpu
@pytest.mark.parametrize("version,shards_ratio,shuffle", [("0.4", None, 2), ("0.4", None, 1), ("0.5", None, 2), ("0.5", (1, 1, 2, 1, 1), 2)])
def test_zarr_device_volume_io(gpu, tmp_path, version, shards_ratio, shuffle):
    """write_volume_device / read_volume_device (GPU-side shuffle) interoperate with the host reader / writer."""
    from biahub_amd import io

    shape = (1, 2, 21, 96, 160)  # 21 planes in chunks of 8: the last chunk overhangs
    comp = {"id": "blosc", "cname": "zstd", "clevel": 1, "shuffle": shuffle, "blocksize": 0}
    io.create_empty_position(tmp_path / "p", ["a", "b"], shape, chunks=(1, 1, 8, 96, 160), dtype=np.uint16, version=version,
                             compressor=comp, shards_ratio=shards_ratio)
    arr = io.open_ome_zarr(tmp_path / "p").data
    rng = np.random.default_rng(2)
    v0 = (rng.poisson(4, shape[2:]) + 100).astype(np.uint16)
    v1 = (rng.poisson(9, shape[2:]) + 300).astype(np.uint16)
    arr.write_volume_device(0, 0, torch.from_numpy(v0).to(gpu))   # device writer -> host reader
    arr.write_volume(0, 1, v1)                                     # host writer -> device reader
    assert np.array_equal(arr.read_volume(0, 0), v0)
    got = arr.read_volume_device(0, 1, gpu)
    assert got.is_cuda and got.dtype == torch.uint16 and np.array_equal(got.cpu().numpy(), v1)
    assert np.array_equal(arr.read_volume_device(0, 0, gpu).cpu().numpy(), v0)
    f = tmp_path / "p" / "0" / ("c/0/0/0/0/0" if version == "0.5" else "0/0/0/0/0")
    assert f.stat().st_size < 8 * 96 * 160 * 2 * (2 if shards_ratio else 1) // 2      # it really is compressed


# ----------------------------------------------------------------------------- wave-private X passes (csrc/fftconv_xw.inc)
@pytest.mark.parametrize("shape,pshape", [
    ((8, 32, 2048), (5, 5, 9)),      # M = 1024: one row pair per wavefront, radix 8 x 16 x 8
    ((4, 64, 1024), (3, 7, 7)),      # M = 512: two row pairs per wavefront, radix 8 x 8 x 8
    ((8, 64, 512), (3, 5, 7)),       # M = 256: four row pairs per wavefront, radix 8 x 4 x 8
    ((8, 32, 1536), (3, 5, 9)),      # M = 768 = 3 x 256: the radix-3 kernels (csrc/fftconv_x3.inc), two row pairs per wavefront
    ((4, 64, 3072), (3, 3, 11)),     # M = 1536 = 3 x 512: one row pair per wavefront
    ((12, 40, 1500), (3, 5, 9)),     # wrap-padded box (16, 64, 1536): forward, fused ratio and plain inverse of the radix-3 kernels
    ((24, 96, 1024), (5, 5, 5)),     # Z, Y of 3 * 2^k: radix-3 column passes around the new X passes
    ((12, 40, 2048), (3, 5, 11)),    # Z, Y too short for the engine as they are: the wrap-padded box (plain inverse with store)
])
def test_wave_private_x_passes(gpu, shape, pshape, monkeypatch):
    """Rows of 512 / 1024 / 2048 voxels (csrc/fftconv_xw.inc) and of 1536 / 3072 voxels (radix-3 first step across the thirds of
    a row, csrc/fftconv_x3.inc) run the wave-private X passes (register FFT stages + two wave-local LDS exchanges, own
    column order of the spectrum).  Richardson-Lucy (fused ratio / update kernels, forward, plain inverse), Tikhonov (filter
    staged through the column map) and phase cross-correlation agree with the oracle and with the tile-based X passes
    (BH_FC_XW=0) on the same inputs."""
    from biahub_amd.deconvolve import compute_tranfser_function, deconvolve, richardson_lucy
    from biahub_amd.estimate_stabilization import phase_cross_corr

    vol = O.synthetic_volume(shape, seed=21, n_blobs=12)
    vol[:, 0, :] += 400.0
    vol[:, :, -1] += 250.0           # structure on the faces: wrap-around / column-order mistakes show up
    psf = O.gaussian_psf(pshape, tuple(max(p / 4.0, 0.8) for p in pshape))
    psf[0, 0, 0] += 0.02             # asymmetric: convolution and correlation differ
    v, pt = torch.from_numpy(vol).to(gpu), torch.from_numpy(psf).to(gpu)
    for it in (1, 4):
        want = O.richardson_lucy_zyx(vol, psf, iterations=it, eps=1e-6)
        monkeypatch.delenv("BH_FC_XW", raising=False)
        new = richardson_lucy(v, pt, it, 1e-6).cpu().numpy()
        monkeypatch.setenv("BH_FC_XW", "0")
        old = richardson_lucy(v, pt, it, 1e-6).cpu().numpy()
        monkeypatch.delenv("BH_FC_XW")
        assert rel_err(new, want) <= FFT_TOL, (it, rel_err(new, want))
        assert rel_err(new, old) <= 2e-5, (it, rel_err(new, old))
    # Tikhonov: the reference's natural-order transfer function is staged into the new column order.  White noise on top:
    # the plain inverse X pass must use THIS pair's Nyquist bins (a smooth volume has next to no energy there, and a build
    # that read the prefetched pair's bins passed on it)
    tf = compute_tranfser_function(psf, shape)
    noisy = (vol + np.random.default_rng(8).random(shape, dtype=np.float32) * 200).astype(np.float32)
    got = deconvolve(noisy[None], transfer_function=tf, regularization_strength=1e-2)
    assert rel_err(got, O.deconvolve_czyx(noisy[None], tf, 1e-2)) <= FFT_TOL
    # phase cross-correlation: bare forward / inverse pair of the engine
    mov = np.roll(vol, (1, -3, 17), axis=(0, 1, 2))
    shift, _ = phase_cross_corr(vol, mov, normalization="magnitude")
    want_shift, _ = O.phase_cross_corr(vol, mov, "magnitude")
    assert np.array_equal(np.asarray(shift), np.asarray(want_shift)), (shift, want_shift)


# ----------------------------------------------------------------------------- compute-tf / apply-inv-tf (BASELINE config 5)
# Parity unpinned: waveorder 3.0.5 (the arithmetic behind biahub's compute-tf / apply-inv-tf) is absent from the reference
# tree; the oracle restates its published algorithm and these tests hold the HIP path to that restatement.
@pytest.mark.parametrize("shape,yx,dz,pad,invert", [
    ((8, 16, 20), 0.1, 0.25, 0, False),     # native sampling
    ((6, 12, 10), 0.1, 0.25, 2, True),      # z padding, inverted contrast
    ((5, 10, 12), 0.16, 0.3, 1, False),     # pixel above the transverse Nyquist: computed on a 2x grid, central cuboid kept
])
def test_phase_transfer_function_vs_oracle(gpu, shape, yx, dz, pad, invert):
    from biahub_amd.compute_transfer_function import phase_transfer_function_3d

    args = (shape, yx, dz, 0.45, pad, 1.3, 0.5, 1.2, invert)
    re, im = phase_transfer_function_3d(*args)
    wre, wim = O.wo_phase_transfer_function_3d(*args)
    assert tuple(re.shape) == (shape[0] + 2 * pad,) + shape[1:] == wre.shape
    scale = np.abs(wre).max()
    assert np.abs(re.cpu().numpy() - wre).max() <= 2e-5 * scale
    assert np.abs(im.cpu().numpy() - wim).max() <= 2e-5 * scale


@pytest.mark.parametrize("shape,yx,dz,pad", [((8, 16, 20), 0.1, 0.25, 0), ((6, 12, 10), 0.15, 0.4, 2)])
def test_fluorescence_transfer_function_vs_oracle(gpu, shape, yx, dz, pad):
    from biahub_amd.compute_transfer_function import fluorescence_transfer_function_3d

    args = (shape, yx, dz, 0.507, pad, 1.3, 1.2)
    otf = fluorescence_transfer_function_3d(*args).cpu().numpy()
    want = O.wo_fluorescence_transfer_function_3d(*args)
    assert otf.shape == want.shape and np.abs(otf - want).max() <= 2e-5


@pytest.mark.parametrize("shape,pad", [
    ((16, 32, 64), 0),     # the fused engine, filter multiplied in its Z pass
    ((12, 32, 64), 2),     # padded to 16 planes: engine, z crop
    ((15, 21, 25), 0),     # library transforms
    ((10, 21, 25), 3),     # library transforms, z padding (mirrored edge planes, as waveorder's pad_zyx_along_z)
    ((3, 21, 25), 5),      # z_padding >= Z: the pad planes stay zero
    ((8, 64, 1024), 0),    # rows of 1024 voxels: wave-private X passes, the filter staged through their column order
])
def test_apply_inverse_transfer_function_vs_oracle(gpu, shape, pad):
    from biahub_amd.apply_inverse_transfer_function import apply_inverse_transfer_function_czyx, apply_inverse_transfer_function_zyx

    rng = np.random.default_rng(5)
    vol = (rng.random(shape, dtype=np.float32) * 50 + 100).astype(np.float32)
    tshape = (shape[0] + 2 * pad,) + shape[1:]
    # a complex transfer function with no symmetry at all: only its Hermitian part may act on a real volume
    H = (rng.standard_normal(tshape) + 1j * rng.standard_normal(tshape)).astype(np.complex64) * 0.3
    for reg, normalize in ((1e-2, True), (1e-3, False)):
        want = O.wo_apply_inverse_transfer_function(vol, H, pad, reg, normalize)
        got = apply_inverse_transfer_function_zyx(vol, H, pad, reg, normalize).cpu().numpy()
        assert rel_err(got, want) <= FFT_TOL, (reg, rel_err(got, want))
    # a real optical transfer function (fluorescence) and the czyx operator
    Hr = np.abs(np.fft.fftn(O.gaussian_psf((5, 5, 5), (1.0, 1.0, 1.0)), tshape, axes=(0, 1, 2))).astype(np.float32)
    want = O.wo_apply_inverse_transfer_function(vol, Hr, pad, 1e-3, False)
    got = apply_inverse_transfer_function_czyx(vol[None], transfer_function=Hr, z_padding=pad, regularization_strength=1e-3)
    assert got.shape == (1,) + shape and rel_err(got[0], want) <= FFT_TOL
    # bfloat16 storage of the staged filter: an 8-bit mantissa per filter bin (engine shapes only)
    from biahub_amd import _lib

    if pad == 0 and shape[2] >= 64 and shape in ((16, 32, 64), (8, 64, 1024)):
        g16 = apply_inverse_transfer_function_zyx(vol, Hr, pad, 1e-3, False, filter_storage="bf16").cpu().numpy()
        assert 1e-7 < rel_err(g16, want) <= 1e-2
    elif shape == (15, 21, 25):
        with pytest.raises(ValueError, match="bfloat16"):
            apply_inverse_transfer_function_zyx(vol, Hr, pad, 1e-3, False, filter_storage="bf16")
    with pytest.raises(ValueError, match="transfer function shape"):
        apply_inverse_transfer_function_zyx(vol, Hr[1:], pad)
    del _lib


def test_config5_full_size_bf16_sweep(gpu):
    """BASELINE config 5 at its size: compute-tf (phase) + apply_inverse_transfer_function on (512, 2048, 2048), staged filter
    in float32 vs bfloat16 over four regularisation strengths (tools/config5_sweep.py; the numbers are committed under
    profiles/).  No oracle finishes at this size: the asserts are the bf16 tolerance and size-independent properties —
    zero mean of a normalised phase reconstruction, linearity, translation covariance, agreement with bh_tikhonov for a real
    transfer function.  Parity unpinned (waveorder absent from the reference tree)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("config5_sweep", ROOT / "tools" / "config5_sweep.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.sweep((512, 2048, 2048), emit=lambda s: None)
    assert len(res["rows"]) == 4
    for row in res["rows"]:
        # bfloat16 keeps 8 significant bits of each filter bin: errors of a few 1e-3 of the peak, never percent-level
        assert 1e-6 < row["max_rel_err_bf16_vs_f32"] < 2e-2, row
        assert row["rms_rel_err_bf16_vs_f32"] < 5e-3, row
        assert row["mean_over_std"] < 1e-3, row           # x / mean - 1 has no DC: neither has its reconstruction
    assert res["linearity_err"] < 2e-5 and res["shift_err"] < 2e-5 and res["vs_bh_tikhonov"] < 2e-5, res


def _load_tool(name):
    import importlib.util

    spec = importlib.util.spec_from_file_location(name, ROOT / "tools" / f"{name}.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_config3_full_size_estimate_and_register(gpu):
    """BASELINE config 3 at its size: two-arm (256, 1024, 1024) volumes, arm B = arm A under a known similarity (2 deg, 1.02x,
    fractional translation); estimate-registration (Mattes-MI similarity estimate from an initial guess 4 voxels off) must
    recover it, and `register` warps with the result (tools/config3_bench.py).  Parity unpinned (ANTs absent): the recovered
    matrix is compared with the ground truth."""
    res = _load_tool("config3_bench").run((256, 1024, 1024), echo=lambda *a: None)
    assert res["dA"] < 1e-4, res           # measured 2e-6
    assert res["centre_error"] < 0.05, res  # voxels; measured 0.002
    assert res["estimate_s"] < 5.0, res


def test_config4_position_chain_full_size(gpu):
    """BASELINE config 4's per-position chain on one GPU at its volume size: T = 4, C = 2, raw (256, 1024, 1024) uint16 ->
    deskew (example settings, fill mean) -> Richardson-Lucy x10 on the deskewed volume -> phase-cross-correlation drift
    against t = 0 -> stabilize.  Whole-volume oracles do not finish at this size; each stage is held to the oracle on a
    sub-volume where the stage is local (deskew: an x slab of the raw volume is a y slab of the result; stabilize: a crop with
    its one-voxel halo) and to size-independent properties where it is global (R-L: flux, sign; drift: the planted shift)."""
    from biahub_amd.deconvolve import richardson_lucy
    from biahub_amd.deskew import fast_deskew_zyx
    from biahub_amd.estimate_stabilization import phase_cross_corr_device
    from biahub_amd.register import affine_device

    T, C, shape = 4, 2, (256, 1024, 1024)
    DK = dict(ls_angle_deg=36.17, px_to_scan_ratio=0.371, keep_overhang=True, average_n_slices=3)
    g = torch.Generator(device=gpu).manual_seed(0xB1A0)
    base = torch.rand(shape, generator=g, device=gpu) * 400 + 100
    zz = torch.randint(8, shape[0] - 8, (2048,), generator=g, device=gpu)
    yy = torch.randint(8, shape[1] - 8, (2048,), generator=g, device=gpu)
    xx = torch.randint(8, shape[2] - 8, (2048,), generator=g, device=gpu)
    base.index_put_((zz, yy, xx), torch.full((2048,), 3000.0, device=gpu), accumulate=True)
    psf = torch.from_numpy(O.gaussian_psf((9, 7, 7), (2.0, 1.2, 1.2))).to(gpu)
    drift = [(0, 0, 0), (1, -2, 3), (2, -4, 6), (3, -6, 9)]      # voxels per time point, (z, y, x) of the raw volume
    ref = {}
    for t in range(T):
        for c in range(C):
            raw = (torch.roll(base, drift[t], (0, 1, 2)) + 20 * c).round_().to(torch.uint16)
            # deskew: slab independence against the oracle (fill 0), then the production settings (fill mean)
            if t == 1 and c == 0:
                x0, x1 = 500, 508
                slab = fast_deskew_zyx(raw[:, :, x0:x1].contiguous(), overhang_fill=0, **DK)
                full0 = fast_deskew_zyx(raw, overhang_fill=0, **DK)
                X = shape[2]
                assert torch.equal(full0[:, X - x1:X - x0, :], slab)  # raw x column -> deskewed y row X - 1 - x
                want = O.fast_deskew_zyx(raw[:, :, x0:x1].cpu().numpy().astype(np.float32), 36.17, 0.371, True, 3, 0)
                assert rel_err(slab.cpu().numpy(), want) <= 1e-5
                del full0, slab
            dsk = fast_deskew_zyx(raw, overhang_fill="mean", **DK)
            assert dsk.dtype == torch.float32 and dsk.shape[1] == shape[2] and float(dsk.min()) > 0  # overhang filled
            # Richardson-Lucy on the deskewed (awkward-shaped) volume: non-negative, flux kept
            rl = richardson_lucy(dsk, psf, 10, 1e-6)
            assert float(rl.min()) >= 0
            f_in, f_out = float(dsk.sum(dtype=torch.float64)), float(rl.sum(dtype=torch.float64))
            assert abs(f_out - f_in) <= 2e-4 * f_in, (f_in, f_out)
            # drift against t = 0 on the raw volumes, exactly the planted shift (estimate_stabilization.py:259-310)
            rawf = raw.to(torch.float32)
            if t == 0:
                ref[c] = rawf
                m = np.eye(4)
            else:
                sh, _ = phase_cross_corr_device(ref[c], rawf, "magnitude", want_corr=False)
                assert tuple(float(s) for s in sh) == tuple(float(-d) for d in drift[t]), (sh, drift[t])
                m = np.eye(4)
                m[:3, 3] = (0.25 * t, -1.5 * t, 2.25 * t)  # a fractional transform in deskewed space
            stab = affine_device(rl, m, tuple(rl.shape), "linear")
            if t == 2 and c == 1:  # stabilize is local: a crop and its halo against the oracle
                z0, y0, x0, n = 100, 300, 700, 24
                fl = np.floor(m[:3, 3]).astype(int)
                sub = rl[z0 + fl[0]:z0 + fl[0] + n + 2, y0 + fl[1]:y0 + fl[1] + n + 2, x0 + fl[2]:x0 + fl[2] + n + 2].cpu().numpy()
                mm = np.eye(4)
                mm[:3, 3] = m[:3, 3] - fl
                want = O.apply_affine_transform(sub, mm, sub.shape)[:n, :n, :n]
                got = stab[z0:z0 + n, y0:y0 + n, x0:x0 + n].cpu().numpy()
                assert rel_err(got, want) <= 1e-5
            del raw, dsk, rl, stab, rawf


# ----------------------------------------------------------------------------- register-stage column passes (csrc/fftconv_colw.inc)
@pytest.mark.parametrize("shape,pshape", [
    ((256, 64, 128), (9, 5, 5)),      # Z = 256: radix 16 x 16, one exchange
    ((512, 32, 64), (7, 3, 5)),       # Z = 512: 16 x 16 x 2
    ((1024, 32, 64), (11, 3, 3)),     # Z = 1024: 16 x 16 x 4
    ((4, 512, 64), (3, 9, 5)),        # Y/2 = 256
    ((8, 1024, 128), (3, 7, 5)),      # Y/2 = 512
    ((4, 2048, 64), (3, 11, 3)),      # Y/2 = 1024
    ((256, 512, 1024), (5, 5, 5)),    # both column kernels + the wave-private X passes; a ragged last column tile (528 = 8 x 64 + 16)
])
def test_register_stage_column_passes(gpu, shape, pshape, monkeypatch):
    """Columns of 256 / 512 / 1024 points run the register-stage column kernels (three register stages, two LDS exchanges;
    the Z pass multiplies by the OTF between its forward and inverse halves without touching LDS).  Richardson-Lucy
    (convolution, correlation, plain forward / inverse Y passes), Tikhonov (real filter), the inverse filter with bfloat16
    storage and phase cross-correlation agree with the oracle and with the LDS-stepped kernels (BH_FC_COLW=0)."""
    from biahub_amd.apply_inverse_transfer_function import apply_inverse_transfer_function_zyx
    from biahub_amd.deconvolve import compute_tranfser_function, deconvolve, richardson_lucy
    from biahub_amd.estimate_stabilization import phase_cross_corr

    big = int(np.prod(shape)) > 1 << 24
    vol = O.synthetic_volume(shape, seed=33, n_blobs=16)
    vol[0, :, :] += 300.0
    vol[:, -1, :] += 200.0
    psf = O.gaussian_psf(pshape, tuple(max(p / 4.0, 0.8) for p in pshape))
    psf[0, 0, 0] += 0.02
    v, pt = torch.from_numpy(vol).to(gpu), torch.from_numpy(psf).to(gpu)
    its = 2 if big else 3
    monkeypatch.setenv("BH_FC_COLW", "1")   # both axes on the register-stage kernels (the default keeps Z on the LDS-stepped ones)
    new = richardson_lucy(v, pt, its, 1e-6).cpu().numpy()
    monkeypatch.setenv("BH_FC_COLW", "0")
    old = richardson_lucy(v, pt, its, 1e-6).cpu().numpy()
    monkeypatch.setenv("BH_FC_COLW", "1")
    assert rel_err(new, old) <= 2e-5, rel_err(new, old)
    if not big:
        want = O.richardson_lucy_zyx(vol, psf, iterations=its, eps=1e-6)
        assert rel_err(new, want) <= FFT_TOL, rel_err(new, want)
    tf = compute_tranfser_function(psf, shape)
    got = deconvolve(vol[None], transfer_function=tf, regularization_strength=1e-2)[0]
    monkeypatch.setenv("BH_FC_COLW", "0")
    got_old = deconvolve(vol[None], transfer_function=tf, regularization_strength=1e-2)[0]
    g16_old = apply_inverse_transfer_function_zyx(v, tf, 0, 1e-2, False, "bf16").cpu().numpy()
    monkeypatch.setenv("BH_FC_COLW", "1")
    assert rel_err(got, got_old) <= 2e-5
    g16 = apply_inverse_transfer_function_zyx(v, tf, 0, 1e-2, False, "bf16").cpu().numpy()
    assert rel_err(g16, g16_old) <= 2e-5   # the same bfloat16 filter through both kernel families
    if not big:
        assert rel_err(got, O.deconvolve_czyx(vol[None], tf, 1e-2)[0]) <= FFT_TOL
    mov = np.roll(vol, (1, -5, 9), axis=(0, 1, 2))
    shift, _ = phase_cross_corr(vol, mov, normalization="magnitude")
    assert tuple(float(s) for s in shift) == (-1.0, 5.0, -9.0)


@pytest.mark.parametrize("shape,pshape", [((512, 32, 64), (7, 3, 5)), ((512, 64, 192), (9, 5, 5))])
def test_radix8_z_pass(gpu, shape, pshape, monkeypatch):
    """512-point Z passes with a spectral product run radix-8 register stages (csrc/fftconv_colz.inc: four LDS round trips).
    Every mode — real transfer function (symmetric PSF, Tikhonov), complex convolution / correlation (asymmetric PSF), the
    complex inverse filter in float32 and bfloat16, the phase-correlation product — agrees with the oracle and with the radix-4
    kernel (BH_FC_COLZ=0) on the same inputs, including a ragged last column tile."""
    from biahub_amd.apply_inverse_transfer_function import apply_inverse_transfer_function_zyx
    from biahub_amd.deconvolve import compute_tranfser_function, deconvolve, richardson_lucy
    from biahub_amd.estimate_stabilization import phase_cross_corr

    vol = O.synthetic_volume(shape, seed=57, n_blobs=16)
    vol[3, :, :] += 250.0
    v = torch.from_numpy(vol).to(gpu)
    sym = O.gaussian_psf(pshape, tuple(max(q / 4.0, 0.8) for q in pshape))
    asym = sym.copy()
    asym[0, 0, 0] += 0.02
    rng = np.random.default_rng(5)
    tfc = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64) * 0.3 + 1.0
    mov = np.roll(vol, (2, -3, 7), axis=(0, 1, 2))

    def run():
        out = {}
        for name, psf in (("rl_real", sym), ("rl_complex", asym)):
            out[name] = richardson_lucy(v, torch.from_numpy(psf).to(gpu), 3, 1e-6).cpu().numpy()
        tf = compute_tranfser_function(sym, shape)
        out["tikhonov"] = deconvolve(vol[None], transfer_function=tf.copy(), regularization_strength=1e-2)[0]
        out["inv_f32"] = apply_inverse_transfer_function_zyx(v, tfc, 0, 1e-2, False, "f32").cpu().numpy()
        out["inv_bf16"] = apply_inverse_transfer_function_zyx(v, tfc, 0, 1e-2, False, "bf16").cpu().numpy()
        for norm in (None, "magnitude"):
            sh, corr = phase_cross_corr(vol, mov, normalization=norm)
            assert tuple(float(x) for x in sh) == (-2.0, 3.0, -7.0)
            out[f"pcc_{norm}"] = corr
        return out

    monkeypatch.delenv("BH_FC_COLZ", raising=False)
    new = run()
    monkeypatch.setenv("BH_FC_COLZ", "0")
    old = run()
    monkeypatch.delenv("BH_FC_COLZ", raising=False)
    for k in new:
        assert rel_err(new[k], old[k]) <= 2e-5, (k, rel_err(new[k], old[k]))
    assert rel_err(new["rl_real"], O.richardson_lucy_zyx(vol, sym, iterations=3, eps=1e-6)) <= FFT_TOL
    assert rel_err(new["rl_complex"], O.richardson_lucy_zyx(vol, asym, iterations=3, eps=1e-6)) <= FFT_TOL
    assert rel_err(new["tikhonov"], O.deconvolve_czyx(vol[None], compute_tranfser_function(sym, shape), 1e-2)[0]) <= FFT_TOL
    assert rel_err(new["pcc_magnitude"], O.phase_cross_corr(vol, mov, "magnitude")[1]) <= 1e-3


@pytest.mark.parametrize("shape,pshape", [((32, 64, 128), (9, 7, 5)), ((256, 64, 1024), (5, 5, 9))])
def test_richardson_lucy_real_otf_of_symmetric_psf(gpu, shape, pshape, monkeypatch):
    """A PSF with odd extents that equals its point mirror has a real transfer function: the Z passes then read one float
    per bin and run convolution and correlation as the same real product.  Same result as the general complex path
    (BH_RL_COMPLEX_OTF=1) and as the oracle; a PSF that is not symmetric (or has an even extent) keeps the complex path."""
    from biahub_amd.deconvolve import richardson_lucy

    vol = O.synthetic_volume(shape, seed=41, n_blobs=12)
    psf = O.gaussian_psf(pshape, tuple(p / 4.0 for p in pshape))
    assert np.array_equal(psf, psf[::-1, ::-1, ::-1])
    v, pt = torch.from_numpy(vol).to(gpu), torch.from_numpy(psf).to(gpu)
    real = richardson_lucy(v, pt, 5, 1e-6).cpu().numpy()
    monkeypatch.setenv("BH_RL_COMPLEX_OTF", "1")
    cplx = richardson_lucy(v, pt, 5, 1e-6).cpu().numpy()
    monkeypatch.delenv("BH_RL_COMPLEX_OTF")
    assert rel_err(real, cplx) <= 2e-6, rel_err(real, cplx)
    assert rel_err(real, O.richardson_lucy_zyx(vol, psf, 5, 1e-6)) <= FFT_TOL
    # asymmetric PSF right after a symmetric one of the same shape: the cache must not hand out the real form
    psf2 = psf.copy()
    psf2[0, 1, 2] *= 1.5
    got = richardson_lucy(v, torch.from_numpy(psf2).to(gpu), 5, 1e-6).cpu().numpy()
    assert rel_err(got, O.richardson_lucy_zyx(vol, psf2, 5, 1e-6)) <= FFT_TOL
    again = richardson_lucy(v, pt, 5, 1e-6).cpu().numpy()
    assert np.array_equal(again, real)


def test_overlapped_pipeline_equals_serial(gpu):
    """biahub_amd.pipeline.run_overlapped: results of the three-stream pipeline (upload i + 1 / compute i / download i - 1)
    equal the serial chain, in order, for more units than pipeline stages and for 0, 1 and 2 units."""
    from biahub_amd.deconvolve import richardson_lucy
    from biahub_amd.deskew import fast_deskew_zyx
    from biahub_amd.pipeline import run_overlapped

    rng = np.random.default_rng(3)
    psf = torch.from_numpy(O.gaussian_psf((5, 5, 5), (1.0, 1.0, 1.0))).to(gpu)
    vols = [torch.from_numpy(rng.integers(100, 4000, (32, 64, 96)).astype(np.uint16)).pin_memory() for _ in range(5)]

    def compute(d):
        return fast_deskew_zyx(richardson_lucy(d, psf, 3, 1e-6), ls_angle_deg=36.0, px_to_scan_ratio=0.4, keep_overhang=False, average_n_slices=1)

    want = [compute(v.to(gpu)).cpu() for v in vols]
    for n in (0, 1, 2, 5):
        landing = [torch.empty(want[0].shape, dtype=torch.float32, pin_memory=True) for _ in range(n)]
        k = [0]

        def download(t):
            buf = landing[k[0]]
            k[0] += 1
            buf.copy_(t, non_blocking=True)
            return buf

        got = list(run_overlapped(vols[:n], lambda h: h.to(gpu, non_blocking=True), compute, download, gpu))
        assert len(got) == n
        for g, w in zip(got, want):
            assert torch.equal(g, w)


# ----------------------------------------------------------------------------- prepared Richardson-Lucy handle
@pytest.mark.parametrize("shape,pshape,backend", [
    ((16, 64, 128), (5, 5, 7), "engine"),            # the fused engine at the volume's own shape
    ((8, 64, 1024), (3, 3, 9), "engine"),            # wave-private X passes
    ((21, 64, 150), (7, 5, 9), "engine-padded"),     # wrap-padded engine box
    ((15, 21, 25), (5, 3, 3), "library"),            # hipFFT
])
def test_prepared_richardson_lucy_handle(gpu, shape, pshape, backend, monkeypatch):
    """bh_richardson_lucy_create / _apply / _destroy: the transfer function is built once, every apply agrees with the
    one-shot entry and the oracle — for a point-symmetric PSF (real transfer function kept) and an asymmetric one, for
    several volumes and iteration counts through one handle, in place, and with iterations = 0."""
    from biahub_amd.deconvolve import PreparedRichardsonLucy, richardson_lucy, richardson_lucy_czyx

    sym = O.gaussian_psf(pshape, tuple(max(p / 4.0, 0.8) for p in pshape))
    asym = sym.copy()
    asym[0, 0, 0] += 0.02
    vols = [O.synthetic_volume(shape, seed=s, n_blobs=8) for s in (31, 32)]
    vols[1][0, :, 0] += 300.0
    for psf, want_real in ((sym, backend != "library"), (asym, False)):
        with PreparedRichardsonLucy(psf, shape, gpu) as h:
            assert h.backend == backend and h.otf_is_real == want_real
            for vol, it in ((vols[0], 4), (vols[1], 2), (vols[0], 4)):
                want = O.richardson_lucy_zyx(vol, psf, iterations=it, eps=1e-6)
                v = torch.from_numpy(vol).to(gpu)
                got = h(v, it, 1e-6)
                assert rel_err(got.cpu().numpy(), want) <= FFT_TOL, (backend, it)
                one = richardson_lucy(v, torch.from_numpy(psf).to(gpu), it, 1e-6)
                assert rel_err(got.cpu().numpy(), one.cpu().numpy()) <= 2e-5
            inplace = torch.from_numpy(vols[1]).to(gpu)
            assert h(inplace, 2, 1e-6, out=inplace) is inplace                      # in == out: the data term is kept aside
            assert rel_err(inplace.cpu().numpy(), O.richardson_lucy_zyx(vols[1], psf, iterations=2, eps=1e-6)) <= FFT_TOL
            assert np.array_equal(h(torch.from_numpy(vols[0]).to(gpu), 0).cpu().numpy(), np.maximum(vols[0], 0))
            with pytest.raises(ValueError):
                h(torch.zeros((4, 32, 64), device=gpu), 1)
    # the numpy operator adapter keeps one handle per PSF object across its calls
    c = richardson_lucy_czyx(np.stack(vols), sym, iterations=3)
    assert rel_err(c[1], O.richardson_lucy_zyx(vols[1], sym, iterations=3, eps=1e-6)) <= FFT_TOL


# ----------------------------------------------------------------------------- oracle parity at the bench size
def test_config2_full_size_oracle_parity(gpu):
    """BASELINE config 2 at its own size against the oracle — the exact kernel instantiations bench.py runs
    (colz_kernel<5>, colw_kernel<10, .>, xw_kernel<10, 4 | 5>, deskew_kernel<float, ...>), not small stand-ins:
    Richardson-Lucy, 2 iterations on (512, 2048, 2048) (scipy.fft on all host cores, about a minute), <= 1e-4; deskew of that
    estimate against the oracle on three x-slabs (first, middle, the ragged last columns), <= 1e-5, with fill 0 and with
    fill "mean" (the fill value read from the full HIP result, whose statistics the slab oracle cannot know)."""
    from biahub_amd.deconvolve import PreparedRichardsonLucy
    from biahub_amd.deskew import fast_deskew_zyx

    shape = (512, 2048, 2048)
    g = torch.Generator(device=gpu).manual_seed(23)
    vol_d = (torch.rand(shape, generator=g, device=gpu) * 300 + 100).round_()
    zz = torch.randint(4, shape[0] - 4, (4096,), generator=g, device=gpu)
    yy = torch.randint(4, shape[1] - 4, (4096,), generator=g, device=gpu)
    xx = torch.randint(4, shape[2] - 4, (4096,), generator=g, device=gpu)
    vol_d.index_put_((zz, yy, xx), torch.full((4096,), 3000.0, device=gpu), accumulate=True)    # beads on a noisy background
    psf = O.gaussian_psf((33, 17, 17), (3.0, 1.5, 1.5))
    with PreparedRichardsonLucy(psf, shape, gpu) as h:
        assert h.backend == "engine" and h.otf_is_real
        est_d = h(vol_d, 2, 1e-6)
    vol = vol_d.cpu().numpy()
    del vol_d
    want = O.richardson_lucy_zyx(vol, psf, iterations=2, eps=1e-6)
    del vol
    est = est_d.cpu().numpy()
    scale = float(np.abs(want).max())
    err = 0.0
    for z0 in range(0, shape[0], 64):   # blockwise: no 8-GB temporaries
        err = max(err, float(np.abs(est[z0:z0 + 64] - want[z0:z0 + 64]).max()))
    assert err / scale <= FFT_TOL, err / scale
    del want

    kw = dict(ls_angle_deg=36.17, px_to_scan_ratio=0.371, keep_overhang=True, average_n_slices=3)
    X = shape[2]
    zero = fast_deskew_zyx(est_d, overhang_fill=0, **kw).cpu().numpy()
    mean = fast_deskew_zyx(est_d, overhang_fill="mean", **kw).cpu().numpy()
    del est_d
    # the fill value the kernel used: voxels deep in the two overhang wedges (exact zeros before the fill) — the last deskewed
    # plane at small x', the first at large x' (ix = r x' - r cos(theta) z' + offset leaves [0, Z) there)
    assert zero[-1, 0, 0] == 0.0 and zero[0, -1, -1] == 0.0
    fill = float(mean[-1, 0, 0])
    assert fill > 0 and float(mean[0, -1, -1]) == fill
    dscale = float(np.abs(zero).max())
    for x0, x1 in ((0, 48), (1000, 1064), (X - 37, X)):
        slab = est[:, :, x0:x1]
        rows = slice(X - x1, X - x0)                                # out[a, yo, xo] reads in[:, :, X - 1 - yo]
        w0 = O.fast_deskew_zyx(slab, overhang_fill=0, **kw)
        assert np.abs(zero[:, rows] - w0).max() <= DESKEW_TOL * dscale, (x0, x1)
        w1 = O.fast_deskew_zyx(slab, overhang_fill=fill, **kw)
        # the 26-connected dilation of the zero mask crosses the slab's y faces: compare the rows it cannot reach from outside
        inner = slice(3, (x1 - x0) - 3)
        got = mean[:, rows][:, inner]
        assert np.abs(got - w1[:, inner]).max() <= DESKEW_TOL * max(dscale, fill), (x0, x1)


# ----------------------------------------------------------------------------- wrap-padded R-L without a fold pass
@pytest.mark.parametrize("shape,pshape", [
    ((21, 64, 1500), (7, 5, 9)),     # box (32, 64, 1536): z and x padded, two row pairs per wavefront
    ((10, 32, 3000), (3, 3, 8)),     # box (16, 32, 3072): even PSF extent along x — the margins differ by one (hi = lo + 1)
    ((16, 32, 1500), (5, 3, 9)),     # z a power of two: only x is padded
    ((44, 64, 1536), (9, 5, 5)),     # x native (3 * 512), z -> 64
    ((30, 128, 1517), (4, 5, 17)),   # even extent along z; the deskewed row length of BASELINE config 4
    ((21, 64, 1000), (7, 5, 9)),     # box (32, 64, 1024): power-of-two rows, the wrap modes of csrc/fftconv_xw.inc (two pairs per wave)
    ((10, 32, 2000), (3, 3, 8)),     # box (16, 32, 2048): one pair per wavefront, even extent along x
    ((12, 32, 500), (5, 3, 7)),      # box (16, 32, 512): four pairs per wavefront
    ((40, 256, 1500), (5, 5, 9)),    # box (48, 256, 1536): 6144 row pairs — every wavefront walks several pairs (prefetch chain)
    ((40, 256, 1000), (5, 5, 9)),    # box (48, 256, 1024): the same for the power-of-two kernels
    ((12, 32, 1100), (5, 3, 9)),     # 1108 columns needed: box (16, 32, 1536), not the 1280-voxel rows of the tile kernels
    ((10, 32, 2300), (3, 3, 7)),     # 2306 needed: box (16, 32, 3072) rather than 2560
])
def test_richardson_lucy_wrap_padded_box(gpu, shape, pshape, monkeypatch):
    """Rows of 1536 / 3072 voxels at a wrap-padded box run Richardson-Lucy in the 8 passes of the unpadded path: estimate and
    ratio are wrap-extended by the fused X passes themselves (along x inside the row, along z by re-reading the mirrored
    plane, out of place), so the correlation needs no fold (fftconv_richardson_lucy_wrap).  Against the oracle, the fold /
    rewrap path (BH_RL_NOWRAP=1) and the library path."""
    from biahub_amd.deconvolve import PreparedRichardsonLucy, richardson_lucy

    vol = O.synthetic_volume(shape, seed=41, n_blobs=10)
    vol[0, 0, :] += 500.0            # structure on every face: a wrong wrap shows up as a wrap-around error
    vol[-1, :, -1] += 300.0
    vol[:, -1, 0] += 200.0
    psf = O.gaussian_psf(pshape, tuple(max(p / 4.0, 0.8) for p in pshape))
    psf[0, 0, 0] += 0.02             # asymmetric: convolution and correlation differ
    v, pt = torch.from_numpy(vol).to(gpu), torch.from_numpy(psf).to(gpu)
    for it in (1, 2, 5):
        want = O.richardson_lucy_zyx(vol, psf, iterations=it, eps=1e-6)
        with PreparedRichardsonLucy(psf, shape, gpu) as h:
            assert h.backend == "engine-padded"
            got = h(v, it, 1e-6).cpu().numpy()
        assert rel_err(got, want) <= FFT_TOL, (it, rel_err(got, want))
        monkeypatch.setenv("BH_RL_NOWRAP", "1")
        fold = richardson_lucy(v, pt, it, 1e-6).cpu().numpy()
        monkeypatch.delenv("BH_RL_NOWRAP")
        assert rel_err(got, fold) <= 2e-5, (it, rel_err(got, fold))
    sym = O.gaussian_psf(pshape, tuple(max(p / 4.0, 0.8) for p in pshape))
    if all(k % 2 for k in pshape):   # point-symmetric PSF: the real transfer function in the wrap path
        with PreparedRichardsonLucy(sym, shape, gpu) as h:
            assert h.otf_is_real
            got = h(v, 3, 1e-6).cpu().numpy()
        assert rel_err(got, O.richardson_lucy_zyx(vol, sym, iterations=3, eps=1e-6)) <= FFT_TOL


@pytest.mark.parametrize("shape,pshape", [((384, 32, 64), (7, 3, 5)), ((384, 64, 160), (9, 5, 5)), ((350, 32, 1500), (9, 3, 9)),
                                          ((768, 32, 64), (7, 3, 5)), ((768, 32, 72), (5, 5, 3)), ((700, 32, 1500), (9, 3, 9))])
def test_radix3_register_z_pass(gpu, shape, pshape, monkeypatch):
    """384- and 768-point Z passes (3 x 128, 3 x 256: the boxes of the deskewed BASELINE config-4 / config-2 volumes) with a spectral product run register
    stages (csrc/fftconv_colz3.inc: radix-3 + radix-2 in registers, two radix-8 steps, four LDS round trips).  Real transfer
    function (symmetric PSF) and complex convolution / correlation (asymmetric PSF) agree with the oracle and with the LDS-step
    kernel (BH_FC_COLZ3=0) on the same inputs, including a ragged last column tile and the wrap-padded path."""
    from biahub_amd.deconvolve import richardson_lucy

    vol = O.synthetic_volume(shape, seed=61, n_blobs=16)
    vol[3, :, :] += 250.0
    vol[-1, 0, :] += 100.0
    v = torch.from_numpy(vol).to(gpu)
    sym = O.gaussian_psf(pshape, tuple(max(q / 4.0, 0.8) for q in pshape))
    asym = sym.copy()
    asym[0, 0, 0] += 0.02
    for name, psf in (("real", sym), ("complex", asym)):
        pt = torch.from_numpy(psf).to(gpu)
        monkeypatch.delenv("BH_FC_COLZ3", raising=False)
        new = richardson_lucy(v, pt, 3, 1e-6).cpu().numpy()
        monkeypatch.setenv("BH_FC_COLZ3", "0")
        old = richardson_lucy(v, pt, 3, 1e-6).cpu().numpy()
        monkeypatch.delenv("BH_FC_COLZ3")
        assert rel_err(new, old) <= 2e-5, (name, rel_err(new, old))
        assert rel_err(new, O.richardson_lucy_zyx(vol, psf, iterations=3, eps=1e-6)) <= FFT_TOL, name


def test_cubic_warp_full_size_properties(gpu):
    """The cubic warp at the bench size (512, 2048, 2048) through size-independent properties (the float64 oracle needs minutes
    there): an integer shift returns the shifted samples (prefilter and B-spline sampling are inverses on the grid) with cval
    where the source leaves the volume; the warp is linear in the volume; the z-uniform plane-combining path and the general
    64-tap path agree; a slab of the rotated result matches the oracle run on the slab's own source rows."""
    from biahub_amd import _lib
    from biahub_amd.register import affine_device

    shape = (512, 2048, 2048)
    g = torch.Generator(device=gpu).manual_seed(5)
    vol = torch.empty(shape, device=gpu).uniform_(10.0, 500.0, generator=g)
    S = np.eye(4)
    S[:3, 3] = (3.0, -5.0, 7.0)
    out = affine_device(vol, S, shape, "cubic", _lib.BOUNDARY_SCIPY_CONSTANT, -1.0)
    assert float((out[:-3, 5:, :-7] - vol[3:, :-5, 7:]).abs().max()) <= 2e-6 * 500.0 * 4  # |coefficients| <= ~1.5 max, 64 float32 taps
    assert bool((out[-3:] == -1.0).all()) and bool((out[:, :5] == -1.0).all()) and bool((out[:, :, -7:] == -1.0).all())
    del out
    th = np.deg2rad(2.0)
    c, s_ = np.cos(th), np.sin(th)
    M = np.eye(4)
    M[1, 1], M[1, 2], M[2, 1], M[2, 2] = c, -s_, s_, c
    ctr = np.array(shape) / 2
    M[:3, 3] = ctr - M[:3, :3] @ ctr + np.array([1.5, -3.25, 2.75])
    a = affine_device(vol, M, shape, "cubic", _lib.BOUNDARY_SCIPY_CONSTANT, 0.0)
    import os

    os.environ["BH_SPLINE_ZUNI"] = "0"
    try:
        b = affine_device(vol, M, shape, "cubic", _lib.BOUNDARY_SCIPY_CONSTANT, 0.0)
    finally:
        del os.environ["BH_SPLINE_ZUNI"]
    assert float((a - b).abs().max()) <= 3e-6 * float(b.abs().max())
    del b
    # linearity: warp(2 v + 3) = 2 warp(v) + 3 inside the volume (cval 0 outside scales the same way only without the offset)
    lin = affine_device(vol * 2.0 + 3.0, M, shape, "cubic", _lib.BOUNDARY_SCIPY_CONSTANT, 3.0)
    inside = a != 0.0
    assert float(((lin - (a * 2.0 + 3.0)) * inside).abs().max()) <= 1e-5 * float(lin.abs().max())
    del lin, inside
    # a slab against the oracle: output planes 200..203 need source planes 198..206 only (z shift 1.5, no z coupling); the
    # z prefilter of the slab's own 9 planes is not the volume's, so compare through a volume that is constant along z instead
    col = vol[:1].expand(24, -1, -1).contiguous()  # 24 identical planes: the z prefilter and the z taps reproduce the plane
    got = affine_device(col, M, (24, 2048, 2048), "cubic", _lib.BOUNDARY_SCIPY_CONSTANT, 0.0)[8, 1000:1016, 900:1100].cpu().numpy()
    plane = vol[0].cpu().numpy()
    want = O.spline_affine_pull(plane, np.array([[c, -s_, M[1, 3]], [s_, c, M[2, 3]], [0, 0, 1.0]]), (2048, 2048), 0.0)[1000:1016, 900:1100]
    assert rel_err(got, want) <= 1e-5


def test_lz4_full_size_round_trip(gpu):
    """The device codec at BASELINE sizes: a (512, 2048, 2048) uint16 camera-like stack (4.3 GB) and a (342, 1024, 1517) float32
    deskewed-like volume (2.1 GB) -> bit shuffle -> LZ4 + Blosc frames on the GPU -> every frame decoded on the GPU in one launch
    -> un-shuffled: the bytes come back exactly; the frames shrink; one frame of each also goes through the host decoder."""
    from biahub_amd import codecs

    g = torch.Generator(device=gpu).manual_seed(9)
    for shape, dt, zc in (((512, 2048, 2048), torch.uint16, 32), ((342, 1024, 1517), torch.float32, 32)):
        Z, Y, X = shape
        if dt == torch.uint16:
            vol = (torch.empty(shape, device=gpu).normal_(110.0, 4.0, generator=g).clamp_(0, 65535)
                   + 200.0 * torch.sin(torch.arange(X, device=gpu) / 40.0).abs()).to(torch.uint16)
        else:
            vol = (torch.empty(shape, device=gpu).normal_(110.0, 4.0, generator=g) * 8.0).round_() / 8.0  # 1/8-count steps
        ts = vol.element_size()
        nch = -(-Z // zc)
        cbytes = zc * Y * X * ts
        bsz = codecs.default_blocksize(ts)
        v8 = vol.view(torch.uint8).reshape(-1)
        raw = torch.zeros(nch * cbytes, dtype=torch.uint8, device=gpu)   # the last chunk overhangs: zero planes behind the volume
        raw[: v8.numel()] = v8
        stage = torch.empty_like(raw)
        for i in range(nch):
            codecs.filter_device(raw[i * cbytes:(i + 1) * cbytes], stage[i * cbytes:(i + 1) * cbytes], bsz, ts, codecs.BLOSC_BITSHUFFLE)
        packed, offs = codecs.blosc_lz4_compress_device(stage, nch, cbytes, bsz, ts, codecs.BLOSC_BITSHUFFLE)
        assert offs[-1] < 0.6 * raw.numel(), (shape, offs[-1] / raw.numel())
        host = packed[: offs[-1]].cpu().numpy()
        frames = [host[offs[i]: offs[i + 1]].tobytes() for i in range(nch)]
        back = torch.empty_like(stage)
        heads = codecs.blosc_lz4_decode_frames_device(frames, back, [i * cbytes for i in range(nch)])
        assert len(heads) == nch and torch.equal(back, stage)
        out = torch.empty_like(raw)
        for i in range(nch):
            codecs.unfilter_device(back[i * cbytes:(i + 1) * cbytes], out[i * cbytes:(i + 1) * cbytes], bsz, ts, codecs.BLOSC_BITSHUFFLE)
        assert torch.equal(out, raw)
        k = nch // 2
        assert np.array_equal(codecs.blosc_decompress(frames[k][: heads[k].cbytes]), raw[k * cbytes:(k + 1) * cbytes].cpu().numpy())
        del vol, raw, stage, packed, back, out
