"""Seeded randomised parity sweeps (GPU): shapes, dtypes and parameters the hand-picked cases do not enumerate."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import oracle_np as O

pytestmark = pytest.mark.gpu


def test_fuzz_fused_engine_vs_oracle(gpu):
    """Fused FFT engine on random shapes it takes as they are — axes of 2^k, 3 * 2^k and 5 * 2^k — and PSF extents (even, odd,
    1) against the NumPy oracle: R-L, Tikhonov, phase cross-correlation."""
    from biahub_amd.deconvolve import richardson_lucy, richardson_lucy_plan, tikhonov_zyx, transfer_function_device
    from biahub_amd.estimate_stabilization import phase_cross_corr_device

    rng = np.random.default_rng(123)
    done = 0
    odd = 0
    while done < 20:
        Z, Y, X = int(2 ** rng.integers(2, 7)), int(2 ** rng.integers(5, 9)), int(2 ** rng.integers(6, 11))
        if done >= 10:  # the second half: an odd first step (3 or 5) on one to three axes
            fz, fy, fx = (int(rng.choice([1, 3, 5])) for _ in range(3))
            Z, Y, X = max(8, Z) * fz // (2 if fz > 1 else 1), max(64, Y) * fy // (2 if fy > 1 else 1), max(128, X) * fx // (4 if fx > 1 else 1)
        if Z * Y * X > 2**22:
            continue
        if richardson_lucy_plan((1, 1, 1), (Z, Y, X))[1] != "engine":
            continue
        odd += any(n & (n - 1) for n in (Z, Y, X))
        done += 1
        pshape = tuple(int(min(rng.integers(1, 12), n)) for n in (Z, Y, X))
        volh = (rng.random((Z, Y, X)) * 300).astype(np.float32)
        psfh = (rng.random(pshape) + 0.05).astype(np.float32)
        vol, psf = torch.from_numpy(volh).to(gpu), torch.from_numpy(psfh).to(gpu)
        got = richardson_lucy(vol, psf, 3, 1e-6).cpu().numpy()
        assert rel_err(got, O.richardson_lucy_zyx(volh, psfh, iterations=3, eps=1e-6)) <= 1e-4, ((Z, Y, X), pshape)
        tf = O.compute_transfer_function(psfh, (Z, Y, X))
        got = tikhonov_zyx(vol, torch.from_numpy(tf).to(gpu), 1e-2).cpu().numpy()
        assert rel_err(got, O.tikhonov_zyx(volh, tf, 1e-2)) <= 1e-4, ((Z, Y, X), pshape)
        assert rel_err(transfer_function_device(psf, (Z, Y, X), gpu).cpu().numpy(), tf) <= 1e-5
        roll = tuple(int(v) for v in rng.integers(-3, 4, 3))
        movh = np.roll(volh, roll, axis=(0, 1, 2))
        sh, corr = phase_cross_corr_device(vol, torch.from_numpy(movh).to(gpu), "magnitude")
        wsh, wcorr = O.phase_cross_corr(volh, movh, "magnitude")
        assert np.array_equal(sh, wsh) and rel_err(corr.cpu().numpy(), wcorr) <= 1e-4, ((Z, Y, X), roll)
    assert odd >= 6


def test_fuzz_richardson_lucy_backends_vs_oracle(gpu, monkeypatch):
    """Random shapes (awkward, 7-smooth, 3 * 2^k, powers of two) and PSF extents: whatever back-end the cost model picks, and
    the wrap-padded engine box and the library box when forced, all give the oracle's circular Richardson-Lucy."""
    from biahub_amd.deconvolve import richardson_lucy, richardson_lucy_plan

    rng = np.random.default_rng(2024)
    pools = ([11, 13, 17, 19, 23, 29, 31, 37, 41, 53], [15, 21, 28, 30, 36, 42, 45, 50, 60], [24, 48, 96], [16, 32, 64])
    seen = set()
    for trial in range(18):
        dims = []
        for a in range(3):
            pool = pools[int(rng.integers(0, 4))]
            n = int(pool[int(rng.integers(0, len(pool)))])
            dims.append(n * (2 if a == 1 else 1) * (2 if a == 2 and n < 40 else 1))  # y >= 22, x a little longer
        shape = tuple(dims) if trial < 16 else ((24, 64, 64), (32, 96, 128))[trial - 16]  # two the engine takes as they are
        pshape = tuple(int(min(rng.integers(1, 8), n // 2)) for n in shape)
        volh = (rng.random(shape) * 300).astype(np.float32)
        volh[0, :, -1] += 500.0
        psfh = (rng.random(pshape) + 0.05).astype(np.float32)
        want = O.richardson_lucy_zyx(volh, psfh, iterations=3, eps=1e-6)
        vol, psf = torch.from_numpy(volh).to(gpu), torch.from_numpy(psfh).to(gpu)
        for force in (None, "1", "0"):
            if force is None:
                monkeypatch.delenv("BH_RL_ENGINE_PAD", raising=False)
            else:
                monkeypatch.setenv("BH_RL_ENGINE_PAD", force)
            box, backend = richardson_lucy_plan(pshape, shape)
            seen.add(backend)
            got = richardson_lucy(vol, psf, 3, 1e-6).cpu().numpy()
            assert rel_err(got, want) <= 1e-4, (shape, pshape, force, box, backend, rel_err(got, want))
    assert seen == {"engine", "engine-padded", "library"}
    from biahub_amd.device import get_context

    # informational: how many 3-D library plans the creation-time self-check had to rebuild in this process
    print("hipFFT plans rebuilt after a failed self-check:", get_context(gpu).fft_plans_replaced())


def test_fuzz_richardson_lucy_wrap_paths_vs_oracle(gpu, monkeypatch):
    """Random volumes whose rows pad to a length the wave-private X passes take (512 ... 3072 voxels) with Y a power of two:
    the wrap-padded box without a fold pass (csrc/fftconv_xw.inc / fftconv_x3.inc WRAP modes, cropped last update) for random
    odd and even PSF extents, a prepared handle reused across two volumes, against the oracle and the fold path."""
    from biahub_amd.deconvolve import PreparedRichardsonLucy, richardson_lucy

    rng = np.random.default_rng(77)
    boxes = set()
    for trial in range(10):
        X = int(rng.choice([rng.integers(400, 505), rng.integers(900, 1010), rng.integers(1300, 1525), rng.integers(1800, 2040),
                            rng.integers(2700, 3060)]))
        Y = int(rng.choice([32, 64]))
        Z = int(rng.integers(5, 30))
        shape = (Z, Y, X)
        pshape = (int(rng.integers(1, min(6, Z // 2 + 1))), int(rng.integers(1, 6)), int(rng.integers(2, 12)))
        psfh = (rng.random(pshape) + 0.05).astype(np.float32)
        with PreparedRichardsonLucy(psfh, shape, gpu) as h:
            if h.backend != "engine-padded":
                continue
            boxes.add(h.box[2])
            for seed in (0, 1):
                volh = (np.random.default_rng(trial * 2 + seed).random(shape) * 300).astype(np.float32)
                volh[0, :, -1] += 500.0
                volh[-1, 0, :] += 250.0
                want = O.richardson_lucy_zyx(volh, psfh, iterations=3, eps=1e-6)
                got = h(torch.from_numpy(volh).to(gpu), 3, 1e-6).cpu().numpy()
                assert rel_err(got, want) <= 1e-4, (shape, pshape, h.box, rel_err(got, want))
        monkeypatch.setenv("BH_RL_NOWRAP", "1")
        fold = richardson_lucy(torch.from_numpy(volh).to(gpu), torch.from_numpy(psfh).to(gpu), 3, 1e-6).cpu().numpy()
        monkeypatch.delenv("BH_RL_NOWRAP")
        assert rel_err(got, fold) <= 3e-5, (shape, pshape)
    assert len(boxes) >= 3, boxes


def test_fuzz_deskew_flatfield_affine_vs_oracle(gpu):
    from biahub_amd.deskew import fast_deskew_zyx
    from biahub_amd.flat_field import flat_field_zyx, median_z_device
    from biahub_amd.register import apply_affine_transform

    rng = np.random.default_rng(77)
    for _ in range(12):
        Z, Y, X = (int(v) for v in rng.integers(3, 40, 3))
        dt = [np.float32, np.uint16, np.uint8, np.int16][int(rng.integers(0, 4))]
        vol = (rng.random((Z, Y, X)) * (200 if dt == np.uint8 else 3000)).astype(dt)
        # deskew: random geometry, both fills
        ang, ratio, N = float(rng.uniform(20, 50)), float(rng.uniform(0.2, 0.8)), int(rng.integers(1, 4))
        fill = ["mean", 0, 55.0][int(rng.integers(0, 3))]
        want = O.fast_deskew_zyx(vol.astype(np.float32), round(ang, 2), round(ratio, 3), True, N, fill)
        got = fast_deskew_zyx(torch.from_numpy(vol).to(gpu), round(ang, 2), round(ratio, 3), True, N, fill).cpu().numpy()
        assert got.shape == want.shape and rel_err(got, want) <= 2e-5, (Z, Y, X, dt, ang, ratio, N, fill)
        # median / flat field
        data = vol if dt != np.float32 else (vol - 1500).astype(np.float32)
        assert np.array_equal(median_z_device(data).cpu().numpy(), np.median(data, axis=0).astype(np.float64))
        if dt == np.uint16:
            w = O.flat_field_zyx(data + 1).astype(np.float32)
            assert np.abs(flat_field_zyx(data + 1) - w).max() <= 2e-7 * np.abs(w).max()
        # affine: random similarity + shear, both interpolations
        A = np.eye(4)
        A[:3, :3] += rng.normal(0, 0.08, (3, 3))
        A[:3, 3] = rng.uniform(-4, 4, 3)
        out_shape = tuple(int(v) for v in rng.integers(3, 40, 3))
        fv = vol.astype(np.float32)
        for interp in ("linear", "nearestneighbor"):
            w = O.apply_affine_transform(fv, A, out_shape, interp)
            g = apply_affine_transform(fv, A, out_shape, interpolation=interp)
            if interp == "linear":
                assert rel_err(g, w) <= 1e-5, (Z, Y, X, out_shape)
            else:
                assert (g != w).mean() <= 1e-3, (Z, Y, X, out_shape)  # float64 ties at half-voxel coordinates


def test_fuzz_round2_engine_kernels_vs_oracle(gpu):
    """Random shapes that land on the round-2 kernels — rows of 1024 / 2048 voxels (wave-private X passes), columns of
    256 / 512 / 1024 points (register-stage column passes) mixed with shorter and 3 * 2^k / 5 * 2^k axes — and random PSFs,
    some of them point-symmetric with odd extents (real transfer function): R-L, Tikhonov and the inverse filter against the
    NumPy oracle."""
    from biahub_amd.apply_inverse_transfer_function import apply_inverse_transfer_function_zyx
    from biahub_amd.deconvolve import richardson_lucy, richardson_lucy_plan, tikhonov_zyx

    rng = np.random.default_rng(20261004)
    zs, ys, xs = [4, 8, 16, 24, 40, 64, 256, 512], [32, 64, 96, 160, 512, 1024, 2048], [1024, 2048]
    forced = [(256, 32, 1024), (512, 32, 1024), (4, 2048, 1024), (8, 1024, 2048), (16, 512, 1024)]  # the column kernels for sure
    done, real_otf, colw = 0, 0, 0
    while done < 12:
        if forced:
            Z, Y, X = forced.pop()
        else:
            Z, Y, X = int(rng.choice(zs)), int(rng.choice(ys)), int(rng.choice(xs))
        if Z * Y * X > 2**24 or richardson_lucy_plan((1, 1, 1), (Z, Y, X))[1] != "engine":
            continue
        done += 1
        colw += (Z in (256, 512, 1024)) + (Y // 2 in (256, 512, 1024))
        pshape = tuple(int(min(2 * rng.integers(0, 5) + 1, n)) for n in (Z, Y, X))
        psfh = (rng.random(pshape) + 0.05).astype(np.float32)
        if done % 2 == 0:  # point-symmetric: the real-OTF path
            psfh = (0.5 * (psfh + psfh[::-1, ::-1, ::-1])).astype(np.float32)
            psfh = np.maximum(psfh, psfh[::-1, ::-1, ::-1])
            real_otf += int(all(p & 1 for p in pshape))
        volh = (rng.random((Z, Y, X)) * 300).astype(np.float32)
        vol, psf = torch.from_numpy(volh).to(gpu), torch.from_numpy(psfh).to(gpu)
        got = richardson_lucy(vol, psf, 2, 1e-6).cpu().numpy()
        assert rel_err(got, O.richardson_lucy_zyx(volh, psfh, iterations=2, eps=1e-6)) <= 1e-4, ((Z, Y, X), pshape)
        tf = O.compute_transfer_function(psfh, (Z, Y, X))
        got = tikhonov_zyx(vol, torch.from_numpy(tf).to(gpu), 1e-2).cpu().numpy()
        assert rel_err(got, O.tikhonov_zyx(volh, tf, 1e-2)) <= 1e-4, ((Z, Y, X), pshape)
        H = (rng.standard_normal((Z, Y, X)) + 1j * rng.standard_normal((Z, Y, X))).astype(np.complex64) * 0.2
        got = apply_inverse_transfer_function_zyx(vol, H, 0, 1e-2, True).cpu().numpy()
        assert rel_err(got, O.wo_apply_inverse_transfer_function(volh, H, 0, 1e-2, True)) <= 1e-4, (Z, Y, X)
    assert real_otf >= 3 and colw >= 3


def test_fuzz_affine_walks_equal_staged_tiles(gpu, monkeypatch):
    """Random matrices, shapes, crops and dtypes through the z walks of csrc/affine_zwalk.inc / affine_zoblique.inc: every result
    bit-identical to the staged-tile kernel's (BH_AFFINE_NOZWALK=1).  Matrices: random similarity about z (any angle), random
    z scale and shear-free shifts, and small random rotations about oblique axes (weak z coupling)."""
    from biahub_amd import _lib
    from biahub_amd.register import affine_device

    rng = np.random.default_rng(2024)

    def rot(axis, deg):
        ax = np.asarray(axis, dtype=np.float64)
        ax /= np.linalg.norm(ax)
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        th = np.deg2rad(deg)
        return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K

    for case in range(24):
        Zi, Yi = int(rng.integers(5, 80)), int(rng.integers(6, 150))
        Xi = int(rng.integers(2, 40)) * (8 if case % 3 else 1) + (0 if case % 3 else int(rng.integers(0, 7)))
        dt = [np.float32, np.uint16, np.int16, np.uint8][case % 4]
        if dt == np.float32:
            src = (rng.random((Zi, Yi, Xi), dtype=np.float32) * 2000 - 500)
            for _ in range(3):
                src[rng.integers(0, Zi), rng.integers(0, Yi), rng.integers(0, Xi)] = rng.choice([np.nan, np.inf, -np.inf])
        else:
            info = np.iinfo(dt)
            src = rng.integers(info.min, info.max, (Zi, Yi, Xi), endpoint=True).astype(dt)
        M = np.eye(4)
        if case % 2 == 0:   # z-separable
            s = rng.uniform(0.5, 1.6)
            M[:3, :3] = s * rot((1, 0, 0), rng.uniform(-180, 180))
            M[0, 0] = rng.choice([0.0, rng.uniform(0.3, 2.5)])
            M[0, 1] = M[0, 2] = M[1, 0] = M[2, 0] = 0.0
        else:               # weakly oblique
            M[:3, :3] = rng.uniform(0.8, 1.25) * rot((1.0, rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5)), rng.uniform(-4, 4))
        M[:3, 3] = rng.uniform(-12, 12, 3) * [1, 2, 3]
        if case % 5 == 0:
            M[:3, 3] = np.round(M[:3, 3] * 2) / 2   # integer / half-integer shifts: the ties of the inside rule
        Zo, Yo, Xo = (int(rng.integers(1, 90)), int(rng.integers(1, 160)), int(rng.integers(1, 300)))
        lo = tuple(int(rng.integers(0, n // 2 + 1)) for n in (Zo, Yo, Xo))
        cs = tuple(int(rng.integers(1, n - l + 1)) for n, l in zip((Zo, Yo, Xo), lo))
        t = torch.from_numpy(src).to(gpu)
        for interp in ("linear", "nearestneighbor"):
            for boundary in (_lib.BOUNDARY_ITK, _lib.BOUNDARY_SCIPY_CONSTANT):
                monkeypatch.delenv("BH_AFFINE_NOZWALK", raising=False)
                got = affine_device(t, M, (Zo, Yo, Xo), interp, boundary, 7.5, lo, cs)
                monkeypatch.setenv("BH_AFFINE_NOZWALK", "1")
                want = affine_device(t, M, (Zo, Yo, Xo), interp, boundary, 7.5, lo, cs)
                assert torch.equal(got, want), (case, src.shape, str(dt), M.tolist(), (Zo, Yo, Xo), lo, cs, interp, boundary)
    monkeypatch.delenv("BH_AFFINE_NOZWALK", raising=False)


def test_fuzz_radix8_z_pass_equals_radix4(gpu, monkeypatch):
    """512-plane volumes of random y / x extents (2^k and 3 * 2^k, ragged last column tiles) and random PSFs through the radix-8
    Z pass (csrc/fftconv_colz.inc) and the radix-4 one (BH_FC_COLZ=0): R-L with a real and a complex transfer function,
    Tikhonov, phase cross-correlation."""
    from biahub_amd.deconvolve import richardson_lucy, tikhonov_zyx, transfer_function_device
    from biahub_amd.estimate_stabilization import phase_cross_corr_device

    rng = np.random.default_rng(77)
    for case in range(5):
        Y = int(rng.choice([32, 48, 64, 96, 128]))
        X = int(rng.choice([64, 96, 128, 192, 256]))
        shape = (512, Y, X)
        volh = (rng.random(shape, dtype=np.float32) * rng.choice([1.0, 300.0, 6e4])).astype(np.float32)
        pshape = tuple(int(rng.integers(1, 6)) * 2 + 1 for _ in range(3))
        sym = O.gaussian_psf(pshape, tuple(max(q / 4.0, 0.7) for q in pshape))
        asym = (rng.random(tuple(int(rng.integers(1, 9)) for _ in range(3))) + 0.05).astype(np.float32)
        vol = torch.from_numpy(volh).to(gpu)
        mov = torch.roll(vol, (3, -2, 5), (0, 1, 2))
        res = {}
        for tag in ("1", "0"):
            monkeypatch.setenv("BH_FC_COLZ", tag)
            out = [richardson_lucy(vol, torch.from_numpy(p).to(gpu), 2, 1e-6).cpu().numpy() for p in (sym, asym)]
            tf = transfer_function_device(torch.from_numpy(sym).to(gpu), shape, gpu)
            out.append(tikhonov_zyx(vol, tf, 1e-2).cpu().numpy())
            sh, corr = phase_cross_corr_device(vol, mov, "magnitude")
            assert tuple(float(v) for v in sh) == (-3.0, 2.0, -5.0)
            out.append(corr.cpu().numpy())
            res[tag] = out
        monkeypatch.delenv("BH_FC_COLZ", raising=False)
        for a, b in zip(res["1"], res["0"]):
            assert rel_err(a, b) <= 2e-5, (shape, pshape, rel_err(a, b))
        assert rel_err(res["1"][0], O.richardson_lucy_zyx(volh, sym, iterations=2, eps=1e-6)) <= 1e-4, shape


def test_workspace_rebuild_stress(gpu):
    """tools/alloc_stress.py: Richardson-Lucy on alternating shapes with the whole workspace given back to the driver every few
    rounds, so that gigabyte-class blocks (virtual-memory mappings of shuffled 2-MiB chunks) are torn down and rebuilt again and
    again; every result must equal the first of its shape bit for bit.  This is the regression test of the address-range reuse
    hazard (csrc/context.hip dev_free: a released range that was reserved and mapped again delivered stale pages now and then)."""
    import subprocess
    import sys

    from conftest import ROOT

    import os

    # the stress shapes are small: the threshold is lowered so that their buffers take the virtual-memory path
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "alloc_stress.py"), "15"], capture_output=True, text=True,
                       env={**os.environ, "BH_ALLOC_VMM_MIN_MB": "64"})
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "0 mismatches" in r.stdout


def test_fuzz_cubic_warp_vs_oracle(gpu):
    """Cubic (SciPy order 3) warp on random shapes and matrices against the float64 oracle: gentle similarities (LDS tiles,
    interior and face tiles, the plane-combining path when z is left alone), strong rotations and shears (cache launch), output
    grids larger and smaller than the input, non-zero cval, rows that are / are not 16-B aligned."""
    from biahub_amd import _lib
    from biahub_amd.register import affine_device

    rng = np.random.default_rng(314)
    for case in range(24):
        shape = (int(rng.integers(1, 28)), int(rng.integers(4, 60)), int(rng.integers(4, 150)))
        if case % 3 == 0:
            shape = shape[:2] + (4 * (shape[2] // 4 + 1),)
        vol = (rng.random(shape) * 200 - 20).astype(np.float32)
        M = np.eye(4)
        kind = case % 4
        if kind == 0:    # leaves z alone: rotation about z, scale >= 1 along z, shifts
            th = np.deg2rad(rng.uniform(-8, 8))
            M[:3, :3] = [[rng.uniform(1.0, 1.3), 0, 0], [0, np.cos(th), -np.sin(th)], [0, np.sin(th), np.cos(th)]]
        elif kind == 1:  # gentle general affine
            M[:3, :3] = np.eye(3) + rng.uniform(-0.06, 0.06, (3, 3))
        elif kind == 2:  # strong rotation / anisotropic scale
            a, b = np.deg2rad(rng.uniform(20, 80)), np.deg2rad(rng.uniform(-40, 40))
            Rz = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
            Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
            M[:3, :3] = Rz @ Ry @ np.diag(rng.uniform(0.6, 1.6, 3))
        else:            # pure fractional shift (z included)
            pass
        ctr = (np.array(shape) - 1) / 2
        M[:3, 3] = ctr - M[:3, :3] @ ctr + rng.uniform(-3, 3, 3)
        oshape = tuple(int(max(1, n + rng.integers(-3, 6))) for n in shape)
        cval = float(rng.uniform(-5, 5))
        got = affine_device(vol, M, oshape, "cubic", _lib.BOUNDARY_SCIPY_CONSTANT, cval).cpu().numpy()
        want = O.spline_affine_pull(vol, M, oshape, cval)
        assert rel_err(got, want) <= 1e-5, (case, shape, oshape)


def test_fuzz_lz4_device_codec(gpu):
    """Random byte streams through the device codec — runs, periodic data with periods up to beyond the decoder's LDS ring and up
    to the format's 65535-byte reach, noise, ramps, mixed — at typesizes 1 / 2 / 4, with and without shuffles, chunks shorter
    than a block and with a short last block: GPU frames -> host decoder (pyarrow's lz4) and -> GPU decoder, bytes exact."""
    from biahub_amd import codecs

    rng = np.random.default_rng(2718)

    def stream(n):
        parts, left = [], n
        while left > 0:
            m = int(min(left, rng.integers(1, 200_000)))
            k = rng.integers(0, 5)
            if k == 0:
                seg = np.full(m, rng.integers(0, 256), np.uint8)
            elif k == 1:
                period = int(rng.choice([1, 2, 3, 7, 64, 255, 4096, 12000, 12290, 20000, 65535, 70000]))
                seg = np.resize(rng.integers(0, 256, period, dtype=np.uint8), m)
            elif k == 2:
                seg = rng.integers(0, 256, m, dtype=np.uint8)
            elif k == 3:
                seg = (np.arange(m) // int(rng.integers(1, 40))).astype(np.uint8)
            else:
                seg = np.where(rng.random(m) < 0.9, 17, rng.integers(0, 256, m)).astype(np.uint8)
            parts.append(seg)
            left -= m
        return np.concatenate(parts)

    for case in range(10):
        ts = int(rng.choice([1, 2, 4]))
        mode = int(rng.choice([0, 1, 2])) if ts > 1 else int(rng.choice([0, 2]))
        nch = int(rng.integers(1, 4))
        cbytes = ts * int(rng.integers(200, 400_000))
        raw = stream(nch * cbytes)
        bsz = codecs.default_blocksize(ts) if case % 2 else int(rng.choice([16384, 65536, 262144]))
        src = torch.from_numpy(raw).to(gpu)
        filt = torch.empty_like(src)
        for i in range(nch):
            codecs.filter_device(src[i * cbytes:(i + 1) * cbytes], filt[i * cbytes:(i + 1) * cbytes], bsz, ts, mode)
        packed, offs = codecs.blosc_lz4_compress_device(filt, nch, cbytes, bsz, ts, mode)
        host = packed[: offs[-1]].cpu().numpy()
        frames = [host[offs[i]: offs[i + 1]].tobytes() for i in range(nch)]
        back = torch.empty_like(filt)
        heads = codecs.blosc_lz4_decode_frames_device(frames, back, [i * cbytes for i in range(nch)])
        assert torch.equal(back, filt), (case, ts, mode, cbytes, bsz)
        for i in range(nch):
            assert np.array_equal(codecs.blosc_decompress(frames[i][: heads[i].cbytes]), raw[i * cbytes:(i + 1) * cbytes]), (case, i)
