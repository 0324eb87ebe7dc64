#!/usr/bin/env python3
"""Headline benchmark: raw-input voxels/s for Richardson-Lucy (10 iterations) + deskew of a
(512, 2048, 2048) float32 volume per GPU (BASELINE.json configs[1]), positions sharded over ranks.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one synthetic position resident in HBM: the reference's
pipeline order (README.md:140-148) deconvolve (raw coordinates) -> deskew, with the deskew settings
of settings/example_deskew_settings.yml (36.17 deg, 0.371, keep_overhang, N=3, overhang_fill mean).
Rank 0 prints ONE JSON line.  Every rank processes its own position (weak scaling, no data-path
collective); only the timing barrier and the max-over-ranks reduction use RCCL.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from biahub_amd import _lib, parallel  # noqa: E402
from biahub_amd.deconvolve import PreparedRichardsonLucy, richardson_lucy  # noqa: E402
from biahub_amd.deskew import fast_deskew_zyx, get_deskewed_data_shape  # noqa: E402
from biahub_amd.device import get_context  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
DESKEW = dict(ls_angle_deg=36.17, px_to_scan_ratio=0.371, keep_overhang=True, average_n_slices=3,
              overhang_fill="mean")
PSF_SHAPE, PSF_SIGMA = (33, 17, 17), (3.0, 1.5, 1.5)


def gaussian_psf(shape, sigma, device):
    ax = [torch.arange(n, dtype=torch.float64, device=device) - (n - 1) / 2 for n in shape]
    g = [torch.exp(-0.5 * (a / s) ** 2) for a, s in zip(ax, sigma)]
    psf = g[0][:, None, None] * g[1][None, :, None] * g[2][None, None, :]
    return (psf / psf.sum()).to(torch.float32)


def _pooled(fn):
    """Run an allocating function inside biahub_amd.device.volume_pool (the library's page layout for volumes in HBM)."""
    import functools

    @functools.wraps(fn)
    def wrapper(*a, **k):
        from biahub_amd.device import volume_pool
        with volume_pool():
            return fn(*a, **k)
    return wrapper


@_pooled
def synthetic_position(shape, seed, device):
    """Camera-count-like volume generated on device: offset 110 + Gaussian noise + sparse bright blobs."""
    g = torch.Generator(device=device).manual_seed(seed)
    Z, Y, X = shape
    vol = torch.empty(shape, dtype=torch.float32, device=device)
    vol.normal_(110.0, 4.0, generator=g)
    n_blobs = max(16, (Z * Y * X) // 2**19)
    zz = torch.randint(2, Z - 2, (n_blobs,), generator=g, device=device)
    yy = torch.randint(2, Y - 2, (n_blobs,), generator=g, device=device)
    xx = torch.randint(2, X - 2, (n_blobs,), generator=g, device=device)
    amp = torch.rand((n_blobs,), generator=g, device=device) * 3800 + 200
    for dz, dy, dx, w in [(0, 0, 0, 1.0), (1, 0, 0, 0.6), (-1, 0, 0, 0.6), (0, 1, 0, 0.6), (0, -1, 0, 0.6),
                          (0, 0, 1, 0.6), (0, 0, -1, 0.6)]:
        vol.index_put_((zz + dz, yy + dy, xx + dx), amp * w, accumulate=True)
    return vol.round_().clamp_(0, 65535)


def pmc_traffic(key, shape):
    """HBM bytes per launch from the newest committed PMC passes (profiles/rNN_pmc_hbm_traffic.json: separate
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this command, fetch doubled for gfx950 as
    MI355X_MICROARCH.md prescribes).  STATIC: collected once per round on the kernels of that round, not during this run
    (counters cannot be read from inside the process); only valid for the shape it was collected on.  -> (bytes, source)"""
    def order(f):  # rNN, rNNa .. rNNz, rNNaa ..: round number, then the run's tag in the order the tags were handed out
        import re

        m = re.match(r"r(\d+)([a-z]*)_", f.name)
        return (int(m.group(1)), len(m.group(2)), m.group(2)) if m else (-1, 0, "")

    files = sorted((ROOT / "profiles").glob("r*_pmc_hbm_traffic.json"), key=order)
    for f in reversed(files):
        rec = json.load(open(f))
        if list(shape) == rec.get("shape") and rec.get(key):
            return rec.get(key), f"static from profiles/{f.name}"
    return None, None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(iterations):
    """The oracle (CPU restatement of the path) timed on bounded samples of the same workload, rank 0 only: the two sizes
    SURVEY.md 8(d) names, (128, 512, 512) and (256, 1024, 1024) — about half a minute of host work together."""
    import scipy.fft

    from oracle import oracle_np as O  # checker / baseline only — never the thing measured as `value`

    samples = []
    for shape in ((128, 512, 512), (256, 1024, 1024)):
        vol = O.synthetic_volume(shape, seed=7, n_blobs=32)
        psf = O.gaussian_psf(PSF_SHAPE, PSF_SIGMA)
        t0 = time.perf_counter()
        rl = O.richardson_lucy_zyx(vol, psf, iterations, 1e-6)
        t1 = time.perf_counter()
        O.fast_deskew_zyx(rl, DESKEW["ls_angle_deg"], DESKEW["px_to_scan_ratio"], True, DESKEW["average_n_slices"],
                          DESKEW["overhang_fill"])
        t2 = time.perf_counter()
        samples.append({"shape": list(shape), "seconds": t2 - t0, "rl_seconds": t1 - t0, "deskew_seconds": t2 - t1,
                        "voxels_per_s": float(np.prod(shape) / (t2 - t0))})
        del vol, rl
    big = samples[-1]
    return {
        "value": big["voxels_per_s"],
        "unit": "voxels/s",
        "cores": os.cpu_count(),
        "threads": {"scipy_fft_workers": os.cpu_count(), "torch": torch.get_num_threads()},
        "cpu_model": cpu_model(),
        "kind": "port",
        "sample": f"oracle RL({iterations} it, scipy.fft workers=all cores) + deskew(fill mean) on one float32 volume of "
                  f"{tuple(samples[0]['shape'])} ({samples[0]['seconds']:.1f} s) and of {tuple(big['shape'])} "
                  f"({big['seconds']:.1f} s); value = the larger sample",
        "samples": samples,
    }


def ops_suite(vol, psf, dev, ctx):
    """The other operators of the path at the bench shape, device-resident, each timed by the library's own HIP events on its
    stream (third run of three): ms, algorithmic bytes and that rate as a fraction of 8 TB/s.  Reported
    beside the headline, never part of `value`."""
    from biahub_amd.apply_inverse_transfer_function import PreparedInverseFilter, apply_inverse_transfer_function_zyx
    from biahub_amd.deconvolve import tikhonov_zyx, transfer_function_device
    from biahub_amd.estimate_stabilization import phase_cross_corr_device
    from biahub_amd.flat_field import flat_field_device
    from biahub_amd.register import affine_device

    shape = tuple(vol.shape)
    V = float(np.prod(shape))
    out = {}

    def rec(name, fn, slot, nbytes, note=None):
        ms = 0.0
        for _ in range(3):
            r = fn()
            ms = ctx.elapsed_ms(slot)
            del r
        out[name] = {"ms": ms, "algorithmic_bytes": nbytes, "GBps": nbytes / ms / 1e6, "frac": nbytes / ms / 1e6 / HBM_PEAK_GBS}
        if note:
            out[name]["note"] = note

    th = np.deg2rad(2.0)
    M = np.array([[1.02, 0, 0, 3.5], [0, 1.02 * np.cos(th), -1.02 * np.sin(th), -12.25],
                  [0, 1.02 * np.sin(th), 1.02 * np.cos(th), 20.75], [0, 0, 0, 1.0]])  # 2 deg, 1.02x, fractional shift
    rec("affine_linear_f32", lambda: affine_device(vol, M, shape, "linear"), _lib.T_AFFINE, 8 * V,
        "4 (V_in + V_out); rotation about z: the wave-private z walk (csrc/affine_zwalk.inc)")
    ax = np.array([1.0, 0.4, 0.3]) / np.linalg.norm([1.0, 0.4, 0.3])
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    Mo = np.eye(4)
    Mo[:3, :3] = 1.02 * (np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K)
    Mo[:3, 3] = (3.5, -12.25, 20.75)
    rec("affine_linear_f32_oblique", lambda: affine_device(vol, Mo, shape, "linear"), _lib.T_AFFINE, 8 * V,
        "the same 2 deg / 1.02x similarity about an oblique axis: z couples weakly with y and x — the z walk with per-lane source "
        "planes (csrc/affine_zoblique.inc); stronger couplings run the staged-tile kernel")
    c0 = np.array([(n - 1) / 2 for n in shape])
    th = np.deg2rad(45.0)
    R45 = np.array([[np.cos(th), 0.0, np.sin(th)], [0.0, 1.0, 0.0], [-np.sin(th), 0.0, np.cos(th)]])
    M45 = np.eye(4)
    M45[:3, :3] = R45
    M45[:3, 3] = c0 - R45 @ c0
    rec("affine_linear_f32_rot45_about_y", lambda: affine_device(vol, M45, shape, "linear"), _lib.T_AFFINE, 8 * V,
        "45 deg about the y axis through the centre: z couples strongly with x — compact 8 x 8 x 16 output blocks, their source "
        "box staged in LDS (affine_gather_kernel; the tile kernel's fallback took 59 ms)")
    x = torch.empty_like(vol)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for _ in range(3):
        ev[0].record()
        x.copy_(vol)
        ev[1].record()
        torch.cuda.synchronize(dev)
    ms = ev[0].elapsed_time(ev[1])
    out["copy_f32_reference"] = {"ms": ms, "algorithmic_bytes": 8 * V, "GBps": 8 * V / ms / 1e6, "frac": 8 * V / ms / 1e6 / HBM_PEAK_GBS,
                                 "note": "torch copy_ of the same volume into a second buffer: the out-of-place streaming rate on this part"}
    del x
    rec("affine_nearest_f32", lambda: affine_device(vol, M, shape, "nearestneighbor"), _lib.T_AFFINE, 8 * V)
    rec("affine_cubic_f32", lambda: affine_device(vol, M, shape, "cubic", _lib.BOUNDARY_SCIPY_CONSTANT), _lib.T_AFFINE, 8 * V,
        "SciPy order 3 (Transform.apply(order=3), apply_affine_transform(method='scipy')): three prefilter passes + 64 taps per "
        "voxel from LDS-staged source boxes (csrc/spline.hip; bound by vector instructions, not HBM); bytes by 4 (V_in + V_out) "
        "like the other warps, the prefilter's passes (24 V more) are not in the model")
    v16 = vol.to(torch.uint16)
    rec("affine_linear_u16_in", lambda: affine_device(v16, M, shape, "linear"), _lib.T_AFFINE, 6 * V, "2 V_in + 4 V_out")
    rec("flat_field_u16", lambda: flat_field_device(v16), _lib.T_FLATFIELD, 8 * V, "two 2-byte reads (median, apply) + one f32 write")
    # deskew + mean fill of a resident float32 volume on its own (no operator in front to hand over row sums): row-sum read,
    # resampling with whole-row stores, and the conditional mask pipeline that returns at once
    out_shape, _ = get_deskewed_data_shape(shape, DESKEW["ls_angle_deg"], DESKEW["px_to_scan_ratio"], True, DESKEW["average_n_slices"])
    dk_bytes = 4.0 * (V + float(np.prod(out_shape)))
    ms_pair = 0.0
    for _ in range(3):
        r = fast_deskew_zyx(vol, **DESKEW)
        ms_pair = ctx.elapsed_ms(_lib.T_DESKEW) + ctx.elapsed_ms(_lib.T_FILL)
        del r
    out["deskew_fill_pair"] = {"ms": ms_pair, "algorithmic_bytes": dk_bytes, "GBps": dk_bytes / ms_pair / 1e6,
                               "frac": dk_bytes / ms_pair / 1e6 / HBM_PEAK_GBS,
                               "note": "deskew with overhang_fill='mean' standalone: geometry bits + row sums of the input (one read) + "
                                       "one-pass resampling kernel + the conditional mask pipeline; 4 (V + V_out)"}
    del v16
    tf = transfer_function_device(psf, shape, dev)
    rec("tikhonov", lambda: tikhonov_zyx(vol, tf, 1e-3), _lib.T_TIKHONOV, 50 * V, "50 V: 2 FFTs (3-pass model) + real filter")
    prep = PreparedInverseFilter(tf, shape, 0, 1e-3, "f32", dev)
    rec("apply_inv_tf_f32", lambda: prep(vol, True), _lib.T_TIKHONOV, 56 * V,
        "config 5 step per volume with the inverse filter staged once per position: 2 FFTs (48 V) + complex filter (8 V), "
        "mean normalisation fused into the forward X pass")
    prep.close()
    prep = PreparedInverseFilter(tf, shape, 0, 1e-3, "bf16", dev)
    rec("apply_inv_tf_bf16", lambda: prep(vol, True), _lib.T_TIKHONOV, 52 * V, "the staged filter kept as bfloat16 pairs")
    prep.close()
    rec("apply_inv_tf_f32_one_shot", lambda: apply_inverse_transfer_function_zyx(vol, tf, 0, 1e-3, True, "f32"), _lib.T_TIKHONOV,
        56 * V, "stage the filter (reads the transfer function twice) + apply + release, per call")
    del tf
    mov = torch.roll(vol, (1, -3, 17), (0, 1, 2))
    phase_cross_corr_device(vol, mov, "magnitude", want_corr=False)  # untimed: allocates the second spectrum and the correlation volume
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(3):
        sh, _ = phase_cross_corr_device(vol, mov, "magnitude", want_corr=False)
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) / 3 * 1e3
    assert tuple(float(v) for v in sh) == (-1.0, 3.0, -17.0), sh
    out["phase_cross_corr"] = {"ms": ms, "algorithmic_bytes": 80 * V, "GBps": 80 * V / ms / 1e6, "frac": 80 * V / ms / 1e6 / HBM_PEAK_GBS,
                               "note": "wall clock incl. the argmax read-back; 3 FFTs (72 V) + product (8 V) by the model; the product runs inside the Z pass of the second transform, the peak search inside the last inverse pass (no correlation volume)"}
    # the stabilisation estimate's loop: every timepoint against ONE stored image (prepared handle: its spectrum is kept)
    from biahub_amd.estimate_stabilization import PreparedPhaseCrossCorr
    with PreparedPhaseCrossCorr(mov, fixed_is_second=True, device=dev) as hp:
        hp(vol, "magnitude")
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(3):
            sh, _ = hp(vol, "magnitude")
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) / 3 * 1e3
    assert tuple(float(v) for v in sh) == (-1.0, 3.0, -17.0), sh
    out["phase_cross_corr_prepared"] = {"ms": ms, "algorithmic_bytes": 56 * V, "GBps": 56 * V / ms / 1e6, "frac": 56 * V / ms / 1e6 / HBM_PEAK_GBS,
                                        "note": "per timepoint with the reference timepoint's spectrum kept (t_reference first): 2 FFTs (48 V) + product (8 V) by the model"}
    del mov
    ctx.release_workspace()
    # Richardson-Lucy on DESKEWED volumes — BASELINE config 4 in its literal order (deskew -> deconvolve), and what the
    # reference's own pipeline does with its FFT reconstruction (nextflow/mantis-v2.nf:116-125): awkward row lengths, the
    # engine at a wrap-padded box.  Prepared handle, per-volume wall time of the third call.
    # a measured PSF (estimate-psf) is never point-symmetric: the complex transfer function, two complex Z-pass products
    try:
        g = torch.Generator(device=dev).manual_seed(9)
        apsf = psf * (1.0 + 0.2 * torch.rand(psf.shape, generator=g, device=dev))
        apsf = apsf / apsf.sum()
        with PreparedRichardsonLucy(apsf, shape, dev) as h:
            res = torch.empty_like(vol)
            for _ in range(3):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                h(vol, 10, 1e-6, out=res)
                torch.cuda.synchronize(dev)
                ms = (time.perf_counter() - t0) * 1e3
            assert not h.otf_is_real
            out["rl10_asymmetric_psf"] = {"ms": ms, "otf_is_real": h.otf_is_real, "voxels_per_s": V / ms * 1e3, "algorithmic_bytes": 1120 * V,
                                          "GBps": 1120 * V / ms / 1e6, "frac": 1120 * V / ms / 1e6 / HBM_PEAK_GBS,
                                          "note": "10 iterations with the COMPLEX transfer function of a PSF that is not point-symmetric "
                                                  "(the headline's Gaussian is: one float per bin in the Z passes)"}
        del res, apsf
    except RuntimeError as e:
        out["rl10_asymmetric_psf"] = {"skipped": str(e)[:200]}
    ctx.release_workspace()
    for name, dshape in (("rl10_deskewed_config4_volume", (342, 1024, 1517)), ("rl10_deskewed_config2_volume", (683, 2048, 3034))):
        try:
            g = torch.Generator(device=dev).manual_seed(5)
            dvol = torch.empty(dshape, dtype=torch.float32, device=dev).uniform_(90.0, 400.0, generator=g)
            with PreparedRichardsonLucy(psf, dshape, dev) as h:
                Vd = float(np.prod(dshape))
                Vb = float(np.prod(h.box))
                res = torch.empty_like(dvol)
                for _ in range(3):
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    h(dvol, 10, 1e-6, out=res)
                    torch.cuda.synchronize(dev)
                    ms = (time.perf_counter() - t0) * 1e3
                out[name] = {"ms": ms, "shape": list(dshape), "box": list(h.box), "backend": h.backend, "voxels_per_s": Vd / ms * 1e3,
                             "algorithmic_bytes": 1120 * Vd, "GBps": 1120 * Vd / ms / 1e6, "frac": 1120 * Vd / ms / 1e6 / HBM_PEAK_GBS,
                             "frac_of_box": 1120 * Vb / ms / 1e6 / HBM_PEAK_GBS,
                             "note": "10 iterations incl. wrap-padding in and crop out; bytes by the 112 V model on the volume's own "
                                     "voxels (frac) and on the padded box the transforms run at (frac_of_box)"}
            del dvol, res
        except RuntimeError as e:  # not enough free HBM beside the resident position
            out[name] = {"skipped": str(e)[:200]}
        ctx.release_workspace()
    return out


def overlapped_units(vol_host_u16, rl, iterations, dev, ctx, n_units=8, n_landing=2):
    """`n_units` positions through biahub_amd.pipeline.run_overlapped: upload of unit i + 1 (pinned uint16), compute of unit i
    (prepared R-L handle: nothing in the compute leg waits on the host) and download of unit i - 1 (float32 deskewed volume
    into one of two pinned landing blocks) on three streams.  Returns ms per unit and the per-leg timeline of the measured run."""
    from biahub_amd.pipeline import run_overlapped, timeline_ms

    from biahub_amd.device import volume_pool
    with volume_pool(dev):
        probe = fast_deskew_zyx(rl(vol_host_u16.to(dev), 0, 1e-6), **DESKEW)
    # n_landing = 1 (the N > 1 line): downloads are serial on their stream and nobody consumes the blocks here, so one pinned
    # landing block bounds what eight ranks pin on one host (8 x (4.3 + 17) GB instead of 8 x 38)
    landing = [torch.empty(probe.shape, dtype=torch.float32, pin_memory=True) for _ in range(n_landing)]
    del probe
    count = [0]

    def download(t):
        buf = landing[count[0] % n_landing]
        count[0] += 1
        buf.copy_(t, non_blocking=True)
        return buf

    def run(rows):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in run_overlapped(range(n_units), lambda i: vol_host_u16.to(dev, non_blocking=True),
                                lambda d: fast_deskew_zyx(rl(d, iterations, 1e-6), **DESKEW), download, dev, timeline=rows):
            pass
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / n_units

    was_timing = ctx.timing
    ctx.set_timing(False)  # the library's event timing reads its events back on the host: a stall inside the compute leg
    try:
        run(None)
        rows = []
        per_unit = run(rows)
        tl = timeline_ms(rows)
    finally:
        ctx.set_timing(was_timing)
    del landing
    legs = {"h2d_ms": float(np.mean([r[1] - r[0] for r in tl])), "compute_ms": float(np.mean([r[3] - r[2] for r in tl])),
            "d2h_ms": float(np.mean([r[5] - r[4] for r in tl]))}
    return per_unit, tl, legs


def end_to_end(vol_host_u16, psf, rl, iterations, dev, ctx):
    """What the reference's operator boundary costs beside the resident figure (biahub/deskew.py:578-579: the worker uploads
    the volume and takes the result back): uint16 camera stack in pinned host memory -> H2D -> R-L + deskew -> float32
    result D2H into a pinned block, through the adapters' own transfer helpers.  PCIe-bound, never `value`."""
    from biahub_amd.device import to_host, volume_pool

    def once():
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        with volume_pool(dev):
            d = vol_host_u16.to(dev, non_blocking=False)       # 2 B/voxel across PCIe
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        out = fast_deskew_zyx(rl(d, iterations, 1e-6), **DESKEW)
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        host = to_host(out)                                    # 4 B/voxel of the deskewed volume, pinned destination
        t3 = time.perf_counter()
        nbytes_out = host.nbytes
        del host
        comp_leg = compressed_unit(out) if measure_compressed else None
        del out, d
        return t1 - t0, t2 - t1, t3 - t2, nbytes_out, comp_leg

    def compressed_unit(out):
        """The same result leaving as the Blosc-lz4 frames a compressed store takes (csrc/lz4.hip; what
        ZarrArray.encode_volume_device does for a device-resident result): bit-shuffle + LZ4 + frame assembly on the GPU, then only
        the frames cross PCIe."""
        from biahub_amd import codecs

        try:
            Zo, Yo, Xo = out.shape
            zc = 16
            nch = -(-Zo // zc)
            cbytes = zc * Yo * Xo * 4
            bsz = codecs.default_blocksize(4)
            v8 = out.view(torch.uint8).reshape(-1)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            stage = torch.empty(nch * cbytes, dtype=torch.uint8, device=dev)
            for i in range(nch):
                src = v8[i * cbytes:(i + 1) * cbytes]
                if src.numel() != cbytes:  # the overhanging chunk: zero planes behind the volume
                    full = torch.zeros(cbytes, dtype=torch.uint8, device=dev)
                    full[: src.numel()] = src
                    src = full
                codecs.filter_device(src, stage[i * cbytes:(i + 1) * cbytes], bsz, 4, codecs.BLOSC_BITSHUFFLE)
            packed, offs = codecs.blosc_lz4_compress_device(stage, nch, cbytes, bsz, 4, codecs.BLOSC_BITSHUFFLE)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            host = to_host(packed[: offs[-1]])
            t2 = time.perf_counter()
            nb = int(host.nbytes)
            del host, packed, stage
            return {"encode_ms": (t1 - t0) * 1e3, "d2h_ms": (t2 - t1) * 1e3, "frames_bytes": nb, "ratio": nb / (Zo * Yo * Xo * 4.0),
                    "chunks": nch, "note": "float32 result as Blosc-lz4 frames (bit shuffle, 256-KiB blocks, chunks of 16 planes) made "
                                           "on the GPU; d2h_ms carries the frames only.  A deconvolved float32 volume has noisy low "
                                           "mantissa planes: the ratio is what LZ4 finds in the upper ones"}
        except (RuntimeError, ValueError) as e:
            return {"skipped": str(e)[:200]}

    measure_compressed = True
    once()  # pinned blocks, allocator and the codec's scratch warm
    h2d, comp, d2h, nb_out, comp_leg = once()
    V = vol_host_u16.numel()
    res = {"voxels_per_s": V / (h2d + comp + d2h), "ms": (h2d + comp + d2h) * 1e3, "h2d_ms": h2d * 1e3, "compute_ms": comp * 1e3,
           "d2h_ms": d2h * 1e3, "h2d_GBps": V * 2 / h2d / 1e9, "d2h_GBps": nb_out / d2h / 1e9,
           "note": "uint16 in (pinned) -> float32 deskewed out (pinned); serial, no overlap between positions"}
    if comp_leg is not None:
        res["compressed_unit"] = comp_leg
    # the same units through biahub_amd.pipeline.run_overlapped: upload of unit i + 1, compute of unit i and download of unit
    # i - 1 on three streams (what a plate job on one GPU can do; the reference's worker is the serial form above)
    try:
        per_unit, tl, legs = overlapped_units(vol_host_u16, rl, iterations, dev, ctx)
        steady = (tl[-1][5] - tl[0][5]) / (len(tl) - 1) if len(tl) > 1 else per_unit * 1e3  # cadence of finished units
        res["overlapped"] = {"ms_per_unit": per_unit * 1e3, "voxels_per_s": V / per_unit, "units": len(tl),
                             "steady_state_ms_per_unit": steady, "steady_state_voxels_per_s": V / steady * 1e3,
                             "legs_while_overlapped": legs, "ratio_to_compute_leg": per_unit * 1e3 / (comp * 1e3),
                             "steady_state_ratio_to_compute_leg": steady / (comp * 1e3),
                             "timeline_ms": tl,
                             "note": "biahub_amd.pipeline.run_overlapped: H2D / compute / D2H of consecutive units on three streams, "
                                     "two pinned landing blocks, prepared R-L handle (no host synchronisation in the compute leg); "
                                     "ms_per_unit = wall time of the batch / units (includes filling and draining the pipeline: one "
                                     "upload before the first and one download after the last compute leg); steady_state = the cadence "
                                     "at which finished units arrive; timeline rows = [h2d0, h2d1, compute0, compute1, d2h0, d2h1] "
                                     "per unit, ms since the first upload started"}
    except RuntimeError as e:  # pinned memory for two 17-GB landing blocks not available on this host
        res["overlapped"] = {"skipped": str(e)[:200]}
    return res


def rccl_evidence(dev, world):
    """Every rank's GPU as the collective library sees the job: device name and PCI bus id of each rank, all_gathered over the
    process group bench.py runs its barrier on (RCCL at N > 1).  N ranks on fewer than N distinct GPUs is an error unless the
    run is a declared rehearsal (BH_DIST_BACKEND=gloo wraps ranks onto the cards there are)."""
    import torch.distributed as dist

    props = torch.cuda.get_device_properties(dev)
    total = props.total_memory  # the bus id distinguishes the cards of one node; the name alone is the same for all eight
    bus = f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
    mine = {"rank": parallel.world_info()[0], "device_index": int(dev.index), "name": props.name, "pci_bus_id": bus,
            "uuid": str(getattr(props, "uuid", "")), "hbm_gb": round(total / 1e9, 1)}
    rows = parallel.gather_objects(mine)
    backend = dist.get_backend() if dist.is_initialized() else "none (single process)"
    distinct = len({r["pci_bus_id"] for r in rows})
    return {"world": world, "backend": "rccl (torch 'nccl')" if backend == "nccl" else backend, "devices": rows,
            "distinct_gpus": distinct}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--shape", type=int, nargs=3, default=[512, 2048, 2048], metavar=("Z", "Y", "X"))
    ap.add_argument("--iterations", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-ops", action="store_true")
    ap.add_argument("--host-fed", choices=["auto", "on", "off"], default="auto",
                    help="after the resident measurement every rank also runs the overlapped host-fed pipeline (pinned uint16 in, "
                         "pinned float32 out) so that host-memory / PCIe contention between ranks shows in the N>1 line; auto = on for N>1")
    ap.add_argument("--host-fed-units", type=int, default=2)
    args = ap.parse_args()

    rank, local_rank, world = parallel.world_info()
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    local_rank %= max(1, torch.cuda.device_count())  # one rank per GPU; wraps only in a BH_DIST_BACKEND=gloo rehearsal
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    parallel.init("nccl", dev)  # RCCL; used only for the timing barrier / max-over-ranks, never on the data path
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    rccl = rccl_evidence(dev, world)  # all_gather over the process group: N ranks must sit on N distinct GPUs
    rehearsal = os.environ.get("BH_DIST_BACKEND") == "gloo"
    if rccl["distinct_gpus"] != world and not rehearsal:
        if rank == 0:
            print(json.dumps({"error": "ranks share GPUs", "rccl": rccl}), flush=True)
        sys.exit(3)
    # one synthetic position per rank and step: position index = rank (round-robin shard of a `world`-position plate)
    my_positions = parallel.shard_positions(range(world), rank, world)
    assert my_positions == [rank]

    shape = tuple(args.shape)
    V = int(np.prod(shape))
    out_shape, _ = get_deskewed_data_shape(shape, DESKEW["ls_angle_deg"], DESKEW["px_to_scan_ratio"], True,
                                           DESKEW["average_n_slices"])
    V_out = int(np.prod(out_shape))
    psf = gaussian_psf(PSF_SHAPE, PSF_SIGMA, dev)
    # two distinct synthetic positions per rank, alternated over the steps (position index = rank: weak scaling)
    vols = [synthetic_position(shape, 0xB1A0 + rank + 1000 * k, dev) for k in range(2)]
    ctx = get_context(dev)
    ctx.set_timing(True)
    nstep = [0]
    # the transfer function of the plate's PSF is built once (the reference: biahub/deconvolve.py:140-149), outside the timed
    # region; every step applies it (bh_richardson_lucy_apply: no host synchronisation of its own)
    rl_prepared = PreparedRichardsonLucy(psf, shape, dev)

    # the last R-L update pass leaves the row sums of its result behind (one float64 per row): the deskew's mean fill is derived
    # from them, so the fill costs no pass of its own over the volume (csrc/deskew_rows.inc)
    row_sums = torch.empty(shape[:2], dtype=torch.float64, device=dev)

    def step():
        vol = vols[nstep[0] % len(vols)]
        nstep[0] += 1
        rl, rs = rl_prepared(vol, args.iterations, 1e-6, row_sums=row_sums)
        return fast_deskew_zyx(rl, row_sums=rs, **DESKEW)

    def fence():
        torch.cuda.synchronize(dev)
        parallel.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        out = step()
        del out
    rl_ms, dk_ms, fill_ms = [], [], []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
        # per-kernel HIP-event timings recorded by the library on ITS stream (reading them syncs that stream,
        # which the step's result needs anyway before the next position overwrites the workspace)
        rl_ms.append(ctx.elapsed_ms(_lib.T_RL_ITER))
        dk_ms.append(ctx.elapsed_ms(_lib.T_DESKEW))
        fill_ms.append(ctx.elapsed_ms(_lib.T_FILL))
        del out
    fence()
    dt = time.perf_counter() - t0
    dt = parallel.max_over_ranks(dt, dev)

    # host-fed leg on EVERY rank (N > 1 by default): each rank pipelines units from its own pinned uint16 block into its own
    # pinned float32 blocks — the one thing that can break 8-GPU scaling (8 x 57 GB/s of D2H into one host) is in the line
    host_fed = None
    if args.host_fed == "on" or (args.host_fed == "auto" and world > 1):
        # every rank runs the SAME sequence of collectives: a rank that fails locally (pinned memory, HBM) does not leave the
        # others waiting in an all_reduce — the failure is agreed on first, then every rank skips the leg's reductions together
        err, per_unit, steady, tl, legs = None, 0.0, 0.0, None, None
        try:
            host = torch.empty(shape, dtype=torch.uint16, pin_memory=True)
            host.copy_(vols[0].to(torch.uint16))
        except RuntimeError as e:
            err, host = str(e)[:200], None
        any_failed = parallel.max_over_ranks(1.0 if err else 0.0, dev) > 0.0
        if not any_failed:
            fence()
            try:
                per_unit, tl, legs = overlapped_units(host, rl_prepared, args.iterations, dev, ctx, n_units=args.host_fed_units,
                                                      n_landing=1 if world > 1 else 2)
                steady = (tl[-1][5] - tl[0][5]) / (len(tl) - 1) / 1e3 if len(tl) > 1 else per_unit
            except RuntimeError as e:
                err = str(e)[:200]
            any_failed = parallel.max_over_ranks(1.0 if err else 0.0, dev) > 0.0
        del host
        if any_failed:
            host_fed = {"skipped": err or "another rank failed"}
        else:
            slowest = parallel.max_over_ranks(per_unit, dev)
            steady = parallel.max_over_ranks(steady, dev)
            host_fed = {"ms_per_unit_slowest_rank": slowest * 1e3, "voxels_per_s": world * V / slowest,
                        "steady_state_ms_per_unit_slowest_rank": steady * 1e3, "steady_state_voxels_per_s": world * V / steady,
                        "units_per_rank": args.host_fed_units, "legs_while_overlapped_rank0": legs,
                        "note": "every rank: pinned uint16 H2D -> R-L + deskew -> float32 D2H into pinned blocks, three streams "
                                "(biahub_amd.pipeline.run_overlapped); all ranks at once, max over ranks; never `value`"}

    if rank == 0:
        rl_iter_s = float(np.mean(rl_ms)) / 1e3
        deskew_s = float(np.mean(dk_ms)) / 1e3
        fill_s = float(np.mean(fill_ms)) / 1e3
        rl_bytes = 112.0 * V            # 4 real 3-D FFTs (3-pass model) + fused pointwise
        rl_bytes_1pass = 48.0 * V       # SURVEY.md §8d lower bound: every FFT one read + one write
        dk_bytes = 4.0 * (V + V_out)    # read every input voxel once, write every output voxel once
        rl_moved, rl_src = pmc_traffic("rl_iteration", shape)
        dk_moved, dk_src = pmc_traffic("deskew_pair", shape)
        pair_s = deskew_s + fill_s
        result = {
            "metric": "voxels/s for deskew+10-iter R-L deconv, 2048^2x512 f32",
            "value": world * args.steps * V / dt,
            "unit": "voxels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"1 position/GPU: R-L {args.iterations} it (PSF {PSF_SHAPE}) then deskew "
                            f"{shape}->{tuple(out_shape)} (36.17 deg, 0.371, N=3, fill mean; the fill value derived from row sums the last R-L pass reduces), input resident in HBM; two "
                            "distinct positions alternate over the steps; the transfer function of the plate's PSF is prepared "
                            "once before the timed region (bh_richardson_lucy_create, as the reference computes it once per plate)",
                "raw_shape_zyx": list(shape),
                "deskewed_shape_zyx": list(out_shape),
                "positions_per_step": world,
            },
            "roofline": {
                "kernel": "one Richardson-Lucy iteration = 8 in-place passes of csrc/fftconv.hip: 2 x (colw_kernel Y fwd "
                          "[register stages, csrc/fftconv_colw.inc], colz_kernel Z fwd*OTF*inv [radix-8 register stages, "
                          "csrc/fftconv_colz.inc; real OTF: the Gaussian PSF is point-symmetric], colw_kernel Y inv, xw_kernel<FUSED_*>: inverse X + RL epilogue + next forward X in "
                          "registers, csrc/fftconv_xw.inc)",
                "bound": "hbm",
                "achieved": rl_bytes / rl_iter_s / 1e9,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": rl_bytes / rl_iter_s / 1e9 / HBM_PEAK_GBS,
                "traffic": rl_moved,
                "traffic_source": rl_src,
                "algorithmic_bytes": rl_bytes,
                "ms": rl_iter_s * 1e3,
                # the same time priced two more ways: the 1-pass-per-FFT lower bound of SURVEY 8(d), and the bytes the
                # engine physically moved (PMC): frac is the contract's model, frac_moved the HBM utilisation
                "frac_1pass": rl_bytes_1pass / rl_iter_s / 1e9 / HBM_PEAK_GBS,
                "frac_moved": (rl_moved / rl_iter_s / 1e9 / HBM_PEAK_GBS) if rl_moved else None,
            },
            "roofline_deskew": {
                # the PAIR: everything bh_deskew launches for "deskew with a mean fill" — geometry bits, the fill value from the
                # row sums R-L's last pass left behind, deskew_kernel<..., 2> writing whole rows (fill included), and the
                # conditional mask pipeline behind it (returns at once unless the data held exact zeros)
                "kernel": "deskew + overhang fill as one pass: deskew_kernel<float, 64, 2, 3, 256, true, 2> (fused shear-interpolate + "
                          "N-mean on LDS-staged tiles, whole rows incl. the fill value; BH_DESKEW_ROWS_KERNEL=pers: the persistent "
                          "double-buffered variant) + csrc/deskew_rows.inc (geometry bits, mean from row sums) + the conditional mask "
                          "pipeline",
                "bound": "hbm",
                "achieved": dk_bytes / pair_s / 1e9,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": dk_bytes / pair_s / 1e9 / HBM_PEAK_GBS,
                "traffic": dk_moved,
                "traffic_source": dk_src,
                "frac_moved": (dk_moved / pair_s / 1e9 / HBM_PEAK_GBS) if dk_moved else None,
                "algorithmic_bytes": dk_bytes,
                "ms": pair_s * 1e3,
                "deskew_ms": deskew_s * 1e3,
                "fill_passes_ms": fill_s * 1e3,
                "row_sums": "from the last Richardson-Lucy update pass (bh_richardson_lucy_apply_rows)",
            },
            "workspace_gb": ctx.workspace_bytes() / 1e9,
            "alloc_layout": __import__("biahub_amd.device", fromlist=["alloc_layout"]).alloc_layout(),
            "rl_handle": {"backend": rl_prepared.backend, "box": list(rl_prepared.box), "otf_is_real": rl_prepared.otf_is_real,
                          "otf_gb": rl_prepared.otf_bytes / 1e9},
            "rccl": rccl,
        }
        if host_fed is not None:
            result["end_to_end_per_rank"] = host_fed
        if not args.no_ops and world == 1:
            del vols[1:]
            ctx.release_workspace()
            result["ops"] = ops_suite(vols[0], psf, dev, ctx)
        if not args.no_end_to_end and world == 1:
            del vols[1:]
            host = torch.empty(shape, dtype=torch.uint16, pin_memory=True)
            host.copy_(vols[0].to(torch.uint16))
            result["end_to_end"] = end_to_end(host, psf, rl_prepared, args.iterations, dev, ctx)
            del host
        if not args.no_cpu_baseline and world == 1:  # reported on rank 0 at N=1 only
            result["cpu_baseline"] = cpu_baseline(args.iterations)
        print(json.dumps(result), flush=True)
    if world > 1:
        parallel.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
