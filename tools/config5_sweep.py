#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: compute-tf + apply_inverse_transfer_function on a (512, 2048, 2048) float32 volume, with
the staged inverse filter kept in float32 and in bfloat16 — the fp32-vs-bf16 tolerance sweep over the regularisation
strength, the timings, and size-independent checks of the result.  `python tools/config5_sweep.py [Z Y X]` prints one JSON
line per measurement; tests/test_gpu_parity.py::test_config5_full_size_bf16_sweep runs `sweep()` and asserts on it.
(Parity unpinned: the arithmetic restates waveorder 3.0.5, absent from the reference tree.)"""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

PHASE = dict(yx_pixel_size=0.1, z_pixel_size=0.25, wavelength_illumination=0.45, z_padding=0, index_of_refraction_media=1.3,
             numerical_aperture_illumination=0.5, numerical_aperture_detection=1.2)


def sweep(shape=(512, 2048, 2048), regs=(1e-1, 1e-2, 1e-3, 1e-4), device="cuda:0", emit=print):
    from biahub_amd import _lib
    from biahub_amd.apply_inverse_transfer_function import PreparedInverseFilter, apply_inverse_transfer_function_zyx
    from biahub_amd.compute_transfer_function import phase_transfer_function_3d
    from biahub_amd.deconvolve import tikhonov_zyx
    from biahub_amd.device import get_context

    dev = torch.device(device)
    ctx = get_context(dev)
    ctx.set_timing(True)
    V = int(np.prod(shape))
    g = torch.Generator(device=dev).manual_seed(5)
    vol = torch.empty(shape, dtype=torch.float32, device=dev).normal_(500.0, 40.0, generator=g).clamp_(1.0)
    out = {"shape": list(shape), "rows": []}
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    H, Him = phase_transfer_function_3d(shape, PHASE["yx_pixel_size"], PHASE["z_pixel_size"], PHASE["wavelength_illumination"],
                                        PHASE["z_padding"], PHASE["index_of_refraction_media"],
                                        PHASE["numerical_aperture_illumination"], PHASE["numerical_aperture_detection"])
    torch.cuda.synchronize(dev)
    out["compute_tf_s"] = time.perf_counter() - t0
    del Him
    ctx.release_workspace()  # the three complex work volumes of compute-tf
    emit(json.dumps({"step": "compute-tf (phase, 3-D)", "seconds": out["compute_tf_s"], "voxels": V}))
    for reg in regs:
        a = apply_inverse_transfer_function_zyx(vol, H, 0, reg, True, "f32")
        ms32 = ctx.elapsed_ms(_lib.T_TIKHONOV)
        b = apply_inverse_transfer_function_zyx(vol, H, 0, reg, True, "bf16")
        ms16 = ctx.elapsed_ms(_lib.T_TIKHONOV)
        prep = PreparedInverseFilter(H, shape, 0, reg, "f32", dev)   # the per-position form: staged once, applied per volume
        for _ in range(2):
            c = prep(vol, True)
            ms_prep = ctx.elapsed_ms(_lib.T_TIKHONOV)
        assert torch.equal(c, a)
        prep.close()
        del c
        amax = float(a.abs().max())
        err = float((a - b).abs().max()) / amax
        rms = float(((a - b) ** 2).mean().sqrt()) / float((a ** 2).mean().sqrt())
        row = {"regularization_strength": reg, "ms_f32": ms32, "ms_bf16": ms16, "ms_f32_prepared": ms_prep,
               "voxels_per_s_f32_prepared": V / (ms_prep / 1e3), "max_rel_err_bf16_vs_f32": err,
               "rms_rel_err_bf16_vs_f32": rms, "mean_over_std": float(a.mean().abs() / a.std()),
               "voxels_per_s_f32": V / (ms32 / 1e3), "voxels_per_s_bf16": V / (ms16 / 1e3)}
        out["rows"].append(row)
        emit(json.dumps({"step": "apply-inv-tf", **row}))
        del a, b
    # linearity (the operator without the mean normalisation is linear) and agreement with the reference's deconvolve
    # operator (bh_tikhonov) for a real transfer function
    Hr = H.abs().to(torch.float32)
    del H
    x1 = vol
    x2 = torch.roll(vol, shifts=(3, 17, 101), dims=(0, 1, 2))
    t1 = apply_inverse_transfer_function_zyx(x1, Hr, 0, 1e-2, False)
    t2 = apply_inverse_transfer_function_zyx(x2, Hr, 0, 1e-2, False)
    mix = x1 * 0.25
    mix += x2 * 1.5
    t12 = apply_inverse_transfer_function_zyx(mix, Hr, 0, 1e-2, False)
    del mix
    ref = t1 * 0.25
    ref += t2 * 1.5
    out["linearity_err"] = float((t12 - ref).abs().max()) / float(ref.abs().max())
    del t12, ref, t2
    tk = tikhonov_zyx(x1, Hr, 1e-2)
    out["vs_bh_tikhonov"] = float((t1 - tk).abs().max()) / float(tk.abs().max())
    # translation covariance: a circular shift of the input shifts the output (size-independent, exact up to rounding)
    t2 = apply_inverse_transfer_function_zyx(x2, Hr, 0, 1e-2, False)
    out["shift_err"] = float((torch.roll(t1, shifts=(3, 17, 101), dims=(0, 1, 2)) - t2).abs().max()) / float(t1.abs().max())
    emit(json.dumps({"step": "properties", "linearity_err": out["linearity_err"], "vs_bh_tikhonov": out["vs_bh_tikhonov"],
                     "shift_err": out["shift_err"]}))
    return out


if __name__ == "__main__":
    shp = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (512, 2048, 2048)
    sweep(shp)
