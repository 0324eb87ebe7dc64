"""``process-with-config`` operators — mirror of ``biahub/process_data.py``.

``binning_czyx`` (process_data.py:29-105) sums or averages ``(bz, by, bx)`` windows, stretches the result to the dtype range
by the reference's rules and casts back to the input dtype.  The windows and the range scan run in ``csrc/binning.hip``
(``bh_bin_reduce``), the affine rescale + cast in ``bh_bin_finish`` with the reference's float32 operation order.
``process_czyx`` chains configured functions; only functions registered here run on the GPU, the reference's
``np.*`` whitelist (cli/resolve_function.py) is not re-exported.
"""

from __future__ import annotations

import ctypes as C
from typing import Literal, Sequence

import numpy as np
import torch

from . import _lib
from .device import _NP_TO_DT, as_device_volume, get_context, ptr, to_host

_TORCH_OUT = {_lib.DT_U8: torch.uint8, _lib.DT_U16: torch.uint16, _lib.DT_I16: torch.int16, _lib.DT_F32: torch.float32}


def _bin_reduce(vol: torch.Tensor, code: int, factor, mean: bool):
    ctx = get_context(vol.device)
    Z, Y, X = (int(s) for s in vol.shape)
    f = [int(v) for v in factor]
    with torch.cuda.device(vol.device):
        out = torch.empty((Z // f[0], Y // f[1], X // f[2]), dtype=torch.float32, device=vol.device)
    mm = (C.c_float * 2)()
    _lib.check(ctx.lib.bh_bin_reduce(ctx.handle, ptr(vol), code, Z, Y, X, (C.c_int * 3)(*f), int(mean), ptr(out), mm))
    return out, float(mm[0]), float(mm[1])


def _bin_finish(v: torch.Tensor, apply: bool, sub: float, mul: float, div: float, out_code: int) -> torch.Tensor:
    ctx = get_context(v.device)
    with torch.cuda.device(v.device):
        out = torch.empty(v.shape, dtype=_TORCH_OUT[out_code], device=v.device)
    _lib.check(ctx.lib.bh_bin_finish(ctx.handle, ptr(v), v.numel(), int(apply), float(sub), float(mul), float(div), out_code,
                                     ptr(out)))
    return out


def binning_czyx(czyx_data: np.ndarray, binning_factor_zyx: Sequence[int] = [1, 2, 2],
                 mode: Literal["sum", "mean"] = "sum", device="cuda") -> np.ndarray:
    """Binning via summing or averaging pixels within bin windows (process_data.py:29-105); same dtype out as in."""
    a = np.asarray(czyx_data)
    if mode not in ("sum", "mean"):
        raise ValueError(f"Invalid mode: {mode}. Must be 'sum' or 'mean'.")
    if a.dtype not in _NP_TO_DT:
        raise TypeError(f"binning_czyx: dtype {a.dtype} is not supported on the GPU path (uint8, uint16, int16, float32)")
    code = _NP_TO_DT[a.dtype]
    is_int = np.issubdtype(a.dtype, np.integer)
    binned = []
    for c in range(a.shape[0]):
        t, _, dev = as_device_volume(a[c], device)
        v, mn, mx = _bin_reduce(t, code, binning_factor_zyx, mean=(mode == "mean"))
        binned.append((v, mn, mx))
    outs = []
    if mode == "sum":
        for v, mn, mx in binned:  # per channel: stretch [min, max] to [0, max_val] when max > 0 (:80-90)
            max_val = float(np.iinfo(a.dtype).max if is_int else np.iinfo(np.uint16).max)
            outs.append(_bin_finish(v, mx > 0, mn, max_val, np.float32(mx) - np.float32(mn), code))
    else:
        gmax = max(b[2] for b in binned)  # integer dtypes: scaled by the maximum over ALL channels (:97-99)
        for v, mn, mx in binned:
            outs.append(_bin_finish(v, is_int, 0.0, float(np.iinfo(a.dtype).max) if is_int else 1.0, gmax if is_int else 1.0,
                                    code))
    return np.stack([to_host(o) for o in outs])


CUSTOM_FUNCTIONS = {
    "biahub.process_data.binning_czyx": binning_czyx,
    "biahub_amd.process_data.binning_czyx": binning_czyx,
}


def resolve_function(function_name: str, custom_functions: dict = None):
    """``cli/resolve_function.py:26-66`` restricted to the functions this package runs on the GPU."""
    table = dict(CUSTOM_FUNCTIONS)
    table.update(custom_functions or {})
    if function_name not in table:
        raise ValueError(f"Function '{function_name}' not found; available: {sorted(table)}")
    return table[function_name]


def process_czyx(czyx_data: np.ndarray, processing_functions: list) -> np.ndarray:
    """Apply the configured functions in order (process_data.py:113-145)."""
    for proc in processing_functions:
        func = resolve_function(proc.function)
        if proc.input_channels is None or len(proc.input_channels) != 1:
            raise ValueError("Only one input channel is supported for now")
        czyx_data = func(czyx_data, **proc.kwargs)
    return czyx_data
