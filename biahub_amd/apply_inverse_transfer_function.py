"""apply-inv-tf on MI355X — host-side mirror of ``biahub/apply_inverse_transfer_function.py`` and of the waveorder
functions it submits (``apply_inverse_transfer_function_single_position``, ``get_reconstruction_output_metadata``,
``estimate_resources``).

Per volume the step is ``crop_z(Re ifftn(fftn(pad_z(n(x))) * conj(H) / (|H|^2 + reg)))`` — ``bh_inverse_filter``
(``csrc/invtf.hip``): on shapes the fused FFT engine takes the filter is multiplied inside its Z pass (5 passes, 48 + 8
bytes per voxel; ``filter_storage="bf16"`` keeps the staged filter as bfloat16 pairs), other shapes go through hipFFT.
waveorder 3.0.5 is absent from the reference tree: the arithmetic restates its published algorithm — **parity unpinned**.
"""

from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from . import _lib
from .compute_transfer_function import _refuse_unsupported, pixel_sizes
from .device import as_device_volume, get_context, ptr, resolve_device, to_host
from .device import empty as device_empty, empty_like as device_empty_like
from .settings import ReconstructionSettings
from .utils.config import yaml_to_model

_STORAGE = {"f32": _lib.FILTER_F32, "bf16": _lib.FILTER_BF16}


def _tf_tensor(transfer_function, dev) -> torch.Tensor:
    H = transfer_function if isinstance(transfer_function, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(transfer_function))
    H = H.to(dev)
    if H.dtype not in (torch.complex64, torch.float32):
        H = H.to(torch.complex64 if H.is_complex() else torch.float32)
    return H.contiguous()


class PreparedInverseFilter:
    """The inverse filter of one transfer function, staged once on the GPU (``bh_inverse_filter_create``) and applied to any
    number of volumes of one shape — what the per-position job needs: waveorder reads the transfer function once per
    position and reconstructs every time point with it.  Staging is a quarter of a one-shot call."""

    def __init__(self, transfer_function, zyx_shape, z_padding: int = 0, regularization_strength: float = 1e-3,
                 filter_storage: str = "f32", device="cuda"):
        import ctypes

        self.device = resolve_device(device)
        H = _tf_tensor(transfer_function, self.device)
        Z, Y, X = (int(s) for s in zyx_shape)
        if H.ndim != 3 or tuple(H.shape) != (Z + 2 * int(z_padding), Y, X):
            raise ValueError(f"transfer function shape {tuple(H.shape)} != padded data shape {(Z + 2 * int(z_padding), Y, X)}")
        if filter_storage not in _STORAGE:
            raise ValueError(f"filter_storage {filter_storage!r}: 'f32' or 'bf16'")
        self.shape = (Z, Y, X)
        self._ctx = get_context(self.device)
        self._handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._ctx.lib.bh_inverse_filter_create(self._ctx.handle, ptr(H), int(H.is_complex()), Z, Y, X,
                                                              int(z_padding), float(regularization_strength),
                                                              _STORAGE[filter_storage], ctypes.byref(self._handle)))
            torch.cuda.current_stream(self.device).synchronize()  # H may be dropped by the caller from here on

    def __call__(self, zyx, normalize: bool = False) -> torch.Tensor:
        x, code, _ = as_device_volume(zyx, self.device)
        if code != _lib.DT_F32:
            x = x.to(torch.float32)
        if tuple(x.shape) != self.shape:
            raise ValueError(f"volume shape {tuple(x.shape)} != the shape this filter was prepared for {self.shape}")
        ctx = get_context(self.device)
        with torch.cuda.device(self.device):
            out = device_empty_like(x)
            _lib.check(ctx.lib.bh_inverse_filter_apply(ctx.handle, self._handle, ptr(x), int(bool(normalize)), ptr(out)))
        return out

    def close(self) -> None:
        if self._handle:
            torch.cuda.synchronize(self.device)
            _lib.check(self._ctx.lib.bh_inverse_filter_destroy(self._handle))
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


def apply_inverse_transfer_function_zyx(zyx, transfer_function, z_padding: int = 0, regularization_strength: float = 1e-3,
                                        normalize: bool = False, filter_storage: str = "f32") -> torch.Tensor:
    """One volume on device.  ``transfer_function``: ``(Z + 2 z_padding, Y, X)`` in natural FFT order, complex64 (phase) or
    float32 (a real optical transfer function); ``normalize`` applies ``x / mean(x) - 1`` first (phase)."""
    x, code, dev = as_device_volume(zyx)
    if code != _lib.DT_F32:
        x = x.to(torch.float32)
    H = transfer_function if isinstance(transfer_function, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(transfer_function))
    H = H.to(dev)
    if H.dtype not in (torch.complex64, torch.float32):
        H = H.to(torch.complex64 if H.is_complex() else torch.float32)
    H = H.contiguous()
    if x.ndim != 3 or H.ndim != 3:
        raise ValueError("volume and transfer function must be 3-D")
    Z, Y, X = (int(s) for s in x.shape)
    if tuple(H.shape) != (Z + 2 * int(z_padding), Y, X):
        raise ValueError(f"transfer function shape {tuple(H.shape)} != padded data shape {(Z + 2 * int(z_padding), Y, X)}")
    if filter_storage not in _STORAGE:
        raise ValueError(f"filter_storage {filter_storage!r}: 'f32' or 'bf16'")
    ctx = get_context(dev)
    with torch.cuda.device(dev):
        out = device_empty_like(x)
        _lib.check(ctx.lib.bh_inverse_filter(ctx.handle, ptr(x), ptr(H), int(H.is_complex()), Z, Y, X, int(z_padding),
                                             float(regularization_strength), int(bool(normalize)), _STORAGE[filter_storage],
                                             ptr(out)))
    return out


def apply_inverse_transfer_function_czyx(czyx_data: np.ndarray, transfer_function=None, z_padding: int = 0,
                                         regularization_strength: float = 1e-3, normalize: bool = False,
                                         absorption_ratio: float = 0.0, imaginary_transfer_function=None,
                                         filter_storage: str = "f32", device="cuda", prepared: PreparedInverseFilter | None = None
                                         ) -> np.ndarray:
    """Operator ``func(czyx, **kwargs) -> czyx`` of the per-position driver: the reconstruction of the (single) input channel.
    Phase: ``transfer_function`` = real potential transfer function, ``normalize=True``; a non-zero ``absorption_ratio`` adds
    that fraction of ``imaginary_transfer_function`` (waveorder's effective transfer function).  ``prepared``: a
    ``PreparedInverseFilter`` staged by the caller (the per-position job stages one for all its time points)."""
    dev = resolve_device(device)
    if prepared is not None:
        return np.stack([to_host(prepared(torch.from_numpy(np.ascontiguousarray(zyx)).to(dev), normalize))
                         for zyx in np.asarray(czyx_data)])
    H = transfer_function if isinstance(transfer_function, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(transfer_function))
    H = H.to(dev)
    if absorption_ratio:
        Hi = imaginary_transfer_function
        Hi = Hi if isinstance(Hi, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(Hi))
        H = H + float(absorption_ratio) * Hi.to(dev)
    out = [to_host(apply_inverse_transfer_function_zyx(torch.from_numpy(np.ascontiguousarray(zyx)).to(dev), H, z_padding,
                                                       regularization_strength, normalize, filter_storage))
           for zyx in np.asarray(czyx_data)]
    return np.stack(out)


def estimate_resources(shape, settings: ReconstructionSettings, num_processes: int):
    """waveorder ``cli.utils.estimate_resources`` (recalled): (num_cpus, GB of RAM per CPU) of one position's job."""
    T, C, Z, Y, X = shape
    gb_per_element = 4 / 2**30
    input_memory = Z * Y * X * gb_per_element
    gb = 0.0
    if settings.birefringence is not None:
        gb += input_memory * 4
    if settings.phase is not None:
        gb += input_memory * 32
    if settings.fluorescence is not None:
        gb += input_memory * 32
    gb_ram_per_cpu = int(np.ceil(max(1.0, gb)))
    return int(min(32, num_processes)), gb_ram_per_cpu


def get_reconstruction_output_metadata(input_position_path, config_filepath) -> dict:
    """waveorder ``get_reconstruction_output_metadata``: shape, scale, channel names, dtype, version of the output plate.
    Pixel sizes come from the config where it gives them (a >5 % disagreement with the input scale is warned about)."""
    import warnings

    from .io import open_ome_zarr

    settings = yaml_to_model(config_filepath, ReconstructionSettings)
    with open_ome_zarr(input_position_path) as ds:
        T, C, Z, Y, X = ds.data.shape
        scale, version, names = list(ds.scale), ds.version, ds.channel_names
    if settings.time_indices == "all":
        times = list(range(T))
    elif isinstance(settings.time_indices, int):
        times = [settings.time_indices]
    else:
        times = list(settings.time_indices)
    tfs = (settings.phase or settings.fluorescence).transfer_function if (settings.phase or settings.fluorescence) else None
    out_scale = scale
    if tfs is not None:
        yx, z = pixel_sizes(tfs, scale)
        for got, want, what in ((scale[-1], yx, "yx_pixel_size"), (scale[-3], z, "z_pixel_size")):
            if abs(got - want) > 0.05 * want:
                warnings.warn(f"config {what}={want} differs from the input zarr scale ({got:.4f}) by more than 5 %")
        out_scale = scale[:2] + [z, yx, yx]
    zdim = Z if settings.reconstruction_dimension == 3 else 1
    return {"shape": (len(times), len(settings.output_channel_names), zdim, Y, X), "chunks": None, "scale": out_scale,
            "channel_names": settings.output_channel_names, "dtype": np.float32, "version": version,
            "input_channel_names": names, "time_indices": times}


def apply_inverse_transfer_function_single_position(input_position_dirpath, transfer_function_dirpath, config_filepath,
                                                    output_position_dirpath, num_processes: int = 1,
                                                    output_channel_names=None, filter_storage: str = "f32",
                                                    device="cuda") -> None:
    """The per-position job of biahub/apply_inverse_transfer_function.py:158-170 (waveorder's function of this name): every
    time point of the configured input channel is reconstructed with the stored transfer function and written to the
    output position.  The transfer function is read and uploaded once per position."""
    from .io import open_ome_zarr, process_single_position

    settings = yaml_to_model(config_filepath, ReconstructionSettings)
    _refuse_unsupported(settings)
    dev = resolve_device(device)
    meta = get_reconstruction_output_metadata(input_position_dirpath, config_filepath)
    tf_store = open_ome_zarr(transfer_function_dirpath)
    in_c = meta["input_channel_names"].index(settings.input_channel_names[0])
    kw: dict = {"filter_storage": filter_storage, "device": dev}
    if settings.phase is not None:
        H = torch.from_numpy(np.ascontiguousarray(tf_store["real_potential_transfer_function"][0, 0])).to(dev)
        kw.update(transfer_function=H, z_padding=settings.phase.transfer_function.z_padding, normalize=True,
                  regularization_strength=settings.phase.apply_inverse.regularization_strength)
    else:
        H = torch.from_numpy(np.ascontiguousarray(tf_store["optical_transfer_function"][0, 0])).to(dev)
        kw.update(transfer_function=H, z_padding=settings.fluorescence.transfer_function.z_padding, normalize=False,
                  regularization_strength=settings.fluorescence.apply_inverse.regularization_strength)
    times = meta["time_indices"]
    with open_ome_zarr(input_position_dirpath) as ds:
        zyx_shape = tuple(ds.data.shape[-3:])
    # the inverse filter is staged once for the position and applied to every time point
    prepared = PreparedInverseFilter(kw.pop("transfer_function"), zyx_shape, kw.pop("z_padding"), kw.pop("regularization_strength"),
                                     kw.pop("filter_storage"), dev)
    del H
    try:
        process_single_position(apply_inverse_transfer_function_czyx, input_position_dirpath, output_position_dirpath,
                                input_channel_indices=[[in_c]], output_channel_indices=[[0]], input_time_indices=times,
                                output_time_indices=list(range(len(times))), prepared=prepared, **kw)
    finally:
        prepared.close()
