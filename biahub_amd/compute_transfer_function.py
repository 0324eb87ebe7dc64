"""compute-tf on MI355X — host-side mirror of ``biahub/compute_transfer_function.py`` (a pass-through to waveorder's
``compute_transfer_function_cli``) and of the waveorder functions behind it.

waveorder 3.0.5 (reference ``uv.lock:6062-6063``) is not part of the reference tree, so the optics below restate its
published algorithm as recalled (``waveorder/models/phase_thick_3d.py``, ``isotropic_fluorescent_thick_3d.py``,
``optics.py``, ``sampling.py``, ``cli/compute_transfer_function.py``): **parity unpinned**.  The transfer functions are built on
the GPU (``csrc/invtf.hip``: pupil / propagation / Green's-function planes, hipFFT complex transforms); this module only
carries the reference's calling convention, the Nyquist oversampling rule and the store layout of the transfer-function zarr.
"""

from __future__ import annotations

import math
from pathlib import Path

import numpy as np
import torch

from . import _lib
from .device import get_context, ptr, resolve_device, to_host
from .settings import ReconstructionSettings
from .utils.config import yaml_to_model


def transverse_nyquist(wavelength_emission, numerical_aperture_illumination, numerical_aperture_detection) -> float:
    """waveorder ``sampling.transverse_nyquist``: the finest transverse period the system passes, halved."""
    return wavelength_emission / (2 * (numerical_aperture_detection + numerical_aperture_illumination))


def axial_nyquist(wavelength_emission, numerical_aperture_detection, index_of_refraction_media) -> float:
    """waveorder ``sampling.axial_nyquist``."""
    n_on_lambda = index_of_refraction_media / wavelength_emission
    cutoff = n_on_lambda - math.sqrt(n_on_lambda**2 - (numerical_aperture_detection / wavelength_emission) ** 2)
    return 1.0 / (2 * cutoff)


def _central_cuboid(ctx, src: torch.Tensor, shape) -> torch.Tensor:
    if tuple(src.shape) == tuple(shape):
        return src
    dst = torch.empty(tuple(shape), dtype=torch.complex64, device=src.device)
    _lib.check(ctx.lib.bh_fourier_central_cuboid(ctx.handle, ptr(src), *(int(n) for n in src.shape), ptr(dst),
                                                 *(int(n) for n in shape)))
    return dst


def phase_transfer_function_3d(zyx_shape, yx_pixel_size, z_pixel_size, wavelength_illumination, z_padding,
                               index_of_refraction_media, numerical_aperture_illumination, numerical_aperture_detection,
                               invert_phase_contrast=False, device="cuda"):
    """``phase_thick_3d.calculate_transfer_function``: (real, imaginary) potential transfer functions, complex64 device
    tensors of shape ``(Z + 2 z_padding, Y, X)``.  A pixel size above Nyquist is handled as waveorder does: the functions are
    computed on a grid refined by an integer factor and the central Fourier cuboid is kept."""
    dev = resolve_device(device)
    ctx = get_context(dev)
    Z, Y, X = (int(s) for s in zyx_shape)
    yx_factor = int(math.ceil(yx_pixel_size / transverse_nyquist(wavelength_illumination, numerical_aperture_illumination,
                                                                   numerical_aperture_detection)))
    z_factor = int(math.ceil(z_pixel_size / axial_nyquist(wavelength_illumination, numerical_aperture_detection,
                                                           index_of_refraction_media)))
    fz, fy, fx = Z * z_factor, Y * yx_factor, X * yx_factor
    with torch.cuda.device(dev):
        re = torch.empty((fz + 2 * z_padding, fy, fx), dtype=torch.complex64, device=dev)
        im = torch.empty_like(re)
        _lib.check(ctx.lib.bh_phase_transfer_function_3d(
            ctx.handle, fz, fy, fx, float(yx_pixel_size) / yx_factor, float(z_pixel_size) / z_factor,
            float(wavelength_illumination), int(z_padding), float(index_of_refraction_media),
            float(numerical_aperture_illumination), float(numerical_aperture_detection), int(bool(invert_phase_contrast)),
            ptr(re), ptr(im)))
        out_shape = (Z + 2 * z_padding, Y, X)
        return _central_cuboid(ctx, re, out_shape), _central_cuboid(ctx, im, out_shape)


def fluorescence_transfer_function_3d(zyx_shape, yx_pixel_size, z_pixel_size, wavelength_emission, z_padding,
                                      index_of_refraction_media, numerical_aperture_detection, device="cuda"):
    """``isotropic_fluorescent_thick_3d.calculate_transfer_function``: the optical transfer function
    ``fftn(|ifft2(pupil * propagation kernel)|^2) / max``, complex64 ``(Z + 2 z_padding, Y, X)``."""
    dev = resolve_device(device)
    ctx = get_context(dev)
    Z, Y, X = (int(s) for s in zyx_shape)
    yx_factor = int(math.ceil(yx_pixel_size / transverse_nyquist(wavelength_emission, numerical_aperture_detection,
                                                                   numerical_aperture_detection)))
    z_factor = int(math.ceil(z_pixel_size / axial_nyquist(wavelength_emission, numerical_aperture_detection,
                                                           index_of_refraction_media)))
    fz, fy, fx = Z * z_factor, Y * yx_factor, X * yx_factor
    with torch.cuda.device(dev):
        otf = torch.empty((fz + 2 * z_padding, fy, fx), dtype=torch.complex64, device=dev)
        _lib.check(ctx.lib.bh_fluorescence_transfer_function_3d(
            ctx.handle, fz, fy, fx, float(yx_pixel_size) / yx_factor, float(z_pixel_size) / z_factor,
            float(wavelength_emission), int(z_padding), float(index_of_refraction_media),
            float(numerical_aperture_detection), ptr(otf)))
        return _central_cuboid(ctx, otf, (Z + 2 * z_padding, Y, X))


def _refuse_unsupported(settings: ReconstructionSettings) -> None:
    if settings.birefringence is not None:
        raise NotImplementedError("birefringence reconstruction is not part of biahub_amd (phase and fluorescence are); use "
                                  "the reference biahub / waveorder for it")
    if settings.reconstruction_dimension != 3:
        raise NotImplementedError("biahub_amd reconstructs 3-D volumes (reconstruction_dimension: 3) only")
    for part in (settings.phase, settings.fluorescence):
        if part is not None and part.apply_inverse.reconstruction_algorithm != "Tikhonov":
            raise NotImplementedError("only the Tikhonov reconstruction_algorithm runs in biahub_amd")
    if len(settings.input_channel_names) != 1:
        raise ValueError("3-D phase / fluorescence reconstruction takes exactly one input channel")


def pixel_sizes(tf_settings, input_scale):
    """(yx, z) pixel sizes: the config is the source of truth, the input store's scale fills what it leaves out
    (biahub/apply_inverse_transfer_function.py:51-54)."""
    yx = tf_settings.yx_pixel_size if tf_settings.yx_pixel_size is not None else float(input_scale[-1])
    z = tf_settings.z_pixel_size if tf_settings.z_pixel_size is not None else float(input_scale[-3])
    return float(yx), float(z)


def compute_transfer_function_cli(input_position_dirpath, config_filepath, output_dirpath, device="cuda") -> None:
    """waveorder ``compute_transfer_function_cli(input_position_dirpath, config_filepath, output_dirpath)`` as
    biahub/compute_transfer_function.py:37 and reconstruct.py:59-64 call it: the transfer function for the shape of that
    position, written to a transfer-function store (named arrays of shape ``(1, 1, Z', Y, X)``, settings in the attributes)."""
    from .io import create_empty_fov, open_ome_zarr

    settings = yaml_to_model(config_filepath, ReconstructionSettings)
    _refuse_unsupported(settings)
    with open_ome_zarr(input_position_dirpath) as ds:
        zyx_shape, scale, names = tuple(ds.data.shape[-3:]), ds.scale, ds.channel_names
    for ch in settings.input_channel_names:
        if ch not in names:
            raise ValueError(f"Channel {ch} not found in the input data (channels: {names})")
    store = create_empty_fov(Path(output_dirpath), ["None"], metadata={"settings": settings.model_dump()})
    if settings.phase is not None:
        tfs = settings.phase.transfer_function
        yx, z = pixel_sizes(tfs, scale)
        re, im = phase_transfer_function_3d(zyx_shape, yx, z, tfs.wavelength_illumination, tfs.z_padding,
                                            tfs.index_of_refraction_media, tfs.numerical_aperture_illumination,
                                            tfs.numerical_aperture_detection, tfs.invert_phase_contrast, device)
        store.create_image("real_potential_transfer_function", to_host(torch.view_as_real(re)).view(np.complex64)[None, None, ..., 0])
        store.create_image("imaginary_potential_transfer_function", to_host(torch.view_as_real(im)).view(np.complex64)[None, None, ..., 0])
    if settings.fluorescence is not None:
        tfs = settings.fluorescence.transfer_function
        yx, z = pixel_sizes(tfs, scale)
        otf = fluorescence_transfer_function_3d(zyx_shape, yx, z, tfs.wavelength_emission, tfs.z_padding,
                                                tfs.index_of_refraction_media, tfs.numerical_aperture_detection, device)
        store.create_image("optical_transfer_function", to_host(torch.view_as_real(otf)).view(np.complex64)[None, None, ..., 0])
