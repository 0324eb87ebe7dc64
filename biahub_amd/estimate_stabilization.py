"""Phase cross-correlation on MI355X — mirror of ``biahub/estimate_stabilization.py:199-256``.

The FFT kernel of the stabilisation *estimate*  is ``bh_phase_cross_corr``: two R2C transforms,
the fused normalised conjugate product, C2R, |.| + first-occurrence argmax on device.  Around it this module keeps the
reference's phase-cross-correlation call chain (``phase_cross_corr_padding``, ``get_tform_from_pcc``,
``estimate_xyz_stabilization_pcc[_per_position]``: estimate_stabilization.py:129-196, 259-310, 444-693) so that
``estimate-stabilization`` with ``stabilization_method: phase-cross-corr`` produces the settings ``stabilize`` consumes;
focus finding, StackReg and bead matching are not part of this package.
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .device import as_device_volume, get_context, ptr, resolve_device, to_host


def phase_cross_corr_device(ref_img, mov_img, normalization=None, device="cuda", want_corr: bool = True):
    """Device-level correlation: ``(shift float32[3], corr tensor or None)``.  With ``want_corr=False`` the correlation
    volume is neither written nor copied (a 1-GB volume costs ~80 ms to bring to the host; the shift costs nothing)."""
    if normalization not in _lib.PCC_NORM:
        raise ValueError(f"unknown normalization {normalization!r}")
    dev = resolve_device(device if not isinstance(ref_img, torch.Tensor) else ref_img.device)
    a, _, _ = as_device_volume(ref_img, dev)
    b, _, _ = as_device_volume(mov_img, dev)
    a, b = a.to(torch.float32), b.to(torch.float32)
    if a.ndim != 3 or a.shape != b.shape:
        raise ValueError(f"expected two 3-D images of one shape, got {tuple(a.shape)} and {tuple(b.shape)}")
    Z, Y, X = (int(s) for s in a.shape)
    ctx = get_context(dev)
    shift = (C.c_float * 3)()
    with torch.cuda.device(dev):
        # irfftn without a shape: an odd last axis comes back one shorter (the reference's behaviour, kept)
        corr = torch.empty((Z, Y, X - (X & 1)), dtype=torch.float32, device=dev) if want_corr else None
        _lib.check(ctx.lib.bh_phase_cross_corr(ctx.handle, ptr(a), ptr(b), Z, Y, X, _lib.PCC_NORM[normalization],
                                               shift, ptr(corr) if want_corr else None))
    return np.array(list(shift), dtype=np.float32), corr


class PreparedPhaseCrossCorr:
    """``phase_cross_corr`` with ONE of the two images fixed (``bh_phase_cross_corr_create``): its spectrum is computed once, a
    call transforms the other image only.  The stabilisation estimate correlates every timepoint with the first one — or, with
    ``roll=True``, with its predecessor: the image of a call then becomes the stored one, its spectrum a by-product of the
    call's own Z pass (estimate_stabilization.py:505-520).  ``fixed_is_second``: the stored image is ``phase_cross_corr``'s
    second argument (``mov_img``), as in ``get_tform_from_pcc``, which passes the varying timepoint first."""

    def __init__(self, fixed_img, fixed_is_second: bool = False, device="cuda"):
        self.device = resolve_device(device if not isinstance(fixed_img, torch.Tensor) else fixed_img.device)
        a, _, _ = as_device_volume(fixed_img, self.device)
        a = a.to(torch.float32)
        if a.ndim != 3:
            raise ValueError(f"expected a 3-D image, got shape {tuple(a.shape)}")
        self.shape = tuple(int(n) for n in a.shape)
        self.fixed_is_second = bool(fixed_is_second)
        self._ctx = get_context(self.device)
        self._handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._ctx.lib.bh_phase_cross_corr_create(self._ctx.handle, ptr(a), *self.shape, int(self.fixed_is_second),
                                                                C.byref(self._handle)))
            torch.cuda.current_stream(self.device).synchronize()  # the image may be dropped by the caller from here on

    def __call__(self, img, normalization=None, want_corr: bool = False, roll: bool = False):
        """``(shift float32[3], fftshift(|corr|) tensor or None)`` of ``phase_cross_corr(fixed, img)`` — of
        ``phase_cross_corr(img, fixed)`` when ``fixed_is_second``."""
        if normalization not in _lib.PCC_NORM:
            raise ValueError(f"unknown normalization {normalization!r}")
        b, _, _ = as_device_volume(img, self.device)
        b = b.to(torch.float32)
        if tuple(b.shape) != self.shape:
            raise ValueError(f"image shape {tuple(b.shape)} != the shape this handle was prepared for {self.shape}")
        Z, Y, X = self.shape
        shift = (C.c_float * 3)()
        with torch.cuda.device(self.device):
            corr = torch.empty((Z, Y, X - (X & 1)), dtype=torch.float32, device=self.device) if want_corr else None
            _lib.check(self._ctx.lib.bh_phase_cross_corr_apply(self._ctx.handle, self._handle, ptr(b), _lib.PCC_NORM[normalization],
                                                               int(bool(roll)), shift, ptr(corr) if want_corr else None))
        return np.array(list(shift), dtype=np.float32), corr

    def close(self) -> None:
        if self._handle:
            torch.cuda.synchronize(self.device)
            _lib.check(self._ctx.lib.bh_phase_cross_corr_destroy(self._handle))
            self._handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


def phase_cross_corr(ref_img, mov_img, normalization=None, output_path=None, verbose: bool = False, device="cuda"):
    """Translation between two equally shaped 3-D images: ``(shift, corr_shifted)`` like the reference.

    ``shift`` is a float32 array (the signed location of max |corr|), ``corr_shifted = fftshift(|corr|)``.
    ``normalization`` is ``None``, ``"magnitude"`` or ``"classic"`` (estimate_stabilization.py:233-238).
    """
    shift, corr = phase_cross_corr_device(ref_img, mov_img, normalization, device, want_corr=True)
    return shift, to_host(corr)


# ----------------------------------------------------------------------------- padding variant and the per-position chain
def pad_to_shape(arr, shape, mode: str, **kwargs):
    """``registration/utils.py:856-896``: centred ``np.pad`` up to ``shape``."""
    assert arr.ndim == len(shape)
    dif = tuple(s - a for s, a in zip(shape, arr.shape))
    assert all(d >= 0 for d in dif)
    return np.pad(arr, pad_width=[[s // 2, s - s // 2] for s in dif], mode=mode, **kwargs)


def center_crop(arr, shape):
    """``registration/utils.py:899-924``."""
    assert arr.ndim == len(shape)
    starts = tuple((cur - s) // 2 for cur, s in zip(arr.shape, shape))
    assert all(s >= 0 for s in starts)
    return arr[tuple(slice(s, s + d) for s, d in zip(starts, shape))]


def match_shape(img, shape):
    """``registration/utils.py:927-958``: reflect-pad the short axes, centre-crop the long ones."""
    if np.any(np.asarray(shape) > np.asarray(img.shape)):
        img = pad_to_shape(img, tuple(np.maximum(img.shape, shape)), mode="reflect")
    if np.any(np.asarray(shape) < np.asarray(img.shape)):
        img = center_crop(img, shape)
    return img


def phase_cross_corr_padding(ref_img, mov_img, maximum_shift: float = 1.2, normalization=None, output_path=None,
                             verbose: bool = False, device="cuda"):
    """``estimate_stabilization.py:129-196``: both images padded / cropped to ``next_fast_len(max(s) * maximum_shift)``, then
    the same correlation; returns ``(peak, fftshift(|corr|))`` with ``peak = s // 2 - argmax`` (note the sign convention
    differs from ``phase_cross_corr``)."""
    from scipy.fftpack import next_fast_len  # 5-smooth sizes, as the reference imports it

    ref_img, mov_img = np.asarray(ref_img), np.asarray(mov_img)
    shape = tuple(int(next_fast_len(int(max(s1, s2) * maximum_shift))) for s1, s2 in zip(ref_img.shape, mov_img.shape))
    if verbose:
        print(f"phase cross corr. fft shape of {shape} for arrays of shape {ref_img.shape} and {mov_img.shape} "
              f"with maximum shift of {maximum_shift}")
    ref_img = np.ascontiguousarray(match_shape(ref_img, shape))
    mov_img = np.ascontiguousarray(match_shape(mov_img, shape))
    shift, corr = phase_cross_corr(ref_img, mov_img, normalization=normalization, device=device)
    # argmax position m of |corr| (signed shift folded back), seen through fftshift: p = (m + s // 2) % s
    peak = tuple(int(s // 2 - ((int(sh) % s) + s // 2) % s) for s, sh in zip(corr.shape, shift))
    if verbose:
        print(f"phase cross corr. peak at {peak}")
    return peak, corr


def get_tform_from_pcc(t: int, source_channel_tzyx, target_channel_tzyx, function_type: str = "custom", normalization=None,
                       output_path=None, verbose: bool = False, device="cuda", want_corr: bool = True):
    """``estimate_stabilization.py:259-310``, kept as written: the image named ``target`` is read from the *source* stack
    and vice versa, and the shift lands as ``transform[0, 3] = dx, [1, 3] = dy, [2, 3] = dz``."""
    target = np.asarray(source_channel_tzyx[t]).astype(np.float32)
    source = np.asarray(target_channel_tzyx[t]).astype(np.float32)
    if function_type == "custom_padding":
        shift, corr = phase_cross_corr_padding(target, source, normalization=normalization, device=device)
    elif function_type == "custom":
        if want_corr:
            shift, corr = phase_cross_corr(target, source, normalization=normalization, device=device)
        else:  # the per-position loop only reads the correlation volume for its verbose statistics
            shift, corr = phase_cross_corr_device(target, source, normalization, device, want_corr=False)
    else:
        raise ValueError(f"unknown function_type {function_type!r}")
    if verbose:
        print(f"Time {t}: shift (dz,dy,dx) = {shift[0]}, {shift[1]}, {shift[2]}")
    dz, dy, dx = shift
    transform = np.eye(4)
    transform[0, 3] = dx
    transform[1, 3] = dy
    transform[2, 3] = dz
    return transform, shift, corr


def remove_beads_fov_from_path_list(position_dirpaths, skip_beads_fov: str):
    """``estimate_stabilization.py:50-74``."""
    if skip_beads_fov != "0":
        print(f"Removing beads FOV {skip_beads_fov} from input data paths")
        position_dirpaths = [p for p in position_dirpaths if skip_beads_fov not in str(p)]
    return position_dirpaths


def _axis_slice(spec, n):
    return slice(0, n) if spec == "all" else slice(spec[0], spec[1])


def estimate_xyz_stabilization_pcc_per_position(input_position_dirpath, output_folder_path, output_shifts_path,
                                                channel_index: int, phase_cross_corr_settings, verbose: bool = False,
                                                device="cuda"):
    """``estimate_stabilization.py:444-587``: one transform per timepoint against the first or the previous timepoint;
    saves ``<row>_<col>_<fov>.npy`` (float32) and, when verbose, the shifts as csv.  Plots are not produced."""
    from pathlib import Path

    from .io import open_ome_zarr

    s = phase_cross_corr_settings
    input_position_dirpath = Path(input_position_dirpath)
    data = open_ome_zarr(input_position_dirpath).data
    T, _, Z, Y, X = data.shape
    # the reference computes a centre crop first and then overrides it with the X/Y slices (:476-496); same here
    x_idx, y_idx, z_idx = _axis_slice(s.X_slice, X), _axis_slice(s.Y_slice, Y), _axis_slice(s.Z_slice, Z)
    if verbose:
        print(f"x_idx: {x_idx}, y_idx: {y_idx}, z_idx: {z_idx}")
    vols = [np.asarray(data[t, channel_index][z_idx, y_idx, x_idx]) for t in range(T)]
    source = vols
    target = [vols[0]] * T if s.t_reference == "first" else [vols[0]] + vols[:-1]
    name = "_".join(input_position_dirpath.parts[-3:])
    transforms, shifts = [], []
    # function_type "custom": get_tform_from_pcc calls phase_cross_corr(source[t], target[t]) — the varying timepoint first, the
    # reference timepoint as the conjugated factor.  The reference image's spectrum is computed once ("first") or falls out of
    # the previous call ("previous": roll) instead of being rebuilt for every timepoint; same shifts (tests/test_io_cli.py).
    prepared = None
    if s.function_type == "custom" and T > 1:
        prepared = PreparedPhaseCrossCorr(vols[0].astype(np.float32), fixed_is_second=True, device=device)
    for t in range(T):
        if t == 0:
            transforms.append(np.eye(4).tolist())
            shifts.append((t, 0, 0, 0))
            continue
        if prepared is not None:
            shift, _ = prepared(vols[t].astype(np.float32), s.normalization, want_corr=False, roll=s.t_reference == "previous")
            if verbose:
                print(f"Time {t}: shift (dz,dy,dx) = {shift[0]}, {shift[1]}, {shift[2]}")
            transform = np.eye(4)
            transform[0, 3], transform[1, 3], transform[2, 3] = shift[2], shift[1], shift[0]
        else:
            transform, shift, _corr = get_tform_from_pcc(t, source, target, function_type=s.function_type,
                                                         normalization=s.normalization, verbose=verbose, device=device,
                                                         want_corr=False)
        transforms.append(transform)
        shifts.append((t, *[float(v) for v in shift]))
    if prepared is not None:
        prepared.close()
    output_folder_path = Path(output_folder_path)
    output_folder_path.mkdir(parents=True, exist_ok=True)
    np.save(output_folder_path / f"{name}.npy", np.array(transforms, dtype=np.float32))
    if verbose:
        Path(output_shifts_path).mkdir(parents=True, exist_ok=True)
        with open(Path(output_shifts_path) / f"{name}.csv", "w") as f:
            f.write("TimepointID,ShiftZ,ShiftY,ShiftX\n")
            for row in shifts:
                f.write(",".join(str(v) for v in row) + "\n")
    return transforms


def estimate_xyz_stabilization_pcc(input_position_dirpaths, output_folder_path, phase_cross_corr_settings,
                                   channel_index: int = 0, sbatch_filepath=None, cluster: str = "local",
                                   verbose: bool = False, device="cuda") -> dict:
    """``estimate_stabilization.py:590-693``: every position (sharded over ranks instead of submitit jobs), then the
    per-position ``.npy`` files are gathered into ``{fov_name: transforms}`` and the temporary folder is removed."""
    import shutil
    from pathlib import Path

    from . import parallel

    input_position_dirpaths = remove_beads_fov_from_path_list(list(input_position_dirpaths),
                                                              phase_cross_corr_settings.skip_beads_fov)
    output_folder_path = Path(output_folder_path)
    transforms_out = output_folder_path / "transforms_per_position"
    shifts_out = output_folder_path / "shifts_per_position"
    transforms_out.mkdir(parents=True, exist_ok=True)
    rank, world = parallel.init()  # binds this rank to GPU LOCAL_RANK before any device call
    for p in parallel.shard_positions(input_position_dirpaths, rank, world):
        estimate_xyz_stabilization_pcc_per_position(p, transforms_out, shifts_out, channel_index,
                                                    phase_cross_corr_settings, verbose=verbose, device=device)
    parallel.barrier()
    fov_transforms = {f.stem: np.load(f).tolist() for f in sorted(transforms_out.glob("*.npy"))}
    parallel.barrier()
    if rank == 0:
        shutil.rmtree(transforms_out)
    return fov_transforms


def save_transforms(model, transforms, output_filepath_settings, output_filepath_plot=None, verbose: bool = False):
    """``registration/utils.py:370-422`` without the plot."""
    from pathlib import Path

    from .utils.config import model_to_yaml

    if transforms is None or len(transforms) == 0:
        raise ValueError("Transforms are empty")
    if not isinstance(transforms, list):
        transforms = transforms.tolist()
    model.affine_transform_zyx_list = transforms
    output_filepath_settings = Path(output_filepath_settings)
    if output_filepath_settings.suffix not in [".yml", ".yaml"]:
        output_filepath_settings = output_filepath_settings.with_suffix(".yml")
    output_filepath_settings.parent.mkdir(parents=True, exist_ok=True)
    model_to_yaml(model, output_filepath_settings)
