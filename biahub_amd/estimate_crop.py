"""`biahub estimate-crop`: the ZYX box in which two co-registered datasets (label-free and light-sheet) both carry data,
written back as the slicing of a concatenate configuration (biahub/estimate_crop.py).

Per position the validity masks (voxel neither 0 nor NaN) of the first channel of every time point of both datasets are
built on the GPU, one bit per voxel (``bh_valid_mask``); the volumes whose valid count is within 20 % of the median are
ANDed (``bh_bits_and``); the largest interior cuboid of the result comes from `register.find_lir`, which needs seven 2-D
slices of the mask only.
"""

from __future__ import annotations

import shutil
from ast import literal_eval
from pathlib import Path

import click
import numpy as np
import torch

from . import _lib
from .device import _NP_TO_DT, get_context, ptr, resolve_device, upload
from .io import open_ome_zarr
from .register import find_lir
from .settings import ConcatenateSettings
from .utils.config import model_to_yaml, yaml_to_model


def valid_mask_device(vol, device="cuda"):
    """(bit mask int32 tensor of ceil(n/64)*2 words, number of valid voxels) of one volume: valid = not 0 and not NaN."""
    dev = resolve_device(device)
    if isinstance(vol, torch.Tensor):
        t = vol.to(dev).contiguous()
        code = {torch.uint8: _lib.DT_U8, torch.uint16: _lib.DT_U16, torch.int16: _lib.DT_I16, torch.float32: _lib.DT_F32}.get(t.dtype)
        if code is None:
            t, code = ((t != 0) & ~torch.isnan(t) if t.is_floating_point() else t != 0).to(torch.uint8), _lib.DT_U8
    else:
        a = np.ascontiguousarray(vol)
        if a.dtype not in _NP_TO_DT:  # e.g. float64: a cast could flush tiny values to zero, so the predicate runs on the host
            a = ((a != 0) & ~np.isnan(a)).astype(np.uint8) if a.dtype.kind == "f" else (a != 0).astype(np.uint8)
        t, code = upload(a, dev)
    n = t.numel()
    bits = torch.empty(((n + 63) // 64) * 2, dtype=torch.int32, device=dev)
    import ctypes

    count = ctypes.c_uint64()
    ctx = get_context(dev)
    with torch.cuda.device(dev):
        _lib.check(ctx.lib.bh_valid_mask(ctx.handle, ptr(t), code, n, ptr(bits), ctypes.byref(count)))
    return bits, int(count.value)


def _unpack(bits, shape, dev):
    n = int(np.prod(shape))
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx = get_context(dev)
    with torch.cuda.device(dev):
        _lib.check(ctx.lib.bh_bits_unpack(ctx.handle, ptr(bits), n, ptr(out)))
    return out.reshape(shape)


def estimate_crop_one_position(lf_dir, ls_dir, lf_mask_radius: float | None = None, output_dir: Path | None = None,
                               device="cuda"):
    """Crop region in which both datasets are non-zero (biahub/estimate_crop.py:28-143).  Returns ([z0, z1], [y0, y1],
    [x0, x1]).  Where the reference leaves `_max_zyx_dims` undefined (equal shapes, :72-79) the common shape is used."""
    lf_dir, ls_dir = Path(lf_dir), Path(ls_dir)
    fov = "/".join(lf_dir.parts[-3:])
    click.echo(f"Processing FOV: {fov}")
    dev = resolve_device(device)
    lf, ls = open_ome_zarr(lf_dir).data, open_ome_zarr(ls_dir).data
    lf_shape, ls_shape = tuple(lf.shape[-3:]), tuple(ls.shape[-3:])
    dims = tuple(int(v) for v in np.asarray([lf_shape, ls_shape]).min(axis=0))
    if lf_shape != ls_shape:
        click.echo("WARNING: Phase and fluorescence datasets should have the same shape, got"
                   f" phase shape: {lf_shape}, fluorescence shape: {ls_shape}")
    if lf.shape[0] != ls.shape[0]:
        raise ValueError("all the input array dimensions except for the concatenation axis must match exactly")
    T = lf.shape[0]
    masks, counts = {}, np.zeros((T, 2), dtype=np.int64)
    for c, arr in enumerate((lf, ls)):
        for t in range(T):
            vol = arr.read_volume(t, 0)  # first channel only (:58, :61)
            if vol.shape != dims:
                vol = np.ascontiguousarray(vol[: dims[0], : dims[1], : dims[2]])
            masks[t, c], counts[t, c] = valid_mask_device(vol, dev)
    median = np.median(counts)
    valid = [(t, c) for t in range(T) for c in range(2) if 0.8 * median < counts[t, c] < 1.2 * median]
    if not valid:
        click.echo("No valid data found for current position, will not crop.")
        return tuple(zip((0, 0, 0), dims))
    acc = masks[valid[0]].clone()
    ctx = get_context(dev)
    with torch.cuda.device(dev):
        for key in valid[1:]:
            _lib.check(ctx.lib.bh_bits_and(ctx.handle, ptr(acc), ptr(masks[key]), acc.numel()))
    combined = _unpack(acc, dims, dev).cpu().numpy().astype(bool)
    if lf_mask_radius is not None:
        click.echo(f"Applying circular mask of radius {lf_mask_radius} to phase channel.")
        if not (0 < lf_mask_radius <= 1):
            raise ValueError("lf_mask_radius must be a fraction of image width (0 < lf_mask_radius <= 1).")
        H, W = dims[1:]  # the reference builds the circle on the mask it has already cropped to the common shape (:106)
        circle = np.zeros((H, W), dtype=bool)
        y, x = np.ogrid[:H, :W]
        center = (H // 2, W // 2)
        radius = int(lf_mask_radius * min(center))
        # the reference pairs x with the row centre and y with the column centre (:111): kept, it only matters off-square
        circle[(x - center[0]) ** 2 + (y - center[1]) ** 2 <= radius**2] = True
        combined = combined * circle[: dims[1], : dims[2]]
    z_slice, y_slice, x_slice = find_lir(combined)
    click.echo(f"Estimated crop for FOV {fov}:\nZ: {z_slice.start} - {z_slice.stop}\nY: {y_slice.start} - {y_slice.stop}\n"
               f"X: {x_slice.start} - {x_slice.stop}")
    result = ([z_slice.start, z_slice.stop], [y_slice.start, y_slice.stop], [x_slice.start, x_slice.stop])
    if output_dir:
        import pandas as pd

        pd.DataFrame([{"fov": fov, "Z": result[0], "Y": result[1], "X": result[2]}]).to_csv(
            Path(output_dir) / f"{fov.replace('/', '_')}.csv", index=False)
    return result


def standardize_ranges(all_ranges) -> np.ndarray:
    """The crop common to all positions (biahub/estimate_crop.py:258-266): largest start, smallest stop per axis.
    ``all_ranges``: (positions, 3 axes, 2); returns [[z0, y0, x0], [z1, y1, x1]]."""
    r = np.asarray(all_ranges)
    return np.concatenate([r[..., 0].max(axis=0, keepdims=True), r[..., 1].min(axis=0, keepdims=True)])


def estimate_crop(config_filepath, output_filepath, lf_mask_radius: float = 0.95, sbatch_filepath=None, local: bool = False,
                  device="cuda") -> None:
    """biahub/estimate_crop.py:146-282: per-position crops of the two datasets a ConcatenateSettings file names (label-free
    first, light-sheet second), merged into one box, written back as that file's Z/Y/X_slice."""
    import pandas as pd

    config_filepath = Path(config_filepath)
    if config_filepath.suffix not in (".yml", ".yaml"):
        raise ValueError("Config file must be a yaml file")
    settings = yaml_to_model(config_filepath, ConcatenateSettings)
    output_dir = Path(output_filepath).parent
    csv_dir = output_dir / "crop_estimates"
    csv_dir.mkdir(exist_ok=True, parents=True)
    lf_dirs = [p for p in config_filepath.parent.glob(settings.concat_data_paths[0]) if p.is_dir()]
    click.echo(f"Found {len(lf_dirs)} phase channels.")
    ls_dirs = [p for p in config_filepath.parent.glob(settings.concat_data_paths[1]) if p.is_dir()]
    click.echo(f"Found {len(ls_dirs)} fluorescence channels.")
    if len(lf_dirs) != len(ls_dirs):
        raise ValueError("Number of phase and fluorescence channels must be the same.")
    for ls_dir, lf_dir in zip(ls_dirs, lf_dirs):
        estimate_crop_one_position(lf_dir, ls_dir, lf_mask_radius=lf_mask_radius, output_dir=csv_dir, device=device)
    csvs = list(csv_dir.glob("*.csv"))
    if not csvs:
        click.echo("No crop CSV files found. Exiting.")
        return
    df = pd.concat([pd.read_csv(f, dtype={"fov": str}) for f in csvs], ignore_index=True)
    df = df.drop_duplicates(subset=["fov", "Z", "Y", "X"]).sort_values("fov")
    for col in ("X", "Y", "Z"):
        df[col] = df[col].apply(literal_eval)
    df.to_csv(output_dir / "crop_slices.csv", index=False)
    ranges = standardize_ranges([[row["Z"], row["Y"], row["X"]] for _, row in df.iterrows()])
    click.echo(f"Standardized ranges:\nZ: {ranges[:, 0].tolist()}\nY: {ranges[:, 1].tolist()}\nX: {ranges[:, 2].tolist()}")
    out = settings.model_copy()
    out.Z_slice, out.Y_slice, out.X_slice = ranges[:, 0].tolist(), ranges[:, 1].tolist(), ranges[:, 2].tolist()
    model_to_yaml(out, output_filepath)
    shutil.rmtree(csv_dir)
    click.echo("Done.")
