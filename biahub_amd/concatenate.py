"""`biahub concatenate`: gather channels (and crops) of several stores into one plate.

Host logic of biahub/concatenate.py — which sources, which channels land where, which ZYX box of each source — around the
one data-parallel step, the crop (`copy_n_paste`, biahub/utils/array_ops.py:9-33 -> ``bh_crop_flip``).  Helper names and
behaviour follow the reference (file:line in each docstring); the job fan-out of the reference (submitit, one job per
position, biahub/concatenate.py:470-540) becomes positions sharded over ranks.
"""

from __future__ import annotations

import glob
import re
from pathlib import Path

import click
import numpy as np

from .array_ops import copy_n_paste
from .io import create_empty_plate, open_ome_zarr, process_single_position
from .settings import ConcatenateSettings
from .utils.config import settings_fingerprint
from .utils.paths import get_output_paths


def _natural_key(s: str):
    """natsort's default ordering for paths: digit runs compare as integers ("A/2/0" before "A/10/0")."""
    return [int(tok) if tok.isdigit() else tok for tok in re.split(r"(\d+)", s)]


def get_path_slice_param(slice_param, path_index, total_paths):
    """The slice specification source `path_index` uses (biahub/concatenate.py:48-75): "all" and a single [start, end]
    apply to every source; a list gives one entry per source, the last one reused when it runs out."""
    if slice_param == "all":
        return "all"
    if isinstance(slice_param, list):
        if len(slice_param) == 2 and all(isinstance(i, int) for i in slice_param):
            return slice_param
        return slice_param[path_index] if path_index < len(slice_param) else slice_param[-1]
    return slice_param


def get_slice(slice_param, max_value: int) -> slice:
    """"all" -> slice(0, max_value); [start, end] -> slice(start, end) (biahub/concatenate.py:203-227)."""
    if slice_param == "all":
        return slice(0, max_value)
    if isinstance(slice_param, list) and len(slice_param) == 2 and all(isinstance(i, int) for i in slice_param):
        return slice(*slice_param)
    raise ValueError(f"Invalid slice parameter: {slice_param}")


def create_path_slicing_params(path_z_slice, path_y_slice, path_x_slice, dataset_shape):
    """[z, y, x] slice objects for a (T,C,Z,Y,X) dataset shape (biahub/concatenate.py:78-95)."""
    return [get_slice(path_z_slice, dataset_shape[2]), get_slice(path_y_slice, dataset_shape[3]),
            get_slice(path_x_slice, dataset_shape[4])]


def calculate_cropped_size(slice_params_zyx) -> tuple[int, int, int]:
    """|stop - start| per axis (biahub/concatenate.py:242-263)."""
    shape = tuple(abs(s.stop - s.start) for s in slice_params_zyx)
    click.echo(f"Output ZYX shape after cropping: {shape}")
    return shape


def validate_slicing_params_zyx(slicing_params_zyx_list) -> None:
    """Every source must crop to the same ZYX size (biahub/concatenate.py:230-239)."""
    first = calculate_cropped_size(slicing_params_zyx_list[0])
    for i, s in enumerate(slicing_params_zyx_list[1:], 1):
        size = calculate_cropped_size(s)
        if size != first:
            raise ValueError(f"Inconsistent slice sizes detected. Path 0 has size {first}, but path {i} has size {size}. "
                             "All paths must have the same slice size.")


def get_channel_combiner_metadata(data_paths_list, processing_channel_names, slicing_params):
    """Expand the source globs and lay the channels out (biahub/concatenate.py:98-200).

    Returns (all_data_paths, all_channel_names, input_channel_idx, output_channel_idx, all_slicing_params), one entry per
    expanded position for the three lists.  A channel name seen before keeps its first output index; the counter then
    continues from that index, exactly as the reference does."""
    z_param, y_param, x_param = slicing_params
    expanded = [[Path(p) for p in sorted(glob.glob(pattern), key=_natural_key) if Path(p).is_dir()] for pattern in data_paths_list]
    if len(expanded) != len(processing_channel_names):
        raise ValueError("zip() argument 2 is " + ("shorter" if len(processing_channel_names) < len(expanded) else "longer") + " than argument 1")
    all_data_paths = [p for paths in expanded for p in paths]
    all_channel_names: list[str] = []
    input_channel_idx, output_channel_idx, all_slicing_params = [], [], []
    counter = 0
    for i, (paths, wanted) in enumerate(zip(expanded, processing_channel_names)):
        sample = open_ome_zarr(paths[0])  # the first position of a source speaks for all of them
        names = sample.channel_names
        zs, ys, xs = (get_path_slice_param(p, i, len(data_paths_list)) for p in (z_param, y_param, x_param))
        for _ in paths:
            all_slicing_params.append(create_path_slicing_params(zs, ys, xs, sample.data.shape))
        if wanted == "all":
            wanted = names
        ins, outs = [], []
        for ch in wanted:
            if ch not in names:
                continue
            if ch not in all_channel_names:
                all_channel_names.append(ch)
                outs.append(counter)
                counter += 1
            else:
                click.echo(f"Warning: Channel {ch} already exists. Skipping and using index from the first entry.")
                counter = all_channel_names.index(ch)
                outs.append(counter)
            ins.append(names.index(ch))
        input_channel_idx.extend([ins for _ in paths])
        output_channel_idx.extend([outs for _ in paths])
    if len(all_slicing_params) > 1:
        validate_slicing_params_zyx(all_slicing_params)
    click.echo(f"Channel names: {all_channel_names}")
    click.echo(f"Input channel indices: {input_channel_idx}")
    click.echo(f"Output channel indices: {output_channel_idx}")
    return all_data_paths, all_channel_names, input_channel_idx, output_channel_idx, all_slicing_params


def _resolve_time_indices(settings: ConcatenateSettings, all_shapes) -> list[int]:
    """biahub/concatenate.py:266-281."""
    if settings.time_indices == "all":
        if len({s[0] for s in all_shapes}) > 1:
            click.echo("Warning: Datasets have different number of time points. Taking the smallest number of time points.")
        return list(range(min(s[0] for s in all_shapes)))
    if isinstance(settings.time_indices, list):
        return settings.time_indices
    return [settings.time_indices]


def prepare_concatenate(settings: ConcatenateSettings, output_dirpath: Path, compressor=None, create: bool = True) -> dict:
    """Resolve the layout, check compatibility and (``create``) lay out the output plate (biahub/concatenate.py:284-396).
    Under several ranks only one of them creates: plate / well metadata has one writer."""
    paths, names, in_idx, out_idx, slicing = get_channel_combiner_metadata(
        settings.concat_data_paths, settings.channel_names, [settings.Z_slice, settings.Y_slice, settings.X_slice])
    outputs = get_output_paths(paths, output_dirpath, ensure_unique_positions=settings.ensure_unique_positions)
    shapes, dtypes, voxels, versions = [], [], [], []
    for p in paths:
        ds = open_ome_zarr(p)
        shapes.append(tuple(ds.data.shape))
        dtypes.append(ds.data.dtype)
        voxels.append(list(ds.scale[-3:]))
        versions.append(ds.version)
    crop_all = settings.Z_slice == "all" and settings.Y_slice == "all" and settings.X_slice == "all"
    same_zyx = all(s[-3:] == shapes[0][-3:] for s in shapes)
    if crop_all and not same_zyx:
        raise ValueError("Datasets have different shapes. All ZYX shapes must match to concatenate when using 'all' for slicing.")
    if any(v != voxels[0] for v in voxels):
        click.echo("Warning: Datasets have different voxel sizes. Taking the first voxel size.")
    T, C, Z, Y, X = shapes[0]
    if all(d == dtypes[0] for d in dtypes):
        dtype = dtypes[0]
    else:
        click.echo("Warning: not all dtypes match. Casting data at float32.")
        dtype = np.dtype(np.float32)
    times = _resolve_time_indices(settings, shapes)
    if not same_zyx:
        click.echo("Warning: Datasets have different shapes, but slicing parameters are specified. Will validate output shapes after cropping.")
    cropped = calculate_cropped_size(slicing[0])
    if cropped[0] > Z or cropped[1] > Y or cropped[2] > X:
        raise ValueError("The cropped shape is larger than the original shape.")
    chunks = [1] + list(settings.chunks_czyx) if settings.chunks_czyx is not None else None
    if create:
        create_empty_plate(output_dirpath, [p.parts[-3:] for p in outputs], names,
                           (len(times), len(names)) + tuple(cropped), chunks=chunks, scale=(1, 1) + tuple(voxels[0]),
                           dtype=dtype, version=settings.output_ome_zarr_version or versions[0], compressor=compressor,
                           shards_ratio=settings.shards_ratio)
        click.echo(f"Created {output_dirpath} ({len(outputs)} positions)")
    return {"all_data_paths": paths, "output_position_paths": outputs, "input_channel_idx_list": in_idx,
            "output_channel_idx_list": out_idx, "all_slicing_params": slicing, "input_time_indices": times,
            "shape": (T, C, Z, Y, X), "dtype": dtype}


def _crop_unit(czyx, zyx_slicing_params, out_dtype=None):
    """One (t, channel-group) unit: NaN -> 0 and the ZYX crop of every channel (the reference hands `copy_n_paste` the
    unit as it comes, biahub/concatenate.py:517-532; its own tests expect the ZYX box, tests/test_concatenate.py:183-240)."""
    out = np.stack([copy_n_paste(zyx, zyx_slicing_params) for zyx in czyx])
    return out if out_dtype is None else out.astype(out_dtype, copy=False)


def concatenate(settings: ConcatenateSettings, output_dirpath: Path, init_only: bool = False, resume: bool = False,
                compressor=None, rank: int = 0, world: int = 1, create: bool = True) -> dict:
    """Create the output plate (``create``), then crop-copy the source positions into it.  Work is sharded by *output*
    position (`rank::world` over the distinct destinations, in first-appearance order): sources that are channel-combined
    into one position share its resume record and zattrs, so one position has one writer."""
    prep = prepare_concatenate(settings, Path(output_dirpath), compressor, create=create)
    if init_only:
        return prep
    units = list(zip(prep["all_data_paths"], prep["output_position_paths"], prep["input_channel_idx_list"],
                     prep["output_channel_idx_list"], prep["all_slicing_params"]))
    by_dst: dict = {}
    for u in units:
        by_dst.setdefault(str(u[1]), []).append(u)
    times = prep["input_time_indices"]
    for key in list(by_dst)[rank::world]:
        for src, dst, cin, cout, box in by_dst[key]:
            existing = open_ome_zarr(dst).zattrs.get("extra_metadata") or {}
            process_single_position(
                _crop_unit, input_position_path=src, output_position_path=dst,
                input_channel_indices=[[c] for c in cin], output_channel_indices=[[c] for c in cout],
                input_time_indices=times, output_time_indices=list(range(len(times))), resume=resume,
                resume_token=settings_fingerprint(settings),
                extra_metadata={**existing, "biahub-concatenate": settings.model_dump()},
                zyx_slicing_params=box, out_dtype=prep["dtype"])
    return prep
