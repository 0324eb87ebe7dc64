"""Command-line surface of the hot-path steps: ``python -m biahub_amd <step> ...``.

Command names, options and the side contracts (``RESOURCES:{json}`` line, ``--init``, ``--cluster debug`` /
``--local``, output paths mirroring ``row/col/fov``, ``slurm_output/submitit_jobs_ids.log``) follow the reference
CLIs (biahub/deskew.py:772-780, deconvolve.py:69-83, register.py:401-408, stabilize.py:327-333, flip.py:8-11,
cli/parsing.py).  Execution is in-process on the visible GPU(s): under ``torchrun`` the positions are sharded over
ranks (biahub_amd/parallel.py); ``--cluster slurm`` is refused — this package does not submit jobs.
"""

from __future__ import annotations

import sys
from pathlib import Path

import click
import numpy as np

from . import parallel
from .io import create_empty_plate, open_ome_zarr, process_single_position
from .settings import (DeconvolveSettings, DeskewSettings, EstimateRegistrationSettings,
                       EstimateStabilizationSettings, FlatFieldCorrectionSettings, PhaseCrossCorrSettings,
                       ProcessingImportFuncSettings, PsfFromBeadsSettings, RegistrationSettings, RichardsonLucySettings,
                       StabilizationSettings)
from .utils.cluster import echo_resources, estimate_resources, get_submitit_cluster
from .utils.config import model_to_yaml, settings_fingerprint, yaml_to_model
from .utils.paths import get_output_paths, sbatch_to_submitit


def _output_compressor():
    """Chunk compressor of the stores this CLI creates.  Default: what iohub gives the reference's outputs — Blosc with
    zstd level 1 and bit shuffle (NGFF 0.4: numcodecs ``Blosc``; 0.5: the zarr v3 ``blosc`` codec).  BH_ZARR_COMPRESSOR =
    none | blosc | blosc-lz4 | zstd | zlib overrides it.  ``blosc-lz4`` keeps the container and the bit shuffle and swaps the
    inner codec for lz4 — any Blosc reader (numcodecs, iohub) takes it — which lets the GPU run the block codec too
    (csrc/lz4.hip): a device-resident result then crosses PCIe compressed and the host only writes files."""
    import os

    kind = os.environ.get("BH_ZARR_COMPRESSOR", "blosc").lower()
    if kind in ("none", "raw", ""):
        return None
    if kind == "blosc":
        return "blosc"
    if kind in ("blosc-lz4", "blosc_lz4", "lz4"):
        return {"id": "blosc", "cname": "lz4", "clevel": 1, "shuffle": 2, "blocksize": 0}
    if kind in ("zstd", "zlib", "gzip"):
        return {"id": kind, "level": 1}
    raise click.UsageError(f"BH_ZARR_COMPRESSOR={kind!r}: expected none, blosc, blosc-lz4, zstd, zlib or gzip")

_MULTI = {"-i", "--input-position-dirpaths", "-s", "--source-position-dirpaths", "-t", "--target-position-dirpaths"}


def expand_eat_all(argv: list[str]) -> list[str]:
    """``-i a b c`` -> ``-i a -i b -i c`` (the reference lets ``-i`` swallow a shell glob)."""
    out, i = [], 0
    while i < len(argv):
        tok = argv[i]
        out.append(tok)
        i += 1
        if tok in _MULTI:
            first = True
            while i < len(argv) and not argv[i].startswith("-"):
                if not first:
                    out.append(tok)
                out.append(argv[i])
                first = False
                i += 1
    return out


def _positions(ctx, param, value):
    paths = [Path(v) for v in value]
    for p in paths:
        if not ((p / ".zgroup").exists() or (p / "zarr.json").exists()):
            raise click.BadParameter(f"{p} is not an OME-Zarr position")
    return sorted(paths)


def _resolve_cluster(cluster: str | None, local: bool = False) -> str:
    resolved = get_submitit_cluster(local=local, cluster=cluster)
    if resolved == "slurm":
        raise click.UsageError("biahub_amd runs in-process on local GPUs: use --cluster debug/local (or --local)")
    return resolved


def _deskew_device(configured: str) -> str:
    """The GPU whenever one is visible (this package is the MI355X path); on a box without one a config that says ``cpu`` —
    the reference's default, what BASELINE config 1 runs with — takes libbhcore's host implementation of the operator."""
    import torch

    if torch.cuda.is_available():
        return "cuda"
    return "cpu" if str(configured).lower() == "cpu" else "cuda"  # "cuda" without a GPU fails loudly in resolve_device


def _create_plate_once(*args, **kwargs):
    """Rank 0 lays the output plate out, the other ranks wait for it: ``create_empty_plate`` rewrites plate / row / well
    metadata, which concurrent ranks would race on (the reference creates the plate once, in the submitting process,
    before any job starts: biahub/deskew.py:668-676)."""
    parallel.init()
    try:
        parallel.rank0_first(create_empty_plate, *args, **kwargs)  # every rank reaches the collective, also when rank 0 fails
    except parallel.Rank0Error as e:
        raise click.ClickException(str(e)) from e


def _run_positions(step: str, inputs, outputs, make_job, out_parent: Path):
    """Shard positions over ranks (rank r on GPU LOCAL_RANK, bound by ``parallel.init``), run them in the foreground,
    write the job-id log the reference writes.  Any rank that saw a position fail exits non-zero."""
    rank, world = parallel.init()
    pairs = list(zip(inputs, outputs))
    log_dir = out_parent / "slurm_output"
    if rank == 0:
        log_dir.mkdir(exist_ok=True)
        (log_dir / "submitit_jobs_ids.log").write_text("\n".join(f"{step}-{i}" for i in range(len(pairs))))
    if world > 1:
        click.echo(f"[rank {rank}/{world}] {step}: device {parallel.bound_device()}, "
                   f"{len(parallel.shard_positions(pairs, rank, world))} of {len(pairs)} position(s)", err=True)

    def one(pair):
        src, dst = pair
        make_job(src, dst)
        click.echo(f"{step.capitalize()} complete: {src}")
        with open_ome_zarr(dst) as d:
            return float(np.prod(d.data.shape))

    st = parallel.process_positions(pairs, one, rank, world, label=lambda pair: str(pair[0]))
    parallel.barrier()
    rows = parallel.gather_stats(st)
    failed = sum(r.n_failed for r in rows)
    if st.n_failed:
        raise click.ClickException(f"{st.n_failed} position(s) failed on rank {rank} (first: {st.first_error})")
    if rank == 0 and failed:
        raise click.ClickException(f"{failed} position(s) failed on other ranks")


@click.group()
def cli():
    """MI355X-native per-position reconstruction steps (biahub-compatible)."""


def _common(f):
    f = click.option("--input-position-dirpaths", "-i", multiple=True, required=True, callback=_positions,
                     help='Paths to input positions, e.g. "input.zarr/*/*/*"')(f)
    f = click.option("--output-dirpath", "-o", required=True, type=click.Path(path_type=Path), help="Path to output.zarr")(f)
    f = click.option("--sbatch-filepath", "-sb", default=None, type=click.Path(exists=True),
                     help="Accepted for compatibility; Slurm parameters are parsed and ignored.")(f)
    f = click.option("--monitor", "-m", is_flag=True, default=False, help="Accepted for compatibility.")(f)
    return f


def _config(f):
    return click.option("--config-filepath", "-c", required=True, type=click.Path(exists=True, path_type=Path),
                        help="Path to YAML configuration file")(f)


@cli.command("deskew")
@_common
@_config
@click.option("--cluster", type=click.Choice(["slurm", "local", "debug"], case_sensitive=False), default="debug",
              show_default=True)
@click.option("--init", "init_only", is_flag=True, default=False, help="Only initialize the output store and exit.")
@click.option("--resume/--no-resume", "resume", default=False, show_default=True)
def deskew_cli(input_position_dirpaths, output_dirpath, sbatch_filepath, monitor, config_filepath, cluster, init_only,
               resume):
    """Deskew oblique light-sheet positions (reference: ``biahub deskew``)."""
    from .deskew import _fast_deskew_czyx, _fast_deskew_czyx_device, get_deskewed_data_shape

    settings = yaml_to_model(config_filepath, DeskewSettings)
    with open_ome_zarr(input_position_dirpaths[0]) as ds:
        channel_names, (T, C, Z, Y, X) = ds.channel_names, ds.data.shape
        if not np.isclose(settings.pixel_size_um, ds.scale[-1], rtol=0.05):
            click.echo(f"Warning: config pixel_size_um={settings.pixel_size_um} differs from the input zarr XY scale "
                       f"({ds.scale[-1]:.4f}).", err=True)
        version = settings.output_ome_zarr_version or ds.version
    out_shape, voxel = get_deskewed_data_shape((Z, Y, X), settings.ls_angle_deg, settings.px_to_scan_ratio,
                                               settings.keep_overhang, settings.average_n_slices, settings.pixel_size_um)
    _create_plate_once(output_dirpath, [p.parts[-3:] for p in input_position_dirpaths], channel_names,
                       (T, C) + tuple(out_shape), scale=(1, 1) + tuple(voxel), version=version, compressor=_output_compressor())
    minutes, cpus, gb = estimate_resources((T, C, Z, Y, X), ram_multiplier=8, time_multiplier=0.5, max_num_cpus=16)
    echo_resources(cpus, cpus * gb, minutes)
    if init_only:
        click.echo(f"Initialized {output_dirpath} ({len(input_position_dirpaths)} positions)")
        return
    if sbatch_filepath:
        sbatch_to_submitit(sbatch_filepath)
    _resolve_cluster(cluster)
    kw = dict(ls_angle_deg=settings.ls_angle_deg, px_to_scan_ratio=settings.px_to_scan_ratio,
              keep_overhang=settings.keep_overhang, average_n_slices=settings.average_n_slices,
              overhang_fill=settings.overhang_fill, device=_deskew_device(settings.device),
              extra_metadata={"biahub-deskew": settings.model_dump()})
    outs = get_output_paths(input_position_dirpaths, output_dirpath)
    # on a GPU the result stays in HBM until the store has permuted (and, for lz4 stores, compressed) it
    op = _fast_deskew_czyx if str(kw["device"]).startswith("cpu") else _fast_deskew_czyx_device
    _run_positions("deskew", input_position_dirpaths, outs,
                   lambda s, d: process_single_position(op, s, d, resume=resume,
                                                        resume_token=settings_fingerprint(settings), **kw),
                   Path(output_dirpath).parent)


@cli.command("process-with-config")
@_common
@_config
@click.option("--local", "-l", is_flag=True, default=False)
def process_with_config_cli(input_position_dirpaths, output_dirpath, sbatch_filepath, monitor, config_filepath, local):
    """Process data with the functions named in the config (reference: ``biahub process-with-config``,
    process_data.py:148-349); every channel of a timepoint reaches the functions at once."""
    from .process_data import process_czyx, resolve_function

    settings = yaml_to_model(config_filepath, ProcessingImportFuncSettings)
    if not settings.processing_functions:
        raise ValueError("Processing functions must be specified")
    with open_ome_zarr(input_position_dirpaths[0]) as ds:
        names, (T, Cn, Z, Y, X), scale, version = ds.channel_names, ds.data.shape, list(ds.scale), ds.version
    for proc in settings.processing_functions:
        if proc.input_channels is not None and len(proc.input_channels) == 1:
            proc.input_channels[0] = names.index(proc.input_channels[0])
        else:
            raise ValueError("Channel must be specified for preprocessing functions")
        try:
            resolve_function(proc.function)
        except ValueError as e:
            raise ValueError(f"Function {proc.function} could not be resolved: {e}") from e
    out_shape, new_scale = (T, Cn, Z, Y, X), scale
    for proc in settings.processing_functions:  # the reference sizes the output from the first binning function (:211-236)
        if proc.function.endswith("process_data.binning_czyx"):
            f = proc.kwargs.get("binning_factor_zyx", (1, 4, 4))
            click.echo(f"Binning factor: {f}")
            out_shape = (T, Cn, Z // f[0], Y // f[1], X // f[2])
            new_scale = scale[:2] + [scale[2] * f[0], scale[3] * f[1], scale[4] * f[2]]
            break
    _create_plate_once(output_dirpath, [p.parts[-3:] for p in input_position_dirpaths], names, out_shape, scale=new_scale,
                       dtype=np.float32, version=settings.output_ome_zarr_version or version, compressor=_output_compressor())
    if sbatch_filepath:
        sbatch_to_submitit(sbatch_filepath)
    allc = [list(range(Cn))]
    outs = get_output_paths(input_position_dirpaths, output_dirpath)
    _run_positions("process-with-config", input_position_dirpaths, outs,
                   lambda s, d: process_single_position(process_czyx, s, d, input_time_indices=list(range(T)),
                                                        input_channel_indices=allc, output_channel_indices=allc,
                                                        processing_functions=settings.processing_functions),
                   Path(output_dirpath).parent)


@cli.command("flat-field")
@_common
@_config
@click.option("--cluster", type=click.Choice(["slurm", "local", "debug"], case_sensitive=False), default="debug",
              show_default=True)
@click.option("--init", "init_only", is_flag=True, default=False, help="Only initialize the output store and exit.")
@click.option("--resume/--no-resume", "resume", default=False, show_default=True)
def flat_field_cli(input_position_dirpaths, output_dirpath, sbatch_filepath, monitor, config_filepath, cluster, init_only,
                   resume):
    """Flat-field correct the selected channels of every (t) volume (reference: ``biahub flat-field``,
    flat_field.py:213-392); the other channels are copied as float32."""
    from .flat_field import _flat_field_czyx

    settings = yaml_to_model(config_filepath, FlatFieldCorrectionSettings)
    with open_ome_zarr(input_position_dirpaths[0]) as ds:
        names, shape = ds.channel_names, ds.data.shape
    _same_shape_plate(input_position_dirpaths, output_dirpath, settings.output_ome_zarr_version)
    minutes, cpus, gb = estimate_resources(shape, ram_multiplier=8, time_multiplier=0.7, max_num_cpus=16)
    echo_resources(cpus, cpus * gb, minutes)
    if init_only:
        click.echo(f"Initialized {output_dirpath} ({len(input_position_dirpaths)} positions)")
        return
    if settings.channel_names is None:
        target = list(names)
        click.echo(f"Flat fielding ALL channels: {names}")
    elif settings.channel_names:
        for n in settings.channel_names:
            if n not in names:
                raise click.ClickException(f"Channel '{n}' not found in input dataset. Available channels: {names}")
        target = list(settings.channel_names)
        click.echo(f"Input channels: {names}")
        click.echo(f"Flat field channels: {target}")
        click.echo("Other channels will be copied as-is")
    else:
        raise click.ClickException("Must specify either 'channel_names' or set channel_names to null in config.")
    if sbatch_filepath:
        sbatch_to_submitit(sbatch_filepath)
    _resolve_cluster(cluster)
    allc = [list(range(shape[1]))]  # _flat_field_czyx sees every channel of a timepoint at once (:144-155)
    outs = get_output_paths(input_position_dirpaths, output_dirpath)
    _run_positions("flat-field", input_position_dirpaths, outs,
                   lambda s, d: process_single_position(_flat_field_czyx, s, d, input_channel_indices=allc,
                                                        output_channel_indices=allc, resume=resume,
                                                        resume_token=settings_fingerprint(settings),
                                                        target_indices=[names.index(n) for n in target],
                                                        extra_metadata={"biahub-flat_field": settings.model_dump()}),
                   Path(output_dirpath).parent)


def _same_shape_plate(inputs, output_dirpath, version_override, dtype=np.float32):
    with open_ome_zarr(inputs[0]) as ds:
        names, shape, scale, version = ds.channel_names, ds.data.shape, ds.scale, version_override or ds.version
    _create_plate_once(output_dirpath, [p.parts[-3:] for p in inputs], names, shape, scale=scale, version=version,
                       dtype=dtype, compressor=_output_compressor())
    return names, shape, scale


@cli.command("deconvolve")
@_common
@_config
@click.option("--psf-dirpath", "-p", required=True, type=click.Path(exists=True, file_okay=False, path_type=Path),
              help="Path to psf.zarr")
@click.option("--local", "-l", is_flag=True, default=False)
def deconvolve_cli(input_position_dirpaths, output_dirpath, sbatch_filepath, monitor, config_filepath, psf_dirpath,
                   local):
    """Tikhonov deconvolution across T and C with a PSF (reference: ``biahub deconvolve``)."""
    from .deconvolve import compute_tranfser_function, deconvolve

    settings = yaml_to_model(config_filepath, DeconvolveSettings)
    _, shape, scale = _same_shape_plate(input_position_dirpaths, output_dirpath, settings.output_ome_zarr_version)
    _resolve_cluster(None, local=True)  # the reference's --local flag; there is no Slurm path here
    with open_ome_zarr(Path(psf_dirpath, "0/0/0")) as psf_ds:
        if scale[-3:] != psf_ds.scale[-3:]:
            click.echo(f"Warning: PSF scale {psf_ds.scale[-3:]} does not match data scale {scale[-3:]}.", err=True)
        psf = psf_ds.data[0, 0]
    click.echo("Computing transfer function...")
    tf = compute_tranfser_function(psf, tuple(shape[-3:]))
    tf_store = Path(output_dirpath).parent / "transfer_function.zarr"
    from .io import create_empty_position

    if parallel.world_info()[0] == 0:  # one writer for the shared store (every rank keeps its own device copy of tf)
        create_empty_position(tf_store, ["PSF"], (1, 1) + tuple(shape[-3:]),
                              chunks=(1, 1, min(256, shape[-3])) + tuple(shape[-2:]), scale=scale)
        open_ome_zarr(tf_store).data[0, 0] = tf
    outs = get_output_paths(input_position_dirpaths, output_dirpath)
    _run_positions("deconvolve", input_position_dirpaths, outs,
                   lambda s, d: process_single_position(deconvolve, s, d, transfer_function=tf,
                                                        regularization_strength=float(settings.regularization_strength)),
                   Path(output_dirpath).parent)


def _apply_inverse_transfer_function(input_position_dirpaths, transfer_function_dirpath, config_filepath, output_dirpath,
                                     sbatch_filepath=None, cluster="debug", init_only=False, filter_storage="f32"):
    """Orchestrator of biahub/apply_inverse_transfer_function.py:74-196: validate the config, lay the output plate out,
    print the resource line, then one in-process job per position (sharded over ranks)."""
    from .apply_inverse_transfer_function import (apply_inverse_transfer_function_single_position, estimate_resources as wo_estimate,
                                                  get_reconstruction_output_metadata)
    from .compute_transfer_function import _refuse_unsupported
    from .settings import ReconstructionSettings

    settings = yaml_to_model(config_filepath, ReconstructionSettings)
    with open_ome_zarr(input_position_dirpaths[0]) as ds:
        input_shape = ds.data.shape
    meta = get_reconstruction_output_metadata(input_position_dirpaths[0], config_filepath)
    from .utils.paths import PROVENANCE_METADATA_KEYS

    # per-position provenance (biahub-*, waveorder, cytoland attributes) travels from the input plate into the reconstruction
    # (biahub/apply_inverse_transfer_function.py:66-72)
    _create_plate_once(output_dirpath, [p.parts[-3:] for p in input_position_dirpaths], meta["channel_names"], meta["shape"],
                       scale=meta["scale"], dtype=np.float32, version=meta["version"], compressor=_output_compressor(),
                       metadata_sources=Path(input_position_dirpaths[0]).parents[2], metadata_keys=PROVENANCE_METADATA_KEYS)
    num_cpus, mem_per_cpu = wo_estimate(list(input_shape), settings, 16)
    T, C, Z, Y, X = input_shape
    minutes, _, _ = estimate_resources((T, len(settings.input_channel_names), Z, Y, X), time_multiplier=3.0, max_num_cpus=16)
    echo_resources(num_cpus, num_cpus * mem_per_cpu, minutes)
    if init_only:
        click.echo(f"Created {output_dirpath} ({len(input_position_dirpaths)} positions, "
                   f"{len(settings.output_channel_names)} output channels)")
        return
    try:
        _refuse_unsupported(settings)
    except (NotImplementedError, ValueError) as e:
        raise click.UsageError(str(e)) from e
    if sbatch_filepath:
        sbatch_to_submitit(sbatch_filepath)
    _resolve_cluster(cluster)
    outs = get_output_paths(input_position_dirpaths, output_dirpath)
    _run_positions("apply-inv-tf", input_position_dirpaths, outs,
                   lambda s, d: apply_inverse_transfer_function_single_position(
                       s, transfer_function_dirpath, config_filepath, d, num_cpus, meta["channel_names"],
                       filter_storage=filter_storage),
                   Path(output_dirpath).parent)


@cli.command("apply-inv-tf")
@_common
@click.option("--transfer-function-dirpath", "-t", default=None, type=click.Path(path_type=Path),
              help="Path to transfer function zarr (not required for --init).")
@_config
@click.option("--cluster", type=click.Choice(["slurm", "local", "debug"], case_sensitive=False), default="debug",
              show_default=True)
@click.option("--init", "init_only", is_flag=True, default=False, help="Only initialize the output store and exit.")
@click.option("--filter-storage", type=click.Choice(["f32", "bf16"]), default="f32", show_default=True,
              help="Storage of the staged inverse filter on the GPU (bf16: half the filter bytes, products in float32).")
def apply_inv_tf_cli(input_position_dirpaths, output_dirpath, sbatch_filepath, monitor, transfer_function_dirpath,
                     config_filepath, cluster, init_only, filter_storage):
    """Apply an inverse transfer function to a dataset using a configuration file (reference: ``biahub apply-inv-tf``,
    apply_inverse_transfer_function.py:199-262)."""
    if not init_only and transfer_function_dirpath is None:
        raise click.UsageError("--transfer-function-dirpath / -t is required unless using --init.")
    _apply_inverse_transfer_function(input_position_dirpaths, transfer_function_dirpath, config_filepath, output_dirpath,
                                     sbatch_filepath, cluster, init_only, filter_storage)


@cli.command("compute-tf")
@click.option("--input-position-dirpaths", "-i", multiple=True, required=True, callback=_positions)
@_config
@click.option("--output-dirpath", "-o", required=True, type=click.Path(path_type=Path), help="Path to output.zarr")
def compute_tf_cli(input_position_dirpaths, config_filepath, output_dirpath):
    """Compute a transfer function using a dataset and configuration file; the shape of the first position counts
    (reference: ``biahub compute-tf``, compute_transfer_function.py:16-38)."""
    from .compute_transfer_function import compute_transfer_function_cli

    parallel.init()
    try:  # one writer; the store is shared by every rank of a later apply-inv-tf
        parallel.rank0_first(compute_transfer_function_cli, input_position_dirpaths[0], config_filepath, output_dirpath)
    except (NotImplementedError, ValueError) as e:
        raise click.UsageError(str(e)) from e
    except parallel.Rank0Error as e:
        raise click.ClickException(str(e)) from e
    click.echo(f"Transfer function computed and saved to {output_dirpath}.")


@cli.command("reconstruct")
@_common
@_config
@click.option("--cluster", type=click.Choice(["slurm", "local", "debug"], case_sensitive=False), default="debug",
              show_default=True)
def reconstruct_cli(input_position_dirpaths, output_dirpath, sbatch_filepath, monitor, config_filepath, cluster):
    """``compute-tf`` followed by ``apply-inv-tf`` (reference: ``biahub reconstruct``, reconstruct.py:27-78): the transfer
    function goes to ``<output parent>/transfer_function_<config stem>.zarr``."""
    from .compute_transfer_function import compute_transfer_function_cli

    tf_path = Path(output_dirpath).parent / ("transfer_function_" + Path(config_filepath).stem + ".zarr")
    parallel.init()
    try:
        parallel.rank0_first(compute_transfer_function_cli, input_position_dirpaths[0], config_filepath, tf_path)
    except (NotImplementedError, ValueError) as e:
        raise click.UsageError(str(e)) from e
    except parallel.Rank0Error as e:
        raise click.ClickException(str(e)) from e
    _apply_inverse_transfer_function(input_position_dirpaths, tf_path, config_filepath, output_dirpath, sbatch_filepath,
                                     cluster)


def _rl_operator(host_op, device_op):
    """The device-resident adapter on a GPU box (volumes arrive through the store's device read path and leave through its
    device codec), the numpy one without a GPU or with BH_PIPE_DEVICE_INPUT=0."""
    import os

    try:
        import torch

        if torch.cuda.is_available() and os.environ.get("BH_PIPE_DEVICE_INPUT", "1") != "0":
            return device_op
    except ImportError:  # pragma: no cover
        pass
    return host_op


@cli.command("rl-deconvolve")
@_common
@_config
@click.option("--psf-dirpath", "-p", required=True, type=click.Path(exists=True, file_okay=False, path_type=Path))
def rl_deconvolve_cli(input_position_dirpaths, output_dirpath, sbatch_filepath, monitor, config_filepath, psf_dirpath):
    """Richardson-Lucy deconvolution (north-star extension; same layout as ``deconvolve``)."""
    from .deconvolve import richardson_lucy_czyx, richardson_lucy_czyx_device

    settings = yaml_to_model(config_filepath, RichardsonLucySettings)
    _same_shape_plate(input_position_dirpaths, output_dirpath, settings.output_ome_zarr_version)
    psf = open_ome_zarr(Path(psf_dirpath, "0/0/0")).data[0, 0]
    outs = get_output_paths(input_position_dirpaths, output_dirpath)
    _run_positions("rl-deconvolve", input_position_dirpaths, outs,
                   lambda s, d: process_single_position(_rl_operator(richardson_lucy_czyx, richardson_lucy_czyx_device), s, d,
                                                        psf_zyx=psf, iterations=settings.iterations, eps=settings.eps),
                   Path(output_dirpath).parent)


@cli.command("stabilize")
@_common
@click.option("--config-filepaths", "--config-filepath", "-c", "config_filepaths", multiple=True, required=True,
              type=click.Path(exists=True, path_type=Path),
              help="YAML configuration file(s): one for all positions, or one per FOV named after it (row_col_fov)")
@click.option("--local", "-l", is_flag=True, default=False)
def stabilize_cli(input_position_dirpaths, output_dirpath, sbatch_filepath, monitor, config_filepaths, local):
    """Apply per-timepoint affines to every channel (reference: ``biahub stabilize``, stabilize.py:93-318: all channels
    of the input are stabilized — ``stabilization_channels = channel_names`` — over ``time_indices``; the output scale
    is ``output_voxel_size``; with several config files each FOV takes the one whose name contains ``row_col_fov``)."""
    from scipy.spatial.transform import Rotation

    from .array_ops import copy_n_paste_czyx
    from .stabilize import apply_stabilization_transform

    config_filepaths = list(config_filepaths)
    settings = yaml_to_model(config_filepaths[0], StabilizationSettings)
    with open_ome_zarr(input_position_dirpaths[0]) as ds:
        names, (T, C, Z, Y, X), version = ds.channel_names, ds.data.shape, ds.version
    stabilization_channels = list(names)  # stabilize.py:150-151
    # a transform that rotates ~90 degrees about x swaps the output's Y and X extents (stabilize.py:165-181)
    U, _, Vt = np.linalg.svd(np.asarray(settings.affine_transform_zyx_list[0], dtype=np.float64)[:3, :3])
    euler = Rotation.from_matrix(U @ Vt).as_euler("xyz", degrees=True)
    out_zyx = (Z, X, Y) if np.isclose(euler[0], 90, atol=10) else (Z, Y, X)
    if settings.time_indices == "all":
        tidx = list(range(T))
    elif isinstance(settings.time_indices, int):
        tidx = [settings.time_indices]
    else:
        tidx = list(settings.time_indices)
    _create_plate_once(output_dirpath, [p.parts[-3:] for p in input_position_dirpaths], names,
                       (len(tidx), len(names)) + out_zyx, scale=list(settings.output_voxel_size),
                       version=settings.output_ome_zarr_version or version, dtype=np.float32,
                       compressor=_output_compressor())
    _, cpus, gb = estimate_resources((T, C, Z, Y, X), ram_multiplier=16, max_num_cpus=16)
    click.echo(f"Preparing jobs: {{'slurm_job_name': 'stabilize', 'slurm_mem_per_cpu': '{gb}G', "
               f"'slurm_cpus_per_task': {cpus}}}")
    if sbatch_filepath:
        sbatch_to_submitit(sbatch_filepath)
    _resolve_cluster(None, local=True)
    otidx = list(range(len(tidx)))  # the output holds len(time_indices) frames, in the order listed
    full = [slice(0, Z), slice(0, Y), slice(0, X)]

    def job(src, dst):
        cfg = config_filepaths[0]
        if len(config_filepaths) > 1:  # per-FOV settings (stabilize.py:262-267)
            fov = "_".join(Path(src).parts[-3:])
            hits = [p for p in config_filepaths if fov in p.name]
            if not hits:
                raise ValueError(f"no stabilization config among {len(config_filepaths)} files matches FOV {fov}")
            cfg = hits[0]
        fov_settings = yaml_to_model(cfg, StabilizationSettings)
        mats = [np.asarray(m, dtype=np.float64) for m in fov_settings.affine_transform_zyx_list]
        if tidx and len(mats) <= max(tidx):
            raise ValueError(f"{cfg}: {len(mats)} transforms, but time index {max(tidx)} is requested")
        for name in names:
            c = [[names.index(name)]]
            if name in stabilization_channels:
                process_single_position(apply_stabilization_transform, src, dst, input_channel_indices=c,
                                        output_channel_indices=c, input_time_indices=tidx, output_time_indices=otidx,
                                        list_of_shifts=mats, output_shape=out_zyx)
            else:  # unreachable while every channel is stabilized, kept as the reference keeps it (stabilize.py:291-304)
                process_single_position(copy_n_paste_czyx, src, dst, input_channel_indices=c, output_channel_indices=c,
                                        input_time_indices=tidx, output_time_indices=otidx, czyx_slicing_params=full)

    outs = get_output_paths(input_position_dirpaths, output_dirpath)
    _run_positions("stabilize", input_position_dirpaths, outs, job, Path(output_dirpath).parent)


@cli.command("register")
@click.option("--source-position-dirpaths", "-s", multiple=True, required=True, callback=_positions)
@click.option("--target-position-dirpaths", "-t", multiple=True, required=True, callback=_positions)
@_config
@click.option("--output-dirpath", "-o", required=True, type=click.Path(path_type=Path))
@click.option("--local", "-l", is_flag=True, default=False)
def register_cli(source_position_dirpaths, target_position_dirpaths, config_filepath, output_dirpath, local):
    """Warp source channels onto the target grid and copy the target channel (reference: ``biahub register``).
    With ``keep_overhang: false`` the output is cropped to the largest interior cuboid of the overlap
    (register.py:469-478)."""
    from .array_ops import copy_n_paste_czyx
    from .register import apply_affine_transform, find_overlapping_volume, rescale_voxel_size

    settings = yaml_to_model(config_filepath, RegistrationSettings)
    M = np.asarray(settings.affine_transform_zyx, dtype=np.float64)
    with open_ome_zarr(source_position_dirpaths[0]) as src, open_ome_zarr(target_position_dirpaths[0]) as tgt:
        src_names, tgt_names = src.channel_names, tgt.channel_names
        T = min(src.data.shape[0], tgt.data.shape[0])
        tgt_shape = tgt.data.shape[-3:]
        src_shape = src.data.shape[-3:]
        out_scale = (1, 1) + tuple(rescale_voxel_size(M[:3, :3], np.asarray(src.scale[-3:])))
    crop = None
    out_zyx = tuple(tgt_shape)
    if not settings.keep_overhang:
        crop = find_overlapping_volume(src_shape, tgt_shape, M)
        out_zyx = tuple(s.stop - s.start for s in crop)
        click.echo(f"Cropping to the overlapping volume: {crop}")
    out_names = list(dict.fromkeys(list(settings.source_channel_names) + [settings.target_channel_name]))
    _create_plate_once(output_dirpath, [p.parts[-3:] for p in source_position_dirpaths], out_names,
                       (T, len(out_names)) + out_zyx, scale=out_scale, compressor=_output_compressor())
    outs = get_output_paths(source_position_dirpaths, output_dirpath)
    tidx = list(range(T)) if settings.time_indices == "all" else list(np.atleast_1d(settings.time_indices))

    def job(pair_src, dst):
        tpath = target_position_dirpaths[list(source_position_dirpaths).index(pair_src)]
        for name in settings.source_channel_names:
            process_single_position(apply_affine_transform, pair_src, dst,
                                    input_channel_indices=[[src_names.index(name)]],
                                    output_channel_indices=[[out_names.index(name)]], input_time_indices=tidx,
                                    output_time_indices=tidx, matrix=M, output_shape_zyx=tuple(tgt_shape),
                                    interpolation=settings.interpolation, crop_output_slicing=crop)
        if settings.target_channel_name not in settings.source_channel_names:
            full = list(crop) if crop is not None else [slice(0, n) for n in tgt_shape]
            process_single_position(copy_n_paste_czyx, tpath, dst,
                                    input_channel_indices=[[tgt_names.index(settings.target_channel_name)]],
                                    output_channel_indices=[[out_names.index(settings.target_channel_name)]],
                                    input_time_indices=tidx, output_time_indices=tidx, czyx_slicing_params=full)

    _run_positions("register", source_position_dirpaths, outs, job, Path(output_dirpath).parent)


@cli.command("estimate-registration")
@click.option("--source-position-dirpaths", "-s", multiple=True, required=True, callback=_positions)
@click.option("--target-position-dirpaths", "-t", multiple=True, required=True, callback=_positions)
@click.option("--output-filepath", "-o", required=True, type=click.Path(path_type=Path))
@_config
@click.option("--sbatch-filepath", "-sb", default=None, type=click.Path(path_type=Path))
@click.option("--local", "-l", is_flag=True, default=False)
@click.option("--registration-target-channel", "-rt", default=None, type=str)
@click.option("--registration-source-channel", "-rs", multiple=True, type=str)
def estimate_registration_cli(source_position_dirpaths, target_position_dirpaths, output_filepath, config_filepath,
                              sbatch_filepath, local, registration_target_channel, registration_source_channel):
    """Estimate the source -> target transform and save it as ``register`` / ``stabilize`` settings
    (reference: ``biahub estimate-registration``, estimate_registration.py:358-536).  ``estimation_method: ants``
    runs on the GPU (Mattes-MI similarity estimate); ``manual`` (napari) and ``beads`` are not part of this package."""
    from .registration.ants import estimate_tczyx

    output_filepath = Path(output_filepath)
    output_dir = output_filepath.parent
    output_dir.mkdir(parents=True, exist_ok=True)
    settings = yaml_to_model(config_filepath, EstimateRegistrationSettings)
    click.echo(f"Settings: {settings}")
    if settings.estimation_method != "ants":
        raise click.UsageError(f"estimation_method {settings.estimation_method!r} is not available in biahub_amd "
                               "(only 'ants'); use the reference biahub package for 'manual' and 'beads'.")
    target_channel_name, source_channel_name = settings.target_channel_name, settings.source_channel_name
    reg_target = registration_target_channel or target_channel_name
    reg_sources = list(registration_source_channel) or [source_channel_name]
    click.echo(f"Target channel: {target_channel_name}")
    click.echo(f"Source channel: {source_channel_name}")
    with open_ome_zarr(source_position_dirpaths[0]) as src, open_ome_zarr(target_position_dirpaths[0]) as tgt:
        source_channel_index = src.channel_names.index(source_channel_name)
        target_channel_index = tgt.channel_names.index(target_channel_name)
        voxel_size = list(tgt.scale)
        transforms = estimate_tczyx(
            mov_tczyx=src.data, ref_tczyx=tgt.data, mov_channel_index=source_channel_index,
            ref_channel_index=target_channel_index, ants_registration_settings=settings.ants_registration_settings,
            affine_transform_settings=settings.affine_transform_settings, verbose=settings.verbose,
            output_folder_path=output_dir, cluster="local", sbatch_filepath=sbatch_filepath)
    if len(transforms) == 1:
        model = RegistrationSettings(source_channel_names=reg_sources, target_channel_name=reg_target,
                                     affine_transform_zyx=transforms[0])
    else:
        model = StabilizationSettings(stabilization_estimation_channel=target_channel_name, stabilization_type="affine",
                                      stabilization_method="ants",
                                      stabilization_channels=[source_channel_name, target_channel_name],
                                      affine_transform_zyx_list=transforms, time_indices="all",
                                      output_voxel_size=voxel_size)
    if parallel.world_info()[0] == 0:
        model_to_yaml(model, output_filepath)
    click.echo(f"Registration settings saved to {output_dir.resolve()}")


@cli.command("optimize-registration")
@click.option("--source-position-dirpaths", "-s", multiple=True, required=True, callback=_positions)
@click.option("--target-position-dirpaths", "-t", multiple=True, required=True, callback=_positions)
@_config
@click.option("--output-filepath", "-o", required=True, type=click.Path(path_type=Path))
@click.option("--display-viewer", "-d", is_flag=True, help="(napari viewer of the reference: not available here)")
def optimize_registration_cli(source_position_dirpaths, target_position_dirpaths, config_filepath, output_filepath,
                              display_viewer):
    """Refine an approximate source -> target transform on the overlapping region (reference: ``biahub
    optimize-registration``, optimize_registration.py:196-312): the ``affine_transform_zyx`` of the input settings is the
    initial guess, the Mattes-MI similarity estimate corrects it on the GPU, the same settings come back with the composed
    matrix."""
    from .registration.ants import estimate_czyx

    settings = yaml_to_model(config_filepath, RegistrationSettings)
    t_idx = settings.time_indices
    if not isinstance(t_idx, int):
        click.echo("Time index 'all' is not supported for optimize-registration, using first time index")
        t_idx = 0
    with open_ome_zarr(source_position_dirpaths[0]) as src, open_ome_zarr(target_position_dirpaths[0]) as tgt:
        source_channel_index = src.channel_names.index(settings.source_channel_names[0])
        target_channel_index = tgt.channel_names.index(settings.target_channel_name)
        source_czyx, target_czyx = src.data[t_idx], tgt.data[t_idx]
        click.echo(f"\nOptimizing registration using source channel {src.channel_names[source_channel_index]} and target "
                   f"channel {tgt.channel_names[target_channel_index]}")
    approx = np.asarray(settings.affine_transform_zyx, dtype=np.float32)
    composed = estimate_czyx(mov_czyx=source_czyx, ref_czyx=target_czyx, initial_tform=approx,
                             mov_channel_index=source_channel_index, ref_channel_index=target_channel_index, crop=True,
                             verbose=settings.verbose)
    click.echo(f"Writing registration parameters to {output_filepath}")
    out = settings.model_copy()
    out.affine_transform_zyx = composed.matrix.tolist()
    Path(output_filepath).parent.mkdir(parents=True, exist_ok=True)
    model_to_yaml(out, output_filepath)
    if display_viewer:
        click.echo("The napari viewer is not part of biahub_amd; inspect the result with `register`.")


@cli.command("estimate-stabilization")
@click.option("--input-position-dirpaths", "-i", multiple=True, required=True, callback=_positions)
@click.option("--output-dirpath", "-o", required=True, type=click.Path(path_type=Path))
@_config
@click.option("--sbatch-filepath", "-sb", default=None, type=click.Path(path_type=Path))
@click.option("--local", "-l", is_flag=True, default=False)
def estimate_stabilization_cli(input_position_dirpaths, output_dirpath, config_filepath, sbatch_filepath, local):
    """Estimate per-timepoint stabilization transforms (reference: ``biahub estimate-stabilization``,
    estimate_stabilization.py:1223-1640).  ``stabilization_type: xyz`` with ``stabilization_method: phase-cross-corr``
    runs here (GPU phase cross-correlation); one ``xyz_stabilization_settings/<fov>.yml`` per position feeds ``stabilize``."""
    from .estimate_stabilization import estimate_xyz_stabilization_pcc, save_transforms

    settings = yaml_to_model(config_filepath, EstimateStabilizationSettings)
    click.echo(f"Settings: {settings}")
    if not (settings.stabilization_type == "xyz" and settings.stabilization_method == "phase-cross-corr"):
        raise click.UsageError(f"stabilization_type {settings.stabilization_type!r} with method "
                               f"{settings.stabilization_method!r} is not available in biahub_amd (only 'xyz' with "
                               "'phase-cross-corr'); use the reference biahub package for focus finding and beads.")
    if settings.eval_transform_settings:
        raise click.UsageError("eval_transform_settings (transform validation / interpolation) is not available in biahub_amd")
    output_dirpath = Path(output_dirpath)
    output_dirpath.mkdir(parents=True, exist_ok=True)
    with open_ome_zarr(input_position_dirpaths[0]) as ds:
        channel_index = ds.channel_names.index(settings.stabilization_estimation_channel)
        voxel_size = list(ds.scale)
    click.echo("Estimating xyz stabilization parameters with phase cross correlation")
    pcc = settings.phase_cross_corr_settings or PhaseCrossCorrSettings()
    fov_transforms = estimate_xyz_stabilization_pcc(input_position_dirpaths, output_dirpath, pcc, channel_index=channel_index,
                                                    sbatch_filepath=sbatch_filepath, cluster="local", verbose=settings.verbose)
    if parallel.world_info()[0] == 0:
        for fov, transforms in fov_transforms.items():
            model = StabilizationSettings(stabilization_type=settings.stabilization_type,
                                          stabilization_method=settings.stabilization_method,
                                          stabilization_estimation_channel=settings.stabilization_estimation_channel,
                                          stabilization_channels=settings.stabilization_channels,
                                          affine_transform_zyx_list=[np.eye(4).tolist()], time_indices="all",
                                          output_voxel_size=voxel_size)
            save_transforms(model, transforms, output_dirpath / "xyz_stabilization_settings" / f"{fov}.yml")
    click.echo(f"Stabilization settings saved to {output_dirpath.resolve()}")


@cli.command("estimate-psf")
@click.option("--input-position-dirpaths", "-i", multiple=True, required=True, callback=_positions)
@_config
@click.option("--output-dirpath", "-o", required=True, type=click.Path(path_type=Path))
def estimate_psf_cli(input_position_dirpaths, config_filepath, output_dirpath):
    """Estimate the PSF from bead volumes and save it as ``<out>/0/0/0`` (reference: ``biahub estimate-psf``,
    estimate_psf.py:19-121): (t, c) = (0, 0) of every position, beads detected, recentred, averaged on the GPU."""
    from .estimate_psf import estimate_psf

    settings = yaml_to_model(config_filepath, PsfFromBeadsSettings)
    click.echo("Loading data...")
    pzyx, zyx_scale = [], None
    for p in input_position_dirpaths:
        with open_ome_zarr(p) as ds:
            pzyx.append(ds.data[0, 0])
            zyx_scale = tuple(ds.scale[-3:])
    if len({v.shape for v in pzyx}) != 1:
        raise ValueError("Concatenating position arrays failed.")
    patch = (settings.axis0_patch_size, settings.axis1_patch_size, settings.axis2_patch_size)
    click.echo("Detecting beads...")
    psf = estimate_psf(pzyx, zyx_scale, patch_size=patch, verbose=True)
    create_empty_plate(output_dirpath, [("0", "0", "0")], ["PSF"], (1, 1) + psf.shape, chunks=(1, 1) + psf.shape,
                       scale=(1, 1) + zyx_scale, dtype=np.float32, compressor=_output_compressor())
    open_ome_zarr(Path(output_dirpath) / "0/0/0").data[0, 0] = psf
    click.echo(f"PSF saved to {Path(output_dirpath).resolve()}")


@cli.command("concatenate")
@_config
@click.option("--output-dirpath", "-o", required=True, type=click.Path(path_type=Path),
              help="Path to output.zarr (or, with --concat-data-paths, to the resolved YAML)")
@click.option("--sbatch-filepath", "-sb", default=None, type=click.Path(exists=True), help="Accepted for compatibility.")
@click.option("--cluster", default=None, type=click.Choice(["slurm", "local", "debug"]))
@click.option("--monitor", "-m", is_flag=True, default=False, help="Accepted for compatibility.")
@click.option("--init", "init_only", is_flag=True, default=False, help="Create the output store, print RESOURCES, exit.")
@click.option("--resume", is_flag=True, default=False, help="Skip the (time, channel) units a previous attempt finished.")
@click.option("--concat-data-paths", multiple=True, type=str,
              help="Resolve mode: inject these concat_data_paths into the config and write it to -o (a YAML file), then exit.")
def concatenate_cli(config_filepath, output_dirpath, sbatch_filepath, cluster, monitor, init_only, resume, concat_data_paths):
    """Concatenate datasets, with optional cropping (reference: ``biahub concatenate``, biahub/concatenate.py:556-636)."""
    import yaml

    from .concatenate import concatenate
    from .settings import ConcatenateSettings

    if concat_data_paths:  # the placeholder is filled in BEFORE validation: a blank list would not validate
        raw = yaml.safe_load(Path(config_filepath).read_text())
        raw["concat_data_paths"] = list(concat_data_paths)
        model_to_yaml(ConcatenateSettings(**raw), output_dirpath)
        click.echo(f"Resolved config written to {output_dirpath}")
        return
    settings = yaml_to_model(config_filepath, ConcatenateSettings)
    if not init_only:
        _resolve_cluster(cluster)
    if sbatch_filepath:
        sbatch_to_submitit(sbatch_filepath)
    rank, world = (0, 1) if init_only else parallel.init()
    if rank != 0:  # one rank lays the plate out; the others wait for it
        parallel.barrier()
    prep = concatenate(settings, output_dirpath, init_only=True, compressor=_output_compressor()) if rank == 0 else None
    if rank == 0:
        T, C, Z, Y, X = prep["shape"]
        batch = settings.shards_ratio[0] if settings.shards_ratio else 1
        _, cpus, gb = estimate_resources((T // batch, C, Z, Y, X), ram_multiplier=8 * batch, max_num_cpus=16)
        echo_resources(cpus, cpus * gb, 360)
        if not init_only and world > 1:
            parallel.barrier()
    if init_only:
        return
    log_dir = Path(output_dirpath).parent / "slurm_output"
    prep = concatenate(settings, output_dirpath, resume=resume, compressor=_output_compressor(), rank=rank, world=world,
                       create=False)  # rank 0 laid the plate out above
    if rank == 0:
        log_dir.mkdir(exist_ok=True)
        (log_dir / "submitit_jobs_ids.log").write_text("\n".join(f"concatenate-{i}" for i in range(len(prep["all_data_paths"]))))
    parallel.barrier()


@cli.command("estimate-crop")
@_config
@click.option("--output-filepath", "-o", required=True, type=click.Path(path_type=Path), help="Path to the output YAML config")
@click.option("--lf-mask-radius", type=float, default=0.95,
              help="Radius of the circular mask, as a fraction of the image width, applied to the phase channel.")
@click.option("--sbatch-filepath", "-sb", default=None, type=click.Path(exists=True), help="Accepted for compatibility.")
@click.option("--local", "-l", is_flag=True, default=False, help="Accepted for compatibility: this build always runs in-process.")
def estimate_crop_cli(config_filepath, output_filepath, lf_mask_radius, sbatch_filepath, local):
    """Estimate the crop in which both the phase and the fluorescence volumes carry data, and write it into a copy of the
    concatenate configuration (reference: ``biahub estimate-crop``, biahub/estimate_crop.py:285-320)."""
    from .estimate_crop import estimate_crop

    if sbatch_filepath:
        sbatch_to_submitit(sbatch_filepath)
    estimate_crop(config_filepath, output_filepath, lf_mask_radius=lf_mask_radius)


@cli.command("flip")
@click.option("--input-position-dirpaths", "-i", multiple=True, required=True, callback=_positions)
@click.option("-x", is_flag=True, help="Flip along x.")
@click.option("-y", is_flag=True, help="Flip along y.")
def flip_cli(input_position_dirpaths, x, y):
    """Flip every (t, c) volume of the positions in place (reference: ``biahub flip``)."""
    from .array_ops import flip_zyx

    for p in input_position_dirpaths:
        click.echo(f"Flipping {p}")
        arr = open_ome_zarr(p).data
        for t in range(arr.shape[0]):
            for c in range(arr.shape[1]):
                arr[t, c] = flip_zyx(arr[t, c], x=x, y=y)


def main(argv=None):
    argv = expand_eat_all(list(sys.argv[1:] if argv is None else argv))
    return cli.main(args=argv, standalone_mode=True)


if __name__ == "__main__":
    main()
