"""Settings models of the hot-path steps — the YAML surface kept from ``biahub/settings.py``.

Field names, defaults, rounding and validation errors follow the reference so the same YAML files load:
``DeskewSettings`` (settings.py:348-383), ``RegistrationSettings`` (:386-415), ``DeconvolveSettings`` (:424-427),
``StabilizationSettings`` (:624-650).  Unknown keys are rejected (``extra="forbid"``, :22-23).
"""

from __future__ import annotations

from typing import Literal

import numpy as np
from pydantic import (BaseModel, ConfigDict, NonNegativeFloat, NonNegativeInt, PositiveFloat, PositiveInt, field_validator,
                      model_validator)

OmeZarrVersion = Literal["0.4", "0.5"]


class _Strict(BaseModel):
    model_config = ConfigDict(extra="forbid")


class ProcessingSettings(_Strict):
    fliplr: bool | None = False
    flipud: bool | None = False
    rot90: int | None = 0


class ProcessingFunctions(_Strict):
    """biahub/settings.py:33-37."""

    function: str
    input_channels: list | None = None
    kwargs: dict = {}
    per_timepoint: bool | None = True


class ProcessingImportFuncSettings(_Strict):
    """biahub/settings.py:40-43."""

    processing_functions: list[ProcessingFunctions] = []
    output_ome_zarr_version: OmeZarrVersion | None = None


class FlatFieldCorrectionSettings(_Strict):
    """biahub/settings.py:336-339."""

    channel_names: list[str] | None = None
    output_ome_zarr_version: OmeZarrVersion | None = None


class DeskewSettings(_Strict):
    pixel_size_um: PositiveFloat
    ls_angle_deg: PositiveFloat
    px_to_scan_ratio: PositiveFloat | None = None
    scan_step_um: PositiveFloat | None = None
    keep_overhang: bool = False
    overhang_fill: Literal["mean"] | float = 0
    average_n_slices: PositiveInt = 3
    device: str = "cpu"
    output_ome_zarr_version: OmeZarrVersion | None = None

    @model_validator(mode="before")
    @classmethod
    def _derive_ratio(cls, data):
        # px_to_scan_ratio defaults to pixel_size_um / scan_step_um, 3 decimals (settings.py:373-382)
        if isinstance(data, dict) and data.get("px_to_scan_ratio") is None:
            if data.get("scan_step_um") is None:
                raise ValueError(
                    "If px_to_scan_ratio is not provided, both pixel_size_um and scan_step_um must be provided"
                )
            data = dict(data)
            data["px_to_scan_ratio"] = round(data["pixel_size_um"] / data["scan_step_um"], 3)
        return data

    @field_validator("ls_angle_deg")
    @classmethod
    def _angle(cls, v):
        if v < 0 or v > 45:
            raise ValueError("Light sheet angle must be be between 0 and 45 degrees")
        return round(float(v), 2)

    @field_validator("px_to_scan_ratio")
    @classmethod
    def _ratio(cls, v):
        return None if v is None else round(float(v), 3)


def _check_4x4(m, what):
    a = np.asarray(m, dtype=object)
    if not isinstance(m, list) or len(m) != 4 or any((not isinstance(r, list)) or len(r) != 4 for r in m):
        raise ValueError(f"{what} must be a 4x4 nested list")
    try:
        np.asarray(m, dtype=float)
    except (TypeError, ValueError):
        raise ValueError("The array must contain valid numerical values.") from None
    del a
    return m


class RegistrationSettings(_Strict):
    source_channel_names: list[str]
    target_channel_name: str
    affine_transform_zyx: list
    keep_overhang: bool = False
    interpolation: str = "linear"
    time_indices: NonNegativeInt | list[NonNegativeInt] | Literal["all"] = "all"
    verbose: bool = False
    output_ome_zarr_version: OmeZarrVersion | None = None

    @field_validator("affine_transform_zyx")
    @classmethod
    def _affine(cls, v):
        return _check_4x4(v, "affine_transform_zyx")


class AffineTransformSettings(_Strict):
    """biahub/settings.py:240-257."""

    t_reference: Literal["first", "previous"] = "first"
    transform_type: Literal["euclidean", "similarity", "affine"] = "euclidean"
    approx_transform: list = np.eye(4).tolist()
    use_prev_t_transform: bool = True
    compute_approx_transform: bool = False

    @field_validator("approx_transform")
    @classmethod
    def _approx(cls, v):
        if v is not None:
            if not isinstance(v, list):
                raise ValueError("approx_transform must be a list")
            if np.array(v).shape != (4, 4):
                raise ValueError("approx_transform must be a 4x4 array")
        return v


class AntsRegistrationSettings(_Strict):
    """biahub/settings.py:260-261 plus the preprocessing switches ``estimate_tczyx`` reads from it
    (registration/ants.py:505-508: crop, ref_mask_radius, clip)."""

    sobel_filter: bool = False
    crop: bool = False
    ref_mask_radius: float | None = None
    clip: bool = False


class EstimateRegistrationSettings(_Strict):
    """biahub/settings.py:270-292; only ``estimation_method: ants`` runs here (manual needs napari, beads is a CPU
    point-matching flow: SURVEY.md §8f), the other methods' sub-settings are carried through untouched."""

    target_channel_name: str
    source_channel_name: str
    estimation_method: Literal["manual", "beads", "ants"] = "manual"
    beads_match_settings: dict | None = None
    focus_finding_settings: dict | None = None
    affine_transform_settings: AffineTransformSettings = AffineTransformSettings()
    eval_transform_settings: dict | None = None
    ants_registration_settings: AntsRegistrationSettings | None = None
    manual_registration_settings: dict | None = None
    verbose: bool = False

    @model_validator(mode="after")
    def _defaults(self):
        if self.estimation_method == "ants" and self.ants_registration_settings is None:
            self.ants_registration_settings = AntsRegistrationSettings()
        return self


class PsfFromBeadsSettings(_Strict):
    """biahub/settings.py:418-421."""

    axis0_patch_size: PositiveInt = 101
    axis1_patch_size: PositiveInt = 101
    axis2_patch_size: PositiveInt = 101


class PhaseCrossCorrSettings(_Strict):
    """biahub/settings.py:205-214."""

    normalization: Literal["magnitude", "classic"] | None = None
    maximum_shift: float = 1.2
    function_type: Literal["custom_padding", "custom"] = "custom"
    t_reference: Literal["first", "previous"] = "first"
    skip_beads_fov: str = "0"
    center_crop_xy: list[int] | None = None
    X_slice: list | Literal["all"] = "all"
    Y_slice: list | Literal["all"] = "all"
    Z_slice: list | Literal["all"] = "all"


class EstimateStabilizationSettings(_Strict):
    """biahub/settings.py:295-310; only ``phase-cross-corr`` runs in this package, the other methods' sub-settings are
    carried as plain dicts."""

    stabilization_estimation_channel: str
    stabilization_channels: list
    stabilization_type: Literal["z", "xy", "xyz"]
    stabilization_method: Literal["beads", "phase-cross-corr", "focus-finding"] = "focus-finding"
    beads_match_settings: dict | None = None
    phase_cross_corr_settings: PhaseCrossCorrSettings | None = None
    stack_reg_settings: dict | None = None
    focus_finding_settings: dict | None = None
    affine_transform_settings: AffineTransformSettings = AffineTransformSettings()
    eval_transform_settings: dict | None = None
    verbose: bool = False


class DeconvolveSettings(_Strict):
    regularization_strength: PositiveFloat = 0.001
    output_ome_zarr_version: OmeZarrVersion | None = None


class RichardsonLucySettings(_Strict):
    """North-star extension (no reference model): parameters of ``richardson_lucy_czyx``."""

    iterations: PositiveInt = 10
    eps: PositiveFloat = 1e-6
    output_ome_zarr_version: OmeZarrVersion | None = None


class StabilizationSettings(_Strict):
    stabilization_estimation_channel: str
    stabilization_type: Literal["z", "xy", "xyz", "affine"]
    stabilization_method: Literal["beads", "phase-cross-corr", "focus-finding", "manual", "ants"] = "focus-finding"
    stabilization_channels: list
    affine_transform_zyx_list: list
    time_indices: NonNegativeInt | list[NonNegativeInt] | Literal["all"] = "all"
    output_voxel_size: list[PositiveFloat] = [1.0, 1.0, 1.0, 1.0, 1.0]
    output_ome_zarr_version: OmeZarrVersion | None = None

    @field_validator("affine_transform_zyx_list")
    @classmethod
    def _affines(cls, v):
        if not isinstance(v, list):
            raise ValueError("affine_transform_list must be a list")
        for m in v:
            if np.asarray(m).shape != (4, 4):
                raise ValueError("Each element in affine_transform_list must be a 4x4 ndarray")
        return v


def _is_range(x) -> bool:
    return isinstance(x, list) and len(x) == 2 and all(isinstance(i, int) for i in x)


def _nonneg(r):
    if not all(i >= 0 for i in r):
        raise ValueError("Slice indices must be non-negative integers.")


class ConcatenateSettings(_Strict):
    """`biahub/settings.py:452-620`: what to concatenate (glob per source), which channels, optional per-source crops,
    output chunking / sharding and NGFF version (0.5 by default: concatenate is the migration path into v3 stores)."""

    concat_data_paths: list[str]
    time_indices: int | list[int] | Literal["all"] = "all"
    channel_names: list[str | list[str]]
    X_slice: list | list[list | Literal["all"]] | Literal["all"] = "all"
    Y_slice: list | list[list | Literal["all"]] | Literal["all"] = "all"
    Z_slice: list | list[list | Literal["all"]] | Literal["all"] = "all"
    chunks_czyx: Literal[None] | list[int] = None
    shards_ratio: list[int] | None = None
    ensure_unique_positions: bool | None = False
    output_ome_zarr_version: OmeZarrVersion | None = "0.5"

    @field_validator("X_slice", "Y_slice", "Z_slice")
    @classmethod
    def _check_slice(cls, v):
        if v == "all":
            return v
        if not isinstance(v, list):
            raise ValueError("Slice must be 'all' or a list.")
        nested = any(isinstance(item, list) and any(isinstance(sub, list) for sub in item) for item in v)
        if nested:  # one specification per source, each possibly a list of ranges itself
            for item in v:
                if item == "all":
                    continue
                if _is_range(item):
                    _nonneg(item)
                elif isinstance(item, list):
                    for sub in item:
                        if sub == "all":
                            continue
                        if not _is_range(sub):
                            raise ValueError("Each slice subitem must be 'all' or a list of two non-negative integers [start, end].")
                        _nonneg(sub)
                else:
                    raise ValueError("Each item in a per-path slice list must be 'all' or a valid slice specification.")
            return v
        if _is_range(v):
            _nonneg(v)
            return v
        for item in v:
            if item == "all":
                continue
            if not _is_range(item):
                raise ValueError("Each slice item must be 'all' or a list of two non-negative integers [start, end].")
            _nonneg(item)
        return v

    @field_validator("chunks_czyx")
    @classmethod
    def _check_chunks(cls, v):
        if v is not None and (not isinstance(v, list) or len(v) != 4 or not all(isinstance(i, int) for i in v)):
            raise ValueError("chunks_czyx must be a list of 4 integers (C, Z, Y, X)")
        return v

    @model_validator(mode="after")
    def _check_slice_lengths(self):
        n = len(self.concat_data_paths)
        if n:
            for axis in ("X", "Y", "Z"):
                s = getattr(self, f"{axis}_slice")
                if isinstance(s, list) and len(s) != n and not _is_range(s):
                    raise ValueError(f"{axis}_slice must be 'all', a single slice specification, or a list with the same length "
                                     f"as concat_data_paths ({n})")
        return self


# ----------------------------------------------------------------------------- reconstruction (compute-tf / apply-inv-tf)
# The YAML surface of waveorder's ReconstructionSettings (waveorder/cli/settings.py, 3.0.5 — absent from the reference tree,
# restated from its published models and from the configurations the reference ships and tests with:
# nextflow/configs/a549/reconstruct.yml, tests/test_cli/test_reconstruct_cli.py:13-37).  biahub validates its configs
# against that model (biahub/apply_inverse_transfer_function.py:113).
class FourierApplyInverseSettings(_Strict):
    reconstruction_algorithm: Literal["Tikhonov", "TV"] = "Tikhonov"
    regularization_strength: NonNegativeFloat = 1e-3
    TV_rho_strength: PositiveFloat = 1e-3
    TV_iterations: NonNegativeInt = 1


class _FourierTransferFunctionSettings(_Strict):
    yx_pixel_size: PositiveFloat | None = None  # None: the input store's scale
    z_pixel_size: PositiveFloat | None = None
    z_padding: NonNegativeInt = 0
    z_focus_offset: int | float | Literal["auto"] = 0
    index_of_refraction_media: PositiveFloat = 1.3
    numerical_aperture_detection: PositiveFloat = 1.2

    @model_validator(mode="after")
    def _na_below_index(self):
        if self.numerical_aperture_detection > self.index_of_refraction_media:
            raise ValueError("numerical_aperture_detection must not exceed index_of_refraction_media")
        return self


class PhaseTransferFunctionSettings(_FourierTransferFunctionSettings):
    wavelength_illumination: PositiveFloat = 0.532
    numerical_aperture_illumination: NonNegativeFloat = 0.5
    invert_phase_contrast: bool = False

    @model_validator(mode="after")
    def _na_ill(self):
        if self.numerical_aperture_illumination > self.numerical_aperture_detection:
            raise ValueError("numerical_aperture_illumination must not exceed numerical_aperture_detection")
        return self


class FluorescenceTransferFunctionSettings(_FourierTransferFunctionSettings):
    wavelength_emission: PositiveFloat = 0.507


class PhaseSettings(_Strict):
    transfer_function: PhaseTransferFunctionSettings = PhaseTransferFunctionSettings()
    apply_inverse: FourierApplyInverseSettings = FourierApplyInverseSettings()


class FluorescenceSettings(_Strict):
    transfer_function: FluorescenceTransferFunctionSettings = FluorescenceTransferFunctionSettings()
    apply_inverse: FourierApplyInverseSettings = FourierApplyInverseSettings()


class ReconstructionSettings(_Strict):
    input_channel_names: list[str] = [f"State{i}" for i in range(4)]
    time_indices: NonNegativeInt | list[NonNegativeInt] | Literal["all"] = "all"
    reconstruction_dimension: Literal[2, 3] = 3
    birefringence: dict | None = None  # accepted so that such configs are refused with a clear message, not a schema error
    phase: PhaseSettings | None = None
    fluorescence: FluorescenceSettings | None = None

    @model_validator(mode="after")
    def _one_modality(self):
        if self.fluorescence is not None and (self.phase is not None or self.birefringence is not None):
            raise ValueError("fluorescence reconstructions cannot be combined with phase or birefringence")
        if self.fluorescence is None and self.phase is None and self.birefringence is None:
            raise ValueError("specify one of birefringence, phase, fluorescence")
        return self

    @property
    def output_channel_names(self) -> list[str]:
        """Channels of the reconstruction (waveorder get_reconstruction_output_metadata)."""
        d = self.reconstruction_dimension
        names = []
        if self.birefringence is not None:
            names += ["Retardance", "Orientation", "BF", "Pol"]
        if self.phase is not None:
            names.append(f"Phase{d}D")
        if self.fluorescence is not None:
            names.append(f"{self.input_channel_names[0]}_Density{d}D")
        return names
