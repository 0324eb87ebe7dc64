"""Intensity-based registration estimate on the GPU — mirror of ``biahub/registration/ants.py``.

The reference hands the two volumes to ``ants.registration(type_of_transform="Similarity",
aff_shrink_factors=(6, 3, 1), aff_iterations=(2100, 1200, 50), aff_smoothing_sigmas=(2, 1, 0))``
(``registration/ants.py:93-109``): ITK's multi-resolution gradient descent on the Mattes mutual-information
metric (32 bins, regular sampling at rate 0.2), initialised by aligning the centres of mass.  ANTs is a
third-party binary, so results cannot be bit-compared; this module keeps the reference's function names,
arguments, conventions (ZYX pull matrices, ``composed = initial @ shift_to_roi @ fwd @ shift_back``) and the
same optimisation recipe, with the data-parallel pieces in ``csrc/regmetric.hip``:

* pyramid level = ``bh_smooth_shrink`` (Gaussian of sigma voxels, integer shrink, centres kept aligned);
* metric value and derivative w.r.t. the 3x4 pull matrix = ``bh_mattes_mi``;
* the optimiser (a few 4x4 products per iteration) runs here on the host: gradient ascent with parameter scales
  from the physical shift at the volume corners, learning rate estimated at the start of each level so that the
  first step moves no corner by more than ``grad_step`` voxels, step halving when the metric drops, and the
  window-slope convergence test (window 10, threshold 1e-6).

Parity status: unpinned — tests compare the recovered matrix with a known ground truth.
"""

from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from ..array_ops import _check_nan_n_zeros
from ..core.transform import Transform
from ..device import as_device_volume, resolve_device
from ..register import affine_device, find_lir
from . import metric as _metric

DEFAULT_ANTS_KWARGS = {
    "type_of_transform": "Similarity",
    "aff_shrink_factors": (6, 3, 1),
    "aff_iterations": (2100, 1200, 50),
    "aff_smoothing_sigmas": (2, 1, 0),
}
_TRANSFORM_TYPES = ("Translation", "Rigid", "Similarity", "Affine")


def _cross_matrix(k: int) -> np.ndarray:
    e = np.zeros(3)
    e[k] = 1.0
    return np.array([[0, -e[2], e[1]], [e[2], 0, -e[0]], [-e[1], e[0], 0]])


def _rotation_from_vector(w: np.ndarray) -> np.ndarray:
    th = float(np.linalg.norm(w))
    if th < 1e-300:
        return np.eye(3)
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1.0 - np.cos(th)) * (K @ K)


def _level_matrix(factor, offset) -> np.ndarray:
    """level index -> full-resolution index."""
    L = np.eye(4)
    L[:3, :3] = np.diag(np.asarray(factor, dtype=np.float64))
    L[:3, 3] = np.asarray(offset, dtype=np.float64)
    return L


class _CentredTransform:
    """T(x) = A (x - c) + c + t with A restricted by the transform type; updates happen in a local chart."""

    def __init__(self, kind: str, centre, translation, planar: bool):
        self.kind = kind
        self.c = np.asarray(centre, dtype=np.float64)
        self.t = np.asarray(translation, dtype=np.float64).copy()
        self.A = np.eye(3)
        self.s = 1.0
        self.planar = planar

    def matrix(self) -> np.ndarray:
        m = np.eye(4)
        m[:3, :3] = self.A
        m[:3, 3] = self.c + self.t - self.A @ self.c
        return m

    def generators(self):
        gens = []
        if self.kind in ("Rigid", "Similarity"):
            for k in ((0,) if self.planar else (0, 1, 2)):  # planar: rotation about z only
                gens.append((_cross_matrix(k) @ self.A, np.zeros(3)))
        if self.kind == "Similarity":
            gens.append((self.A / self.s, np.zeros(3)))
        if self.kind == "Affine":
            for a in range(3):
                for b in range(3):
                    if self.planar and (a == 0 or b == 0):
                        continue
                    dA = np.zeros((3, 3))
                    dA[a, b] = 1.0
                    gens.append((dA, np.zeros(3)))
        for k in ((1, 2) if self.planar else (0, 1, 2)):
            dt = np.zeros(3)
            dt[k] = 1.0
            gens.append((np.zeros((3, 3)), dt))
        return gens

    def apply_update(self, gens, step):
        step = np.asarray(step, dtype=np.float64)
        dA = sum(s * g[0] for s, g in zip(step, gens))
        dt = sum(s * g[1] for s, g in zip(step, gens))
        if self.kind in ("Rigid", "Similarity"):
            nrot = 1 if self.planar else 3
            w = np.zeros(3)
            w[:nrot] = step[:nrot]  # planar: the only rotation generator is about z (axis 0)
            R = _rotation_from_vector(w) @ (self.A / self.s)
            if self.kind == "Similarity":
                self.s = self.s + step[nrot]
            self.A = self.s * R
        elif self.kind == "Affine":
            self.A = self.A + dA
        self.t = self.t + dt

    def copy_state(self):
        return (self.A.copy(), self.s, self.t.copy())

    def restore(self, st):
        self.A, self.s, self.t = st[0].copy(), st[1], st[2].copy()


def _converged(values, window=10, threshold=1e-6) -> bool:
    """Slope of the last `window` metric values (normalised by their magnitude) below `threshold`."""
    if len(values) < window:
        return False
    w = np.asarray(values[-window:], dtype=np.float64)
    scale = np.abs(w).mean()
    if scale == 0:
        return True
    x = np.arange(window) - (window - 1) / 2.0
    slope = float((x * (w - w.mean())).sum() / (x * x).sum()) / scale
    return slope < threshold  # maximising: a flat or falling profile has stopped improving


def estimate(ref, mov, verbose: bool = False, ants_kwargs: dict = None, device=None) -> tuple[Transform, Transform]:
    """Estimate the transform aligning ``mov`` to ``ref`` (``registration/ants.py:55-122``).

    Returns ``(fwd, inv)``: ``fwd`` is the ZYX pull matrix, ``ref(p) ~ mov(fwd p)`` — what
    ``ants.registration(...)["fwdtransforms"]`` read through ``Transform.from_ants`` gives the reference.
    2-D (Y, X) inputs are handled as one-plane volumes with in-plane parameters only.
    """
    ref_nd = np.ndim(ref) if not isinstance(ref, torch.Tensor) else ref.ndim
    mov_nd = np.ndim(mov) if not isinstance(mov, torch.Tensor) else mov.ndim
    if ref_nd not in (2, 3) or mov_nd not in (2, 3):
        raise ValueError(f"Images must be 2D or 3D, got ref.ndim={ref_nd}, mov.ndim={mov_nd}")
    if ref_nd != mov_nd:
        raise ValueError(f"Dimension mismatch: ref.ndim={ref_nd}, mov.ndim={mov_nd}")
    kw = dict(DEFAULT_ANTS_KWARGS)
    kw.update(ants_kwargs or {})
    kind = kw["type_of_transform"]
    if kind not in _TRANSFORM_TYPES:
        raise ValueError(f"type_of_transform must be one of {_TRANSFORM_TYPES}, got {kind!r}")
    shrinks, iters, sigmas = kw["aff_shrink_factors"], kw["aff_iterations"], kw["aff_smoothing_sigmas"]
    if not (len(shrinks) == len(iters) == len(sigmas)):
        raise ValueError("aff_shrink_factors, aff_iterations and aff_smoothing_sigmas must have the same length")
    bins = int(kw.get("aff_sampling", 32))
    rate = float(kw.get("aff_random_sampling_rate", 0.2))
    max_step = float(kw.get("grad_step", 0.25))
    stride = max(1, int(round(1.0 / rate)))

    planar = ref_nd == 2
    dev = resolve_device("cuda" if device is None else device)
    fixed = _as_f32_volume(ref, dev, planar)
    moving = _as_f32_volume(mov, dev, planar)
    T = _optimise(fixed, moving, _metric, kind, shrinks, iters, sigmas, bins, stride, max_step, planar, verbose)
    if planar:
        T2 = np.eye(3)
        T2[:2, :2], T2[:2, 2] = T[1:3, 1:3], T[1:3, 3]
        return Transform(T2), Transform(np.linalg.inv(T2))
    return Transform(T), Transform(np.linalg.inv(T))


def _optimise(fixed, moving, ops, kind, shrinks, iters, sigmas, bins, stride, max_step, planar, verbose=False):
    """The multi-resolution ascent.  ``ops`` provides image_stats / smooth_shrink / mattes_mi on whatever the volumes
    are (``registration.metric`` on device tensors in the product; the host-logic tests drive it with the CPU oracle)."""
    fs, ms = ops.image_stats(fixed), ops.image_stats(moving)
    tf = _CentredTransform(kind, fs["center_of_mass"], ms["center_of_mass"] - fs["center_of_mass"], planar)
    corners = np.array([[z, y, x] for z in (0, fixed.shape[0] - 1) for y in (0, fixed.shape[1] - 1)
                        for x in (0, fixed.shape[2] - 1)], dtype=np.float64)

    for level, (f, n_it, sg) in enumerate(zip(shrinks, iters, sigmas)):
        f3 = [1 if planar else int(f), int(f), int(f)]
        s3 = [0.0 if planar else float(sg), float(sg), float(sg)]
        if int(f) == 1 and float(sg) == 0.0:
            F, M, of, om = fixed, moving, (0, 0, 0), (0, 0, 0)
        else:
            F, of = ops.smooth_shrink(fixed, s3, f3)
            M, om = ops.smooth_shrink(moving, s3, f3)
        Lf, Lm_inv = _level_matrix(f3, of), np.linalg.inv(_level_matrix(f3, om))
        fst, mst = ops.image_stats(F), ops.image_stats(M)
        if not (fst["max"] > fst["min"] and mst["max"] > mst["min"]):
            raise ValueError("Failed to estimate registration transform: constant image at a pyramid level.")
        rng = (fst["min"], fst["max"], mst["min"], mst["max"])

        def evaluate():
            P = (Lm_inv @ tf.matrix() @ Lf)[:3]
            val, G, n = ops.mattes_mi(F, M, P, rng, bins=bins, stride=stride)
            G4 = np.zeros((4, 4))
            G4[:3] = G
            D = (Lm_inv.T @ G4 @ Lf.T)[:3]  # dMI/dT in full-resolution index space
            return val, D, n

        lr = None
        values = []
        best = (-np.inf, tf.copy_state())
        prev_val, prev_state = None, None
        for it in range(int(n_it)):
            val, D, n = evaluate()
            if n == 0:
                raise ValueError("Failed to estimate registration transform: the volumes do not overlap.")
            if prev_val is not None and val < prev_val:  # overshoot: go back, halve the step
                tf.restore(prev_state)
                lr *= 0.5
                if lr < 1e-9:
                    break
                val, D, n = evaluate()
            values.append(val)
            if val > best[0]:
                best = (val, tf.copy_state())
            if _converged(values):
                break
            gens = tf.generators()
            DA = D[:, :3] - np.outer(D[:, 3], tf.c)  # derivative w.r.t. A at fixed centre (b = c + t - A c)
            grad = np.array([(DA * g[0]).sum() + (D[:, 3] * g[1]).sum() for g in gens])
            rel = corners - tf.c
            shifts = np.array([[np.linalg.norm(g[0] @ r + g[1]) for r in rel] for g in gens])
            scales = np.maximum((shifts**2).max(axis=1), 1e-12)
            delta = grad / scales
            step_shift = max(np.linalg.norm(sum(d * (g[0] @ r + g[1]) for d, g in zip(delta, gens))) for r in rel)
            if step_shift <= 0:
                break
            if lr is None:
                lr = max_step / step_shift  # estimated once per level
            step = min(lr, max_step / step_shift) * delta  # never move a corner by more than max_step
            prev_val, prev_state = val, tf.copy_state()
            tf.apply_update(gens, step)
            if verbose and it % 50 == 0:
                print(f"level {level} (shrink {f}, sigma {sg}) it {it}: MI {val:.6f}, samples {n}, lr {lr:.3g}")
        tf.restore(best[1])
        if verbose:
            print(f"level {level} done after {len(values)} evaluations: MI {best[0]:.6f}")

    T = tf.matrix()
    if not np.all(np.isfinite(T)):
        raise ValueError("Failed to estimate registration transform.")
    return T


def _as_f32_volume(a, dev, planar) -> torch.Tensor:
    t, _, _ = as_device_volume(a, dev)
    t = t.to(torch.float32)
    if planar:
        t = t[None]
    return t.contiguous()


def preprocess_czyx(
    mov_czyx: np.ndarray,
    ref_czyx: np.ndarray,
    initial_tform: Transform,
    mov_channel_index: int | list = 0,
    ref_channel_index: int = 0,
    crop: bool = False,
    ref_mask_radius: float | None = None,
    clip: bool = False,
    sobel_filter: bool = False,
    verbose: bool = False,
    device="cuda",
) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """``registration/ants.py:125-278``: initial warp of the moving channels onto the reference grid, optional crop to
    the largest interior rectangle of the overlap, hard-coded clipping, Sobel filter, channel sum."""
    mov_czyx = np.asarray(mov_czyx).astype(np.float32)
    ref_czyx = np.asarray(ref_czyx).astype(np.float32)
    if ref_mask_radius is not None and not (0 < ref_mask_radius <= 1):
        raise ValueError("ref_mask_radius must be given as a fraction of image width, i.e. (0, 1].")
    if _check_nan_n_zeros(mov_czyx) or _check_nan_n_zeros(ref_czyx):
        raise ValueError("Input data contains NaN or zeros.")
    ref_zyx = ref_czyx[ref_channel_index]
    if ref_zyx.ndim != 3:
        raise ValueError(f"Expected 3D reference channel, got shape {ref_zyx.shape}")
    if not isinstance(mov_channel_index, list):
        mov_channel_index = [mov_channel_index]
    dev = resolve_device(device)
    mov_channels = []
    for idx in mov_channel_index:
        ch = np.asarray(mov_czyx[idx]).astype(np.float32)
        if ch.ndim != 3:
            raise ValueError(f"Expected 3D moving channel, got shape {ch.shape}")
        mov_channels.append(affine_device(ch, initial_tform.matrix, ref_zyx.shape, "linear", device=dev))

    offset = np.zeros(3, dtype=np.float32)
    ref_t = torch.from_numpy(np.ascontiguousarray(ref_zyx)).to(dev)
    if crop:
        mask = ((ref_t != 0) & (mov_channels[0] != 0)).cpu().numpy()
        if ref_mask_radius is not None:
            ref_mask = np.zeros(ref_zyx.shape[-2:], dtype=bool)
            y, x = np.ogrid[: ref_mask.shape[-2], : ref_mask.shape[-1]]
            center = (ref_mask.shape[-2] // 2, ref_mask.shape[-1] // 2)
            radius = int(ref_mask_radius * min(center))
            ref_mask[(x - center[0]) ** 2 + (y - center[1]) ** 2 <= radius**2] = True  # (sic) as the reference :230
            mask = mask * ref_mask
        z_slice, y_slice, x_slice = find_lir(mask.astype(np.uint8))
        if verbose:
            print(f"Cropping to region z={z_slice.start}:{z_slice.stop}, y={y_slice.start}:{y_slice.stop}, "
                  f"x={x_slice.start}:{x_slice.stop}")
        offset = np.asarray([s.start for s in (z_slice, y_slice, x_slice)], dtype=np.float32)
        ref_t = ref_t[z_slice, y_slice, x_slice].contiguous()
        mov_channels = [c[z_slice, y_slice, x_slice].contiguous() for c in mov_channels]
    if clip:  # hard-coded limits of the reference (:263-269); plain elementwise glue, not a hot path
        ref_t = ref_t.clamp(0, 0.5)
        mov_channels = [c.clamp(110, float(np.quantile(c.cpu().numpy(), 0.99))) for c in mov_channels]
    if sobel_filter:
        ref_t = _metric.sobel(ref_t.contiguous())
        mov_channels = [_metric.sobel(c.contiguous()) for c in mov_channels]
    mov_t = mov_channels[0] if len(mov_channels) == 1 else torch.stack(mov_channels).sum(dim=0)
    return ref_t.cpu().numpy(), mov_t.cpu().numpy(), offset


def postprocess_transform(initial_transform: Transform, fwd_transform: Transform, preprocess_offset) -> Transform:
    """``registration/ants.py:369-404``: composed = initial @ shift_to_roi @ fwd @ shift_back."""
    shift_to_roi = np.eye(4)
    shift_to_roi[:3, -1] = preprocess_offset
    shift_back = np.eye(4)
    shift_back[:3, -1] = -np.asarray(preprocess_offset)
    return Transform(initial_transform.matrix @ shift_to_roi @ fwd_transform.matrix @ shift_back)


def estimate_czyx(
    mov_czyx: np.ndarray,
    ref_czyx: np.ndarray,
    initial_tform: np.ndarray,
    mov_channel_index: int | list = 0,
    ref_channel_index: int = 0,
    crop: bool = False,
    ref_mask_radius: float | None = None,
    clip: bool = False,
    sobel_filter: bool = False,
    verbose: bool = False,
    t_idx: int = 0,
    output_folder_path: str | None = None,
    device="cuda",
) -> Transform:
    """``registration/ants.py:281-366``: preprocess, estimate on the preprocessed pair, compose with the initial guess."""
    initial = Transform(matrix=initial_tform)
    ref_zyx, mov_zyx, offset = preprocess_czyx(
        mov_czyx=mov_czyx, ref_czyx=ref_czyx, initial_tform=initial, mov_channel_index=mov_channel_index,
        ref_channel_index=ref_channel_index, crop=crop, clip=clip, ref_mask_radius=ref_mask_radius,
        sobel_filter=sobel_filter, verbose=verbose, device=device)
    fwd, _inv = estimate(ref=ref_zyx, mov=mov_zyx, verbose=verbose, device=device)
    composed = postprocess_transform(initial, fwd, offset)
    if output_folder_path:
        output_folder_path = Path(output_folder_path)
        output_folder_path.mkdir(parents=True, exist_ok=True)
        np.save(output_folder_path / f"{t_idx}.npy", composed.matrix)
    return composed


def estimate_tczyx(
    mov_tczyx,
    ref_tczyx,
    mov_channel_index: int | list[int],
    ref_channel_index: int,
    ants_registration_settings,
    affine_transform_settings,
    verbose: bool = False,
    output_folder_path: Path = None,
    cluster: str = "local",
    sbatch_filepath: Path = None,
    device="cuda",
) -> list[np.ndarray]:
    """One transform per timepoint (``registration/ants.py:407-532``).

    The reference submits one ``estimate_czyx`` job per timepoint through submitit and reads the saved
    ``xyz_transforms/{t}.npy`` back; here the timepoints of this rank run in-process on its GPU (timepoints are dealt
    round-robin over ranks under ``torchrun``, like positions in the other commands) and the same files are written.
    """
    from .. import parallel

    T = mov_tczyx.shape[0]
    initial = np.asarray(affine_transform_settings.approx_transform, dtype=np.float64)
    output_folder_path = Path(output_folder_path)
    out = output_folder_path / "xyz_transforms"
    out.mkdir(parents=True, exist_ok=True)
    rank, world = parallel.init()  # binds this rank to GPU LOCAL_RANK before any device call
    for t in range(rank, T, world):
        estimate_czyx(
            mov_czyx=np.asarray(mov_tczyx[t]), ref_czyx=np.asarray(ref_tczyx[t]), initial_tform=initial,
            mov_channel_index=mov_channel_index, ref_channel_index=ref_channel_index,
            crop=getattr(ants_registration_settings, "crop", False),
            ref_mask_radius=getattr(ants_registration_settings, "ref_mask_radius", None),
            clip=getattr(ants_registration_settings, "clip", False),
            sobel_filter=ants_registration_settings.sobel_filter, verbose=verbose, t_idx=t, output_folder_path=out,
            device=device)
    parallel.barrier()
    transforms = [np.load(out / f"{t}.npy").tolist() for t in range(T) if (out / f"{t}.npy").exists()]
    if len(transforms) != T:
        raise ValueError(f"Number of transforms {len(transforms)} does not match number of timepoints {T}")
    return transforms
