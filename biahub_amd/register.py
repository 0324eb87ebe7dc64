"""3-D affine register / crop on MI355X — host-side mirror of ``biahub/register.py``.

The reference resamples with ANTs/ITK (``ANTsTransform.apply_to_image``, register.py:261-269);
here the same pull-resample ``out(p) = in(A p + t)`` (ZYX index space, origin 0, spacing 1) runs
in ``csrc/affine.hip``.  ITK's boundary rule (inside iff -0.5 <= c < N-0.5, edge clamp) is the
default; see DESIGN.md §3 for what is pinned.
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .device import as_device_volume, get_context, ptr, resolve_device, to_host
from .device import empty as device_empty, empty_like as device_empty_like

# "cubic": SciPy's order-3 B-spline (prefilter + 64 taps) — only with BOUNDARY_SCIPY_CONSTANT; the two ANTs names as before
_INTERP = {"linear": _lib.INTERP_LINEAR, "nearestneighbor": _lib.INTERP_NEAREST, "cubic": _lib.INTERP_CUBIC}


def get_3D_rescaling_matrix(start_shape_zyx, scaling_factor_zyx=(1, 1, 1), end_shape_zyx=None):
    """Centre-preserving YX scaling (biahub/register.py:32-57)."""
    cy0, cx0 = np.array(start_shape_zyx)[-2:] / 2
    cy1, cx1 = (cy0, cx0) if end_shape_zyx is None else np.array(end_shape_zyx)[-2:] / 2
    sz, sy, sx = scaling_factor_zyx[-3], scaling_factor_zyx[-2], scaling_factor_zyx[-1]
    return np.array([[sz, 0, 0, 0], [0, sy, 0, -cy0 * sy + cy1], [0, 0, sx, -cx0 * sx + cx1], [0, 0, 0, 1]])


def get_3D_rotation_matrix(start_shape_zyx: tuple, angle: float = 0.0, end_shape_zyx: tuple = None) -> np.ndarray:
    """Rotation about Z through the YX centre (biahub/register.py:60-111)."""
    cy0, cx0 = np.array(start_shape_zyx)[-2:] / 2
    cy1, cx1 = (cy0, cx0) if end_shape_zyx is None else np.array(end_shape_zyx)[-2:] / 2
    t = np.radians(angle)
    c, s = np.cos(t), np.sin(t)
    return np.array([[1, 0, 0, 0], [0, c, -s, -cy0 * c + s * cx0 + cy1], [0, s, c, -cy0 * s - cx0 * c + cx1],
                     [0, 0, 0, 1]])


def get_3D_fliplr_matrix(start_shape_zyx: tuple, end_shape_zyx: tuple = None) -> np.ndarray:
    """Left-right flip (biahub/register.py:114-145)."""
    cx0 = start_shape_zyx[-1] / 2
    cx1 = cx0 if end_shape_zyx is None else end_shape_zyx[-1] / 2
    return np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, -1, 2 * cx1], [0, 0, 0, 1]])


def rescale_voxel_size(affine_matrix, input_scale):
    """Row norms times scale (biahub/register.py:397-398)."""
    return np.linalg.norm(affine_matrix, axis=1) * input_scale


def convert_transform_to_ants(T_numpy: np.ndarray) -> np.ndarray:
    """4x4 -> the 12 ITK AffineTransform parameters ``[A row-major ; t]`` (register.py:148-168).

    Returns the parameter vector itself (there is no ANTs object here); centre is 0.
    """
    T = np.asarray(T_numpy, dtype=np.float64)
    assert T.shape == (4, 4)
    return np.concatenate([T[:3, :3].ravel(), T[:3, 3]])


def convert_transform_to_numpy(params: np.ndarray, fixed_parameters=(0.0, 0.0, 0.0)) -> np.ndarray:
    """Inverse of the above incl. the centre term ``t + (I - A) c`` (register.py:171-199)."""
    p = np.asarray(params, dtype=np.float64)
    T = np.eye(4)
    T[:3, :3] = p[:9].reshape(3, 3)
    T[:3, 3] = p[9:12] + (np.eye(3) - T[:3, :3]) @ np.asarray(fixed_parameters, dtype=np.float64)
    return T


def affine_device(vol, matrix, output_shape_zyx, interpolation="linear", boundary=_lib.BOUNDARY_ITK, cval=0.0,
                  crop_lo=(0, 0, 0), crop_shape=None, device=None) -> torch.Tensor:
    """Device-level warp: tensor/array in, float32 tensor out (the sub-box ``crop`` of the target grid)."""
    if interpolation not in _INTERP:
        raise ValueError(f"Unknown interpolation {interpolation!r}; expected one of {sorted(_INTERP)}")
    t, code, dev = as_device_volume(vol, device)
    if t.ndim != 3:
        raise ValueError(f"expected a 3-D volume, got shape {tuple(t.shape)}")
    M = np.asarray(matrix, dtype=np.float64)
    if M.shape != (4, 4):
        raise ValueError(f"matrix must be 4x4, got {M.shape}")
    shape = tuple(int(s) for s in (crop_shape if crop_shape is not None else output_shape_zyx))
    m12 = (C.c_double * 12)(*M[:3, :].ravel())
    lo = (C.c_int64 * 3)(*[int(v) for v in crop_lo])
    Zi, Yi, Xi = (int(s) for s in t.shape)
    ctx = get_context(dev)
    with torch.cuda.device(dev):
        out = device_empty(shape, torch.float32, dev)
        if out.numel():
            _lib.check(ctx.lib.bh_affine(ctx.handle, ptr(t), code, Zi, Yi, Xi, m12, _INTERP[interpolation],
                                         int(boundary), float(cval), ptr(out), shape[0], shape[1], shape[2], lo))
    return out


def cast_like_scipy(t: torch.Tensor, dtype) -> torch.Tensor:
    """SciPy's conversion of an interpolated value into the output dtype (ni_interpolation.c, CASE_INTERP_OUT_*): floating
    types are cast; integers round half away from zero (unsigned: negatives -> 0) and saturate at the type's range."""
    dtype = np.dtype(dtype)
    if not np.issubdtype(dtype, np.integer):
        return t if dtype == np.float32 else t.to(getattr(torch, dtype.name))
    info = np.iinfo(dtype)
    r = torch.where(t > 0, torch.floor(t + 0.5), torch.ceil(t - 0.5) if info.min < 0 else torch.zeros_like(t))
    r = r.clamp_(float(info.min), float(info.max))
    tdt = {"uint8": torch.uint8, "int8": torch.int8, "int16": torch.int16, "uint16": torch.uint16, "int32": torch.int32,
           "int64": torch.int64}.get(dtype.name)
    if tdt is None:
        raise TypeError(f"unsupported output dtype {dtype}")
    if tdt == torch.uint16:  # no float -> uint16 cast kernel in this torch build: through int32
        return r.to(torch.int32).to(torch.uint16)
    return r.to(tdt)


def largest_interior_rectangle(mask2d: np.ndarray):
    """(x, y, width, height) of the largest axis-aligned all-True rectangle of a 2-D boolean grid.

    Stands in for ``largestinteriorrectangle.lir`` (0.2.1, uv.lock:2327 — absent here) as the reference uses it
    (register.py:289-308): maximal-rectangle-under-histogram sweep, O(H*W); among equal areas the first found in
    row-major order of the bottom-right corner wins.
    """
    m = np.asarray(mask2d, dtype=bool)
    H, W = m.shape
    heights = np.zeros(W + 1, dtype=np.int64)
    best = (0, 0, 0, 0, 0)  # area, x, y, w, h
    for r in range(H):
        heights[:W] = np.where(m[r], heights[:W] + 1, 0)
        stack = []
        for c in range(W + 1):
            h = heights[c]
            start = c
            while stack and stack[-1][1] > h:
                s0, sh = stack.pop()
                area = sh * (c - s0)
                if area > best[0]:
                    best = (area, s0, r - sh + 1, c - s0, sh)
                start = s0
            if not stack or stack[-1][1] < h:
                stack.append((start, h))
    return best[1], best[2], best[3], best[4]


def find_lir(registered_zyx: np.ndarray, plot: bool = False) -> tuple:
    """Largest interior cuboid of a registered mask, the reference's heuristic (register.py:284-342): LIR of the
    mid-Z YX plane, then the Z extent common to six probe ZY / ZX slices through that rectangle."""
    reg = np.asarray(registered_zyx, dtype=bool)
    x, y, width, height = (int(v) for v in largest_interior_rectangle(reg[reg.shape[0] // 2]))
    x_start, x_stop, y_start, y_stop = x, x + width, y, y + height
    x_slice, y_slice = slice(x_start, x_stop), slice(y_start, y_stop)
    spans = []
    for _x in (x_start, x_start + (x_stop - x_start) // 2, x_stop - 1):
        _, z, _, depth = largest_interior_rectangle(reg[:, y_slice, _x])
        spans.append((int(z), int(z + depth)))
    for _y in (y_start, y_start + (y_stop - y_start) // 2, y_stop - 1):
        _, z, _, depth = largest_interior_rectangle(reg[:, _y, x_slice])
        spans.append((int(z), int(z + depth)))
    spans = np.asarray(spans)
    return slice(int(spans[:, 0].max()), int(spans[:, 1].min())), y_slice, x_slice


def find_overlapping_volume(input_zyx_shape: tuple, target_zyx_shape: tuple, transformation_matrix: np.ndarray,
                            method: str = "LIR", plot: bool = False, device="cuda") -> tuple:
    """ZYX slices of the cuboid covered by the warped source inside the target grid (register.py:345-394): an
    all-ones volume is warped on the GPU (ITK boundary rule, like ANTs) and the LIR heuristic runs on the mask."""
    if method != "LIR":
        raise ValueError(f"Unknown method {method}")
    dev = resolve_device(device)
    ones = torch.ones(tuple(int(s) for s in input_zyx_shape), dtype=torch.float32, device=dev)
    warped = affine_device(ones, transformation_matrix, tuple(int(s) for s in target_zyx_shape), "linear",
                           _lib.BOUNDARY_ITK, 0.0, device=dev)
    return find_lir((warped > 0).cpu().numpy(), plot=plot)


def _slice_bounds(sl: slice, n: int):
    start, stop, step = sl.indices(n)
    if step != 1:
        raise ValueError("crop slices must have step 1")
    return start, max(stop - start, 0)


def apply_affine_transform(
    zyx_data: np.ndarray,
    matrix: np.ndarray,
    output_shape_zyx: tuple,
    method="ants",
    interpolation: str = "linear",
    crop_output_slicing: bool = None,
    device="cuda",
) -> np.ndarray:
    """Apply a 4x4 ZYX affine, optionally crop (biahub/register.py:202-281).

    4-D input recurses per channel (:241-252); NaN -> 0 before resampling (:254, fused into the
    kernel load); ``method`` accepts "ants" (ITK boundary rule) or "scipy" — the reference's raw
    ``scipy.ndimage.affine_transform(zyx_data, matrix, output_shape_zyx)`` (:271-272): cubic B-spline, SciPy "constant"
    boundary, and ``output_shape_zyx`` lands in SciPy's ``offset`` slot, which a homogeneous matrix overrides, so the
    result keeps the INPUT's shape and dtype (reproduced, not fixed); anything else raises
    ``ValueError("Unknown method ...")`` (:275).  Only the cropped sub-box is computed (the reference warps the full grid
    and slices, :278-279).
    """
    if method not in ("ants", "scipy"):
        raise ValueError(f"Unknown method {method}")
    zyx_data = np.asarray(zyx_data)
    if method == "scipy" and zyx_data.ndim != 4:
        Zs, Ys, Xs = (int(s) for s in zyx_data.shape)
        lo, shape = (0, 0, 0), (Zs, Ys, Xs)
        if crop_output_slicing is not None:
            b = [_slice_bounds(s, n) for s, n in zip(crop_output_slicing, (Zs, Ys, Xs))]
            lo, shape = tuple(v[0] for v in b), tuple(v[1] for v in b)
        dev = resolve_device(device)
        out = affine_device(zyx_data, matrix, (Zs, Ys, Xs), "cubic", _lib.BOUNDARY_SCIPY_CONSTANT, 0.0, lo, shape, dev)
        return to_host(cast_like_scipy(out, zyx_data.dtype))
    Z, Y, X = (int(s) for s in output_shape_zyx)
    lo, shape = (0, 0, 0), (Z, Y, X)
    if crop_output_slicing is not None:
        b = [_slice_bounds(s, n) for s, n in zip(crop_output_slicing, (Z, Y, X))]
        lo, shape = tuple(v[0] for v in b), tuple(v[1] for v in b)
    if zyx_data.ndim == 4:
        out = np.zeros((zyx_data.shape[0],) + shape, dtype=np.float32)
        for c in range(zyx_data.shape[0]):
            out[c] = apply_affine_transform(zyx_data[c], matrix, output_shape_zyx, method=method,
                                            interpolation=interpolation, crop_output_slicing=crop_output_slicing,
                                            device=device)
        return out
    dev = resolve_device(device)
    return to_host(affine_device(zyx_data, matrix, (Z, Y, X), interpolation, _lib.BOUNDARY_ITK, 0.0, lo, shape, dev))
