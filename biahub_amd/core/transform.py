"""Immutable homogeneous transform — mirror of ``biahub/core/transform.py`` for the 3-D hot path.

``Transform.apply`` is the SciPy-semantics resample (``core/transform.py:374-396``): the stored
matrix is a PUSH transform, the kernel pulls with its inverse, boundary mode "constant" (no
interpolation past the edge, ``cval`` outside).  It runs in ``csrc/affine.hip`` with
``BH_BOUNDARY_SCIPY_CONSTANT``; orders 0 and 1 are supported.
"""

from __future__ import annotations

import numpy as np

from .. import _lib


class Transform:
    """Homogeneous 3-D (4x4) or 2-D (3x3) transform; operations return new instances."""

    __slots__ = ("_matrix", "_ndim", "_transform_type")

    def __init__(self, matrix, transform_type: str = "affine"):
        m = np.array(matrix, dtype=np.float64)
        if m.ndim != 2 or m.shape[0] != m.shape[1] or m.shape[0] not in (3, 4):
            raise ValueError(f"matrix must be 3x3 (2D) or 4x4 (3D), got shape {m.shape}")
        m.setflags(write=False)
        self._matrix = m
        self._ndim = m.shape[0] - 1
        self._transform_type = transform_type

    # -- properties (core/transform.py:78-108) ------------------------------------------
    @property
    def matrix(self):
        return self._matrix.copy()

    @property
    def ndim(self) -> int:
        return self._ndim

    @property
    def transform_type(self) -> str:
        return self._transform_type

    @property
    def translation(self):
        return self._matrix[:-1, -1].copy()

    @property
    def linear(self):
        return self._matrix[:-1, :-1].copy()

    @property
    def is_identity(self) -> bool:
        return bool(np.allclose(self._matrix, np.eye(self._ndim + 1)))

    # -- constructors (core/transform.py:110-166) ----------------------------------------
    @classmethod
    def identity(cls, ndim: int = 3) -> "Transform":
        return cls(np.eye(ndim + 1), "identity")

    @classmethod
    def from_translation(cls, offset) -> "Transform":
        offset = np.asarray(offset, dtype=np.float64)
        m = np.eye(len(offset) + 1)
        m[:-1, -1] = offset
        return cls(m, "translation")

    @classmethod
    def from_skimage(cls, skimage_transform, ndim: int = 3) -> "Transform":
        """From a scikit-image geometric transform (anything with a ``.params`` homogeneous matrix), core/transform.py:
        169-227: the type comes from the class name; a 2-D transform asked for in 3-D acts in the YX plane, Z untouched."""
        params = np.asarray(skimage_transform.params)
        name = type(skimage_transform).__name__.lower()
        kind = next((k for k in ("euclidean", "similarity", "affine") if k in name), "affine")
        if params.shape == (3, 3):
            if ndim == 2:
                return cls(params, transform_type=kind)
            if ndim == 3:
                m = np.eye(4, dtype=np.float64)
                m[1:3, 1:3] = params[:2, :2]
                m[1:3, 3] = params[:2, 2]
                return cls(m, transform_type=kind)
            return None  # the reference falls through for other ndim
        if params.shape == (4, 4):
            if ndim == 3:
                return cls(params, transform_type=kind)
            raise ValueError("Cannot convert 3D skimage transform to 2D")
        raise ValueError(f"Unexpected skimage transform shape: {params.shape}")

    # -- algebra (core/transform.py:231-298) ----------------------------------------------
    def invert(self) -> "Transform":
        return Transform(np.linalg.inv(self._matrix), self._transform_type)

    def compose(self, other: "Transform") -> "Transform":
        if not isinstance(other, Transform):
            raise TypeError(f"Cannot compose Transform with {type(other)}")
        if other._ndim != self._ndim:
            raise ValueError(f"Cannot compose {self._ndim}D transform with {other._ndim}D transform")
        return Transform(self._matrix @ other._matrix, "affine")

    def __matmul__(self, other):
        return self.compose(other)

    def apply_points(self, points):
        points = np.asarray(points, dtype=np.float64)
        if points.ndim != 2:
            raise ValueError(f"points must be 2D array (N, D), got shape {points.shape}")
        if points.shape[1] != self._ndim:
            raise ValueError(f"points must have {self._ndim} columns, got {points.shape[1]}")
        h = np.hstack([points, np.ones((points.shape[0], 1))])
        return (self._matrix @ h.T).T[:, :-1]

    # -- resampling (core/transform.py:321-396) -------------------------------------------
    def apply(self, moving, reference=None, order: int = 1, mode: str = "constant", cval: float = 0.0,
              backend: str = "scipy", device="cuda"):
        """Resample ``moving`` into the reference grid on the GPU.

        ``backend="scipy"`` -> SciPy "constant" boundary; ``backend="ants"`` -> ITK boundary rule.
        Result dtype follows the reference: the input dtype for SciPy, float32 for ANTs.
        """
        from ..register import affine_device, cast_like_scipy

        moving = np.asarray(moving)
        if moving.ndim != self._ndim:
            raise ValueError(f"Expected {self._ndim}D array, got {moving.ndim}D")
        if backend not in ("scipy", "ants"):
            raise ValueError(f"Unknown backend: {backend}")
        if backend == "scipy" and mode != "constant":
            raise NotImplementedError(f"boundary mode {mode!r} is not implemented on the GPU path")
        if order not in (0, 1, 3) or (order == 3 and backend != "scipy"):
            raise NotImplementedError("interpolation orders 0, 1 and 3 (3: SciPy backend) are implemented on the GPU path")
        if self._ndim == 2 and backend != "scipy":
            raise NotImplementedError("2-D images are resampled by the SciPy backend only")
        out_shape = tuple(reference.shape) if reference is not None else tuple(moving.shape)
        inv = np.linalg.inv(self._matrix)
        vol = moving
        if self._ndim == 2:  # a 2-D image is the single plane of a (1, Y, X) volume: z is the identity
            vol = moving[None]
            inv3 = np.eye(4)
            inv3[1:, 1:] = inv
            inv, out_shape = inv3, (1,) + out_shape
        boundary = _lib.BOUNDARY_SCIPY_CONSTANT if backend == "scipy" else _lib.BOUNDARY_ITK
        interp = {0: "nearestneighbor", 1: "linear", 3: "cubic"}[order]
        out = affine_device(vol, inv, out_shape, interp, boundary, float(cval) if backend == "scipy" else 0.0, device=device)
        if backend == "scipy":  # SciPy writes into an array of the input's dtype: integers round and saturate
            out = cast_like_scipy(out, moving.dtype)
        out = out.cpu().numpy()
        return out[0] if self._ndim == 2 else out

    # -- ANTs parameter packing (core/transform.py:427-495) -------------------------------
    def to_ants(self):
        """The 12 (3-D) ITK AffineTransform parameters ``[A row-major ; t]`` (centre 0)."""
        n = self._ndim
        return np.concatenate([self._matrix[:n, :n].ravel(), self._matrix[:n, n]])

    @classmethod
    def from_ants(cls, parameters, fixed_parameters=None) -> "Transform":
        p = np.asarray(parameters, dtype=np.float64)
        n = 3 if p.size == 12 else 2
        m = np.eye(n + 1)
        m[:n, :n] = p[: n * n].reshape(n, n)
        fixed = np.zeros(n) if fixed_parameters is None else np.asarray(fixed_parameters, dtype=np.float64)
        m[:n, n] = p[n * n :] + (np.eye(n) - m[:n, :n]) @ fixed
        return cls(m)

    # -- (de)serialisation (core/transform.py:499-530) ------------------------------------
    def to_list(self):
        return self._matrix.tolist()

    @classmethod
    def from_list(cls, data, transform_type: str = "affine") -> "Transform":
        return cls(np.array(data), transform_type)

    def to_dict(self) -> dict:
        return {"matrix": self.to_list(), "transform_type": self._transform_type, "ndim": self._ndim}

    @classmethod
    def from_dict(cls, data: dict) -> "Transform":
        return cls(np.array(data["matrix"]), data.get("transform_type", "affine"))

    def __repr__(self) -> str:
        return f"Transform(ndim={self._ndim}, type='{self._transform_type}', translation={self.translation.round(3).tolist()})"

    def __str__(self) -> str:
        return f"Transform({self._transform_type}, {self._ndim}D)\n" + np.array2string(self._matrix, precision=4, suppress_small=True)

    def __eq__(self, other) -> bool:
        return isinstance(other, Transform) and np.allclose(self._matrix, other._matrix)

    def __hash__(self) -> int:
        return hash(self._matrix.tobytes())
