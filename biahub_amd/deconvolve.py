"""FFT deconvolution on MI355X — host-side mirror of ``biahub/deconvolve.py``.

``compute_tranfser_function`` (the reference's spelling, deconvolve.py:30) and ``deconvolve``
(:46) keep their signatures; ``richardson_lucy`` / ``richardson_lucy_czyx`` are the north-star
extension (no reference counterpart).  FFTs are hipFFT R2C/C2R plans cached in the context;
the pointwise steps are the fused kernels of ``csrc/deconv.hip``.
"""

from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .device import as_device_volume, get_context, ptr, resolve_device, to_host
from .device import empty as device_empty, empty_like as device_empty_like


def _f32_device(x, device=None):
    t, code, dev = as_device_volume(x, device)
    if code != _lib.DT_F32:
        t = t.to(torch.float32)
    return t, dev


def transfer_function_device(psf_zyx, output_zyx_shape, device="cuda") -> torch.Tensor:
    """|FFT(zero-padded PSF)| / max as a full-spectrum float32 device tensor (deconvolve.py:30-43)."""
    psf, dev = _f32_device(psf_zyx, device)
    if psf.ndim != 3 or len(output_zyx_shape) != 3:
        raise ValueError("psf and output shape must be 3-D")
    Z, Y, X = (int(s) for s in output_zyx_shape)
    pz, py, px = (int(s) for s in psf.shape)
    if pz > Z or py > Y or px > X:
        # np.pad in the reference raises on negative pad widths
        raise ValueError("index can't contain negative values")
    ctx = get_context(dev)
    with torch.cuda.device(dev):
        tf = device_empty((Z, Y, X), torch.float32, dev)
        _lib.check(ctx.lib.bh_transfer_function(ctx.handle, ptr(psf), pz, py, px, Z, Y, X, ptr(tf)))
    return tf


def compute_tranfser_function(psf_zyx_data: np.ndarray, output_zyx_shape: tuple, device="cuda") -> np.ndarray:
    """numpy in / numpy out transfer function (biahub/deconvolve.py:30-43)."""
    return to_host(transfer_function_device(psf_zyx_data, output_zyx_shape, device))


def tikhonov_zyx(zyx, transfer_function, regularization_strength: float = 1e-3) -> torch.Tensor:
    """Zero-phase Tikhonov inverse filter of one volume on device.

    ``real(ifftn(fftn(x) * conj(H) / (|H|^2 + reg)))`` — what waveorder's
    ``apply_inverse_transfer_function(x, H, z_padding=0, regularization_strength)`` computes at the
    reference's call site (biahub/deconvolve.py:58-63); H is real and even so R2C/C2R is exact.
    """
    x, dev = _f32_device(zyx)
    H, _ = _f32_device(transfer_function, dev)
    if tuple(H.shape) != tuple(x.shape):
        raise ValueError(f"transfer function shape {tuple(H.shape)} != data shape {tuple(x.shape)}")
    Z, Y, X = (int(s) for s in x.shape)
    ctx = get_context(dev)
    with torch.cuda.device(dev):
        out = device_empty_like(x)
        _lib.check(ctx.lib.bh_tikhonov(ctx.handle, ptr(x), ptr(H), Z, Y, X, float(regularization_strength),
                                       ptr(out)))
    return out


# Prepared inverse filters of the transfer functions `deconvolve` was last called with.  The reference's orchestrator hands
# every (t, c) unit of a plate the same transfer function (biahub/deconvolve.py:183-191; each worker re-reads it from zarr,
# :52-54); here it is uploaded and staged once (bh_inverse_filter_create) and every later call only runs the five passes.
# Keyed by the identity of the object that was passed (kept alive by the entry), its shape, the regularisation and the device.
# The in-place-edit guard samples ~4096 strided elements: editing a transfer function in place between calls is NOT supported
# (pass a new array); the sample only catches the common whole-array rewrites.  The oldest entry is evicted first, so two
# transfer functions used in alternation both stay staged.
_PREPARED: "dict[tuple, tuple]" = {}
_PREPARED_MAX = 2


def _prepared_filter(transfer_function, regularization_strength: float, dev):
    from .apply_inverse_transfer_function import PreparedInverseFilter

    shape = tuple(int(s) for s in transfer_function.shape)
    flat = transfer_function.reshape(-1)
    probe = flat[:: max(1, flat.shape[0] // 4096)]                       # in-place edits of the same object must miss
    probe = probe.cpu().numpy() if isinstance(probe, torch.Tensor) else np.asarray(probe)
    key = (id(transfer_function), shape, float(regularization_strength), str(dev), hash(probe.tobytes()))
    hit = _PREPARED.get(key)
    if hit is not None and hit[0] is transfer_function:
        return hit[1]
    while len(_PREPARED) >= _PREPARED_MAX:
        old = _PREPARED.pop(next(iter(_PREPARED)))  # dicts keep insertion order: the oldest entry goes
        old[1].close()
    H, _ = _f32_device(transfer_function, dev)
    prep = PreparedInverseFilter(H, shape, 0, regularization_strength, "f32", dev)
    _PREPARED[key] = (transfer_function, prep)
    return prep


def deconvolve(
    czyx_raw_data: np.ndarray,
    transfer_function=None,
    transfer_function_store_path: str = None,
    regularization_strength: float = 1e-3,
    device="cuda",
) -> np.ndarray:
    """Per-channel Tikhonov deconvolution, numpy CZYX in / out (biahub/deconvolve.py:46-66).

    ``transfer_function`` may be a numpy array or a tensor; it is uploaded and staged as an inverse filter ONCE and kept
    for the following calls with the same object (the reference re-reads it from zarr in every worker, :52-54).
    ``transfer_function_store_path`` is read through ``biahub_amd.io`` when given.
    """
    dev = resolve_device(device)
    if transfer_function is None:
        if transfer_function_store_path is None:
            raise ValueError("either transfer_function or transfer_function_store_path is required")
        from .io import read_fov_array  # local: optional IO helper

        transfer_function = read_fov_array(transfer_function_store_path)[0, 0]
    czyx = np.asarray(czyx_raw_data)
    if tuple(transfer_function.shape) != tuple(czyx.shape[-3:]):
        raise ValueError(f"transfer function shape {tuple(transfer_function.shape)} != data shape {tuple(czyx.shape[-3:])}")
    prep = _prepared_filter(transfer_function, regularization_strength, dev)
    return np.stack([to_host(prep(_f32_device(zyx, dev)[0], False)) for zyx in czyx])


def richardson_lucy(zyx, psf_zyx, iterations: int = 10, eps: float = 1e-6) -> torch.Tensor:
    """Richardson-Lucy deconvolution of one volume on device (north-star extension, C3).

    Definition: h = psf/sum(psf) centred at the origin, circular boundary;
    e0 = max(d,0); e <- max(e * corr_h(d / max(conv_h(e), eps)), 0), ``iterations`` times.
    """
    d, dev = _f32_device(zyx)
    psf, _ = _f32_device(psf_zyx, dev)
    if d.ndim != 3 or psf.ndim != 3:
        raise ValueError("volume and psf must be 3-D")
    Z, Y, X = (int(s) for s in d.shape)
    pz, py, px = (int(s) for s in psf.shape)
    ctx = get_context(dev)
    with torch.cuda.device(dev):
        out = device_empty_like(d)
        _lib.check(ctx.lib.bh_richardson_lucy(ctx.handle, ptr(d), ptr(psf), pz, py, px, Z, Y, X, int(iterations),
                                              float(eps), ptr(out)))
    return out


class PreparedRichardsonLucy:
    """Richardson-Lucy with the transfer function of one PSF built once for one volume shape (``bh_richardson_lucy_create``)
    and applied to any number of volumes — the shape of the reference's deconvolve, which computes the transfer function
    once per plate and hands it to every (position, t, c) unit (biahub/deconvolve.py:140-149, 183-191).  A call only
    enqueues kernels on the current stream: no read-back, no host synchronisation (the one-shot ``richardson_lucy`` validates
    its cached transfer function against the PSF's bytes on every call), so an upload / compute / download pipeline overlaps.
    """

    def __init__(self, psf_zyx, zyx_shape, device="cuda"):
        import ctypes

        self.device = resolve_device(device)
        psf, _ = _f32_device(psf_zyx, self.device)
        if psf.ndim != 3 or len(zyx_shape) != 3:
            raise ValueError("volume and psf must be 3-D")
        self.shape = tuple(int(n) for n in zyx_shape)
        self._ctx = get_context(self.device)
        self._handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._ctx.lib.bh_richardson_lucy_create(self._ctx.handle, ptr(psf), *(int(k) for k in psf.shape),
                                                               *self.shape, ctypes.byref(self._handle)))
            torch.cuda.current_stream(self.device).synchronize()  # the PSF may be dropped by the caller from here on
        box = (ctypes.c_int64 * 3)()
        backend, is_real, nbytes = ctypes.c_int(), ctypes.c_int(), ctypes.c_uint64()
        _lib.check(self._ctx.lib.bh_richardson_lucy_info(self._handle, box, ctypes.byref(backend), ctypes.byref(is_real),
                                                         ctypes.byref(nbytes)))
        self.box = tuple(int(b) for b in box)
        self.backend = ("engine", "engine-padded", "library")[backend.value]
        self.otf_is_real = bool(is_real.value)
        self.otf_bytes = int(nbytes.value)

    def __call__(self, zyx, iterations: int = 10, eps: float = 1e-6, out: torch.Tensor | None = None,
                 row_sums: torch.Tensor | None = None):
        """The deconvolved volume.  With ``row_sums`` (a float64 ``(Z, Y)`` device tensor) the call returns
        ``(volume, row_sums or None)``: the last update pass also reduces ``sum over x`` of every row it stores when the back-end
        can (``None`` comes back when it cannot) — hand them to ``fast_deskew_zyx(..., row_sums=...)`` and its overhang fill
        needs no pass of its own over the volume."""
        import ctypes

        d, _ = _f32_device(zyx, self.device)
        if tuple(d.shape) != self.shape:
            raise ValueError(f"volume shape {tuple(d.shape)} != the shape this handle was prepared for {self.shape}")
        if row_sums is not None and (row_sums.dtype != torch.float64 or tuple(row_sums.shape) != self.shape[:2]
                                     or not row_sums.is_contiguous() or row_sums.device != d.device):
            raise ValueError("row_sums must be a contiguous float64 (Z, Y) tensor on the volume's device")
        ctx = get_context(self.device)
        with torch.cuda.device(self.device):
            if out is None:
                out = device_empty_like(d)
            if row_sums is None:
                _lib.check(ctx.lib.bh_richardson_lucy_apply(ctx.handle, self._handle, ptr(d), int(iterations), float(eps), ptr(out)))
                return out
            produced = ctypes.c_int(0)
            _lib.check(ctx.lib.bh_richardson_lucy_apply_rows(ctx.handle, self._handle, ptr(d), int(iterations), float(eps), ptr(out),
                                                             ptr(row_sums), ctypes.byref(produced)))
        return out, (row_sums if produced.value else None)

    def close(self) -> None:
        if self._handle:
            torch.cuda.synchronize(self.device)
            _lib.check(self._ctx.lib.bh_richardson_lucy_destroy(self._handle))
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


def richardson_lucy_plan(psf_shape_zyx, volume_shape_zyx) -> tuple[tuple[int, int, int], str]:
    """(transform box, back-end) ``richardson_lucy`` uses for a shape — "engine" (fused FFT engine at the volume's shape),
    "engine-padded" (the engine at a larger wrap-padded box) or "library" (hipFFT, 7-smooth pad-and-fold box if needed).
    Host logic only (``bh_richardson_lucy_plan``): no GPU involved."""
    import ctypes

    lib = _lib.load()
    box = (ctypes.c_int64 * 3)()
    backend = ctypes.c_int()
    _lib.check(lib.bh_richardson_lucy_plan(*(int(k) for k in psf_shape_zyx), *(int(n) for n in volume_shape_zyx), box,
                                           ctypes.byref(backend)))
    return tuple(int(b) for b in box), ("engine", "engine-padded", "library")[backend.value]


def richardson_lucy_czyx(czyx_raw_data: np.ndarray, psf_zyx: np.ndarray, iterations: int = 10, eps: float = 1e-6,
                         device="cuda") -> np.ndarray:
    """CZYX numpy adapter with the reference's operator signature ``func(czyx, **kwargs)``."""
    dev = resolve_device(device)
    czyx = np.asarray(czyx_raw_data)
    prep = _prepared_rl(psf_zyx, tuple(czyx.shape[-3:]), dev)
    return np.stack([to_host(prep(_f32_device(zyx, dev)[0], iterations, eps)) for zyx in czyx])


def richardson_lucy_czyx_device(czyx_raw_data, psf_zyx, iterations: int = 10, eps: float = 1e-6, device="cuda") -> torch.Tensor:
    """``richardson_lucy_czyx`` with both ends in HBM: takes a numpy array or a ``(C, Z, Y, X)`` device tensor (what
    ``io.process_single_position`` hands operators marked ``device_input``: the store's device read path) and returns a float32
    device tensor (``device_resident``: the output store permutes and, for lz4 stores, compresses it on the GPU)."""
    dev = resolve_device(device)
    prep = _prepared_rl(psf_zyx, tuple(int(n) for n in czyx_raw_data.shape[-3:]), dev)
    return torch.stack([prep(_f32_device(zyx, dev)[0], iterations, eps) for zyx in czyx_raw_data])


richardson_lucy_czyx_device.device_resident = True
richardson_lucy_czyx_device.device_input = True


# The prepared handle of the PSF `richardson_lucy_czyx` was last called with (a plate job calls it once per (t, c) unit with
# the same PSF array): keyed like _PREPARED by the identity of the object, its bytes' hash, the volume shape and the device.
_PREPARED_RL: "dict[tuple, tuple]" = {}


def _prepared_rl(psf_zyx, shape, dev) -> PreparedRichardsonLucy:
    # 1. the same object as last time: no copy to the host, no hash (a device tensor would be a synchronisation per call; a host
    #    array's few kilobytes are hashed anyway, which also catches an array that was modified in place)
    on_device = isinstance(psf_zyx, torch.Tensor) and psf_zyx.is_cuda
    ident = ("id", id(psf_zyx), shape, str(dev))
    hit = _PREPARED_RL.get(ident)
    host = None
    if hit is not None and hit[0] is psf_zyx:
        if on_device:
            return hit[1]
        host = psf_zyx.numpy() if isinstance(psf_zyx, torch.Tensor) else np.asarray(psf_zyx)
        if hit[2] == hash(host.tobytes()):
            return hit[1]
    # 2. another object with the same contents (a PSF re-read per unit): found by its bytes
    if host is None:
        host = psf_zyx.cpu().numpy() if isinstance(psf_zyx, torch.Tensor) else np.asarray(psf_zyx)
    digest = hash(host.tobytes())
    content = ("bytes", digest, tuple(host.shape), str(host.dtype), shape, str(dev))
    hit = _PREPARED_RL.get(content)
    if hit is not None:
        _PREPARED_RL[ident] = (psf_zyx, hit[1], digest)
        return hit[1]
    while len(_PREPARED_RL) >= 2 * _PREPARED_MAX:
        old = _PREPARED_RL.pop(next(iter(_PREPARED_RL)))
        if not any(v[1] is old[1] for v in _PREPARED_RL.values()):
            old[1].close()
    prep = PreparedRichardsonLucy(psf_zyx, shape, dev)
    _PREPARED_RL[ident] = (psf_zyx, prep, digest)
    _PREPARED_RL[content] = (None, prep, digest)
    return prep
