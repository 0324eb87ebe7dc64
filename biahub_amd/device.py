"""Device plumbing: torch provides device memory and streams, libbhcore does the work.

One ``bh_ctx`` per (process, device); every call binds it to torch's *current* stream so
the kernels order correctly with torch copies issued around them.
"""

from __future__ import annotations

import contextlib
import ctypes as C
import os
import threading

import numpy as np
import torch

from . import _lib

_CTX: dict[int, "Context"] = {}

_NP_TO_DT = {
    np.dtype(np.uint8): _lib.DT_U8,
    np.dtype(np.uint16): _lib.DT_U16,
    np.dtype(np.float32): _lib.DT_F32,
    np.dtype(np.int16): _lib.DT_I16,
}
_TORCH_TO_DT = {
    torch.uint8: _lib.DT_U8,
    torch.uint16: _lib.DT_U16,
    torch.float32: _lib.DT_F32,
    torch.int16: _lib.DT_I16,
}


def resolve_device(device) -> torch.device:
    """Turn the reference's ``device`` argument ("cuda", "cuda:1", torch.device) into a GPU.

    A CPU device is refused: this package is the MI355X path and has no CPU fallback.
    """
    dev = torch.device(device) if not isinstance(device, torch.device) else device
    if dev.type != "cuda":
        raise RuntimeError(
            f"biahub_amd runs on AMD GPUs only (got device={device!r}); it has no CPU path. "
            "Use the reference biahub package for CPU execution."
        )
    if not torch.cuda.is_available():
        raise RuntimeError("biahub_amd: no GPU visible to torch (torch.cuda.is_available() is False)")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


# ---- volumes in HBM: where their pages go matters ------------------------------------------------------
# hipMalloc hands out physically contiguous gigabytes, and the strided streams of the transform passes (rows 8 KiB apart,
# planes 16 MiB apart, four of them per kernel) then collide in the HBM channel / bank hash: the same kernels run ~10 % faster on
# buffers whose 2-MiB physical chunks are mapped in a shuffled order (library default for its own workspace: csrc/context.hip
# dev_alloc).  ``volume_pool`` gives torch tensors the same layout: a torch MemPool whose blocks come from the library's
# allocator; tensors allocated inside the context manager live in it (and return to it, cached, when they are dropped).
_POOL = None
_POOL_ALLOCATOR = None
_POOL_TLS = threading.local()
_POOL_LOCK = threading.Lock()  # one pool per process: two threads entering volume_pool for the first time must not both build one


@contextlib.contextmanager
def volume_pool(device=None):
    """Context manager: torch allocations inside it come from the library's allocator (``bh_torch_alloc``).  Re-entrant (torch
    refuses to nest ``use_mem_pool`` on one pool: the inner uses are no-ops); a no-op when ``BH_VOLUME_POOL=0`` or when this
    torch build has no pluggable allocator."""
    global _POOL, _POOL_ALLOCATOR
    if os.environ.get("BH_VOLUME_POOL", "1") == "0" or getattr(_POOL_TLS, "depth", 0) > 0:
        yield
        return
    if _POOL is None:
        with _POOL_LOCK:
            if _POOL is None:
                try:
                    _lib.load()
                    path = str(os.environ.get("BHCORE_LIB", _lib.LIB_PATH))
                    _POOL_ALLOCATOR = torch.cuda.memory.CUDAPluggableAllocator(path, "bh_torch_alloc", "bh_torch_free")
                    _POOL = torch.cuda.MemPool(_POOL_ALLOCATOR.allocator())
                except (AttributeError, RuntimeError):  # torch without MemPool / pluggable allocators
                    _POOL = False
    if _POOL is False:
        yield
        return
    _POOL_TLS.depth = 1
    try:
        with torch.cuda.use_mem_pool(_POOL, device=device):
            yield
    finally:
        _POOL_TLS.depth = 0


def alloc_layout() -> dict:
    """How the library lays out gigabyte buffers right now (``bh_alloc_layout``): chunk size, shuffled or not, live blocks."""
    lib = _lib.load()
    kib, shuf, nblk, nbytes = C.c_int(), C.c_int(), C.c_uint64(), C.c_uint64()
    _lib.check(lib.bh_alloc_layout(C.byref(kib), C.byref(shuf), C.byref(nblk), C.byref(nbytes)))
    rr, rb = C.c_uint64(), C.c_uint64()
    _lib.check(lib.bh_alloc_retained(C.byref(rr), C.byref(rb)))
    # retained_*: address ranges of released blocks that are kept reserved (csrc/context.hip dev_free): address space only, no memory
    return {"chunk_kib": kib.value, "shuffled": bool(shuf.value), "live_blocks": int(nblk.value), "live_gb": nbytes.value / 1e9,
            "retained_va_ranges": int(rr.value), "retained_va_gb": rb.value / 1e9,
            "volume_pool": os.environ.get("BH_VOLUME_POOL", "1") != "0" and _POOL is not False}


def release_volume_pool() -> None:
    """Give the pool's cached blocks back to the driver (blocks that live tensors still use go when those tensors do): the
    pool object is dropped — torch releases a pool's blocks with its last reference — and a new one is made on the next use."""
    global _POOL
    if _POOL:
        _POOL = None
        torch.cuda.empty_cache()


def empty(shape, dtype, device) -> torch.Tensor:
    """``torch.empty`` on the GPU inside ``volume_pool``: where the operators allocate their results.  The pool's cached blocks
    serve only allocations made inside it, and torch does not release a private pool's blocks on its own out-of-memory retry:
    on OOM the pool and torch's cache are given back to the driver and the allocation is tried once more (``BH_VOLUME_POOL=0``
    trades the layout for one shared cache)."""
    shape = tuple(int(n) for n in shape)
    try:
        with volume_pool(device):
            return torch.empty(shape, dtype=dtype, device=device)
    except torch.cuda.OutOfMemoryError:
        release_volume_pool()
        torch.cuda.empty_cache()
        with volume_pool(device):
            return torch.empty(shape, dtype=dtype, device=device)


def empty_like(t: torch.Tensor) -> torch.Tensor:
    return empty(t.shape, t.dtype, t.device)


class Context:
    def __init__(self, index: int):
        self.index = index
        self.lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self.lib.bh_ctx_create(index, None, C.byref(h)))
        self.handle = h
        self._stream = 0
        self.timing = False

    def bind_stream(self):
        s = torch.cuda.current_stream(self.index).cuda_stream
        if s != self._stream:
            _lib.check(self.lib.bh_ctx_set_stream(self.handle, C.c_void_p(s)))
            self._stream = s

    def synchronize(self):
        _lib.check(self.lib.bh_ctx_synchronize(self.handle))

    def set_timing(self, on: bool):
        """Per-operator HIP-event timing (``elapsed_ms``).  While it is on, Richardson-Lucy reads its per-iteration events back on
        the host inside the call — turn it off for anything that overlaps streams."""
        _lib.check(self.lib.bh_ctx_set_timing(self.handle, int(on)))
        self.timing = bool(on)

    def elapsed_ms(self, what: int) -> float:
        ms = C.c_float()
        _lib.check(self.lib.bh_last_elapsed_ms(self.handle, what, C.byref(ms)))
        return float(ms.value)

    def workspace_bytes(self) -> int:
        b = C.c_uint64()
        _lib.check(self.lib.bh_ctx_workspace_bytes(self.handle, C.byref(b)))
        return int(b.value)

    def release_workspace(self):
        _lib.check(self.lib.bh_ctx_release_workspace(self.handle))
        _lib.check(self.lib.bh_inverse_filter_trim())  # the pool of released staged-filter blocks (gigabytes each) goes too
        release_volume_pool()

    def fft_plans_replaced(self) -> int:
        """hipFFT 3-D plans that failed their creation-time round trip and were rebuilt decomposed."""
        import ctypes

        n = ctypes.c_int()
        _lib.check(self.lib.bh_ctx_fft_plans_replaced(self.handle, ctypes.byref(n)))
        return n.value


def get_context(device) -> Context:
    dev = resolve_device(device)
    ctx = _CTX.get(dev.index)
    if ctx is None:
        with torch.cuda.device(dev):
            torch.cuda.current_stream(dev)  # make sure torch has initialised the device
            ctx = Context(dev.index)
        _CTX[dev.index] = ctx
    ctx.bind_stream()
    return ctx


def ptr(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(t.data_ptr())


def upload(array: np.ndarray, device: torch.device) -> tuple[torch.Tensor, int]:
    """Host array -> contiguous device tensor in a dtype the kernels read natively.

    uint8/uint16/int16/float32 travel as they are (a uint16 camera stack crosses PCIe at 2 B/voxel
    and is widened to float32 inside the kernel); anything else is cast to float32 on the host, as
    the reference does with ``.to(dtype=torch.float32)`` (biahub/deskew.py:578).
    """
    a = np.asarray(array)
    if a.dtype not in _NP_TO_DT:
        a = a.astype(np.float32)
    a = np.ascontiguousarray(a)
    code = _NP_TO_DT[a.dtype]
    with volume_pool(device):
        return torch.from_numpy(a).to(device), code


_PIN_MIN_BYTES = 8 << 20  # small results are not worth a pinned allocation


def to_host(t: torch.Tensor) -> np.ndarray:
    """Device tensor -> numpy.  Large results land in pinned host memory from torch's caching host allocator: the copy
    runs at PCIe rate instead of through a pageable staging buffer, and a plate's worth of equally shaped results reuses
    the same blocks instead of page-faulting in gigabytes of fresh memory per volume (2.1 GB deskewed volume: 0.30 s ->
    0.09 s).  The array keeps its storage alive; the block returns to the cache when the array is dropped."""
    if not t.is_cuda:
        return t.numpy()
    t = t.contiguous()
    if t.numel() * t.element_size() < _PIN_MIN_BYTES:
        return t.cpu().numpy()
    try:
        host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    except RuntimeError:  # pinned memory exhausted / locked-memory limit: pageable is slower but always works
        return t.cpu().numpy()
    host.copy_(t)  # synchronous for a pinned destination copied with non_blocking=False
    return host.numpy()


def as_device_volume(x, device=None) -> tuple[torch.Tensor, int, torch.device]:
    """Accept a numpy array or a torch tensor; return (contiguous device tensor, dtype code, device)."""
    if isinstance(x, torch.Tensor):
        dev = resolve_device(x.device if device is None else device)
        with volume_pool(dev):
            t = x.to(dev)
            if t.dtype not in _TORCH_TO_DT:
                t = t.to(torch.float32)
            t = t.contiguous()
        return t, _TORCH_TO_DT[t.dtype], dev
    dev = resolve_device("cuda" if device is None else device)
    t, code = upload(x, dev)
    return t, code, dev
