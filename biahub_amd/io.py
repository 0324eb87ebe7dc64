"""OME-Zarr reader/writer (NGFF 0.4 on zarr v2, NGFF 0.5 on zarr v3 with optional sharding; directory stores) and the
per-position driver.

iohub / zarr / numcodecs are not available in this image, and the reference reaches them only through
``open_ome_zarr``, ``create_empty_plate`` and ``process_single_position`` (biahub/deskew.py:608-640,738-749).
This module provides those three entry points with the argument names the reference uses, over plain files:
HCS layout ``plate.zarr/<row>/<col>/<fov>/0`` with one 5-D ``(T,C,Z,Y,X)`` array per position on any regular chunk
grid; chunk bytes uncompressed or through biahub_amd.codecs (Blosc as iohub writes it, zstd, gzip/zlib, lz4; zarr v3
``bytes`` / ``blosc`` / ``zstd`` / ``gzip`` / ``crc32c`` / ``sharding_indexed``).  New stores default to chunks
``(1,1,zc,Y,X)``, "/" keys and no compression (`compressor="blosc"` gives iohub's default).
"""

from __future__ import annotations

import inspect
import json
import os
import threading
import zlib
from pathlib import Path

import numpy as np

from .array_ops import _check_nan_n_zeros

_AXES = [
    {"name": "T", "type": "time", "unit": "second"},
    {"name": "C", "type": "channel"},
    {"name": "Z", "type": "space", "unit": "micrometer"},
    {"name": "Y", "type": "space", "unit": "micrometer"},
    {"name": "X", "type": "space", "unit": "micrometer"},
]


def _write_json(path: Path, obj) -> None:
    # unique per writer: ranks / I/O threads that touch the same group must not share (and rename away) one temp file
    tmp = path.with_suffix(f"{path.suffix}.{os.getpid()}.{threading.get_ident()}.tmp")
    tmp.write_text(json.dumps(obj, indent=1))
    os.replace(tmp, path)


_IO_POOL = None


def _io_pool():
    """Shared pool for chunk I/O (file reads/writes, the entropy coders and the native shuffles release the GIL); size via
    BH_IO_THREADS (default: the cores this process may use, at most 16)."""
    global _IO_POOL
    if _IO_POOL is None:
        from concurrent.futures import ThreadPoolExecutor

        _IO_POOL = ThreadPoolExecutor(max_workers=max(1, int(os.environ.get("BH_IO_THREADS", 0) or _default_io_threads())),
                                      thread_name_prefix="bh-io")
    return _IO_POOL


def _default_io_threads() -> int:
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 8
    return max(1, min(16, n))


def _io_map(fn, items):
    items = list(items)
    if len(items) <= 1 or threading.current_thread().name.startswith("bh-io"):
        for it in items:  # already on a pool thread (read-ahead / write-behind): do not nest
            fn(it)
        return
    for r in _io_pool().map(fn, items):
        pass


_TORCH_DT = None


def _torch_dtype(dtype):
    """torch dtype of a numpy dtype a volume can have, or None."""
    global _TORCH_DT
    if _TORCH_DT is None:
        import torch

        _TORCH_DT = {np.dtype(np.uint8): torch.uint8, np.dtype(np.uint16): torch.uint16, np.dtype(np.int16): torch.int16,
                     np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64, np.dtype(np.int32): torch.int32}
    return _TORCH_DT.get(np.dtype(dtype))


def _host_volume(shape, dtype) -> np.ndarray:
    """Destination of a volume read: pinned host memory from torch's caching host allocator when a GPU is present (the
    upload that follows is then a direct DMA, and a plate's equally shaped volumes reuse the blocks), plain numpy else."""
    dtype = np.dtype(dtype)
    if int(np.prod(shape)) * dtype.itemsize >= (8 << 20):
        try:
            import torch

            if torch.cuda.is_available():
                tdt = _torch_dtype(dtype)
                if tdt is not None:
                    return torch.empty(tuple(int(v) for v in shape), dtype=tdt, pin_memory=True).numpy()
        except Exception:  # no torch / no pinned memory: pageable is always correct
            pass
    return np.empty(shape, dtype=dtype)


_V3_DTYPES = {"bool": "|b1", "int8": "|i1", "uint8": "|u1", "int16": "<i2", "uint16": "<u2", "int32": "<i4",
              "uint32": "<u4", "int64": "<i8", "uint64": "<u8", "float16": "<f2", "float32": "<f4", "float64": "<f8",
              "complex64": "<c8"}
_V3_NAMES = {np.dtype(v): k for k, v in _V3_DTYPES.items()}
_MISSING = 0xFFFFFFFFFFFFFFFF  # offset and length of an absent inner chunk in a shard index

BLOSC_DEFAULT = {"id": "blosc", "cname": "zstd", "clevel": 1, "shuffle": 2, "blocksize": 0}  # what iohub writes


def _fill(value, dtype):
    if value is None:
        return 0
    if isinstance(value, str):
        return {"NaN": np.nan, "Infinity": np.inf, "-Infinity": -np.inf}.get(value, 0) if dtype.kind == "f" else 0
    return value


class ZarrArray:
    """5-D ``(T,C,Z,Y,X)`` zarr array — v2 (``.zarray``, NGFF 0.4) or v3 (``zarr.json``, NGFF 0.5, optionally sharded) —
    read and written in whole (t, c) volumes.  Any regular chunk grid; chunk bytes through biahub_amd.codecs."""

    def __init__(self, path: Path):
        self.path = Path(path)
        self._locks: dict = {}
        self._locks_mu = threading.Lock()
        if (self.path / "zarr.json").exists():
            self._init_v3(json.loads((self.path / "zarr.json").read_text()))
        elif (self.path / ".zarray").exists():
            self._init_v2(json.loads((self.path / ".zarray").read_text()))
        else:
            raise FileNotFoundError(f"{path} is not a zarr array")
        if len(self.shape) != 5 or len(self.chunks) != 5:
            raise NotImplementedError(f"{path}: expected a 5-D (T,C,Z,Y,X) array, got shape {self.shape}")
        if any(c % i for c, i in zip(self.chunks, self.inner)):
            raise ValueError(f"{path}: shard shape {self.chunks} is not a multiple of the chunk shape {self.inner}")
        self.dtype = self.store_dtype.newbyteorder("=") if self.store_dtype.byteorder == ">" else self.store_dtype
        self.fill_value = _fill(self.fill_value, self.dtype)

    # ---- metadata -------------------------------------------------------------------------------------------
    def _init_v2(self, meta):
        from .codecs import codec_from_v2

        if meta.get("zarr_format") != 2:
            raise ValueError(f"{self.path}: .zarray with zarr_format {meta.get('zarr_format')}")
        if meta.get("filters"):
            raise NotImplementedError(f"{self.path}: zarr filters are not supported")
        if meta.get("order", "C") != "C":
            raise NotImplementedError(f"{self.path}: only C-order chunks are supported")
        self.zarr_format = 2
        self.shape = tuple(meta["shape"])
        self.chunks = self.inner = tuple(meta["chunks"])
        self.store_dtype = np.dtype(meta["dtype"])
        self.fill_value = meta.get("fill_value", 0)
        self.compressor = meta.get("compressor")
        c = codec_from_v2(self.compressor, self.store_dtype.itemsize)
        self.codecs = [c] if c is not None else []
        self.sharded = False
        sep = meta.get("dimension_separator", ".")
        self._key = lambda idx: sep.join(str(v) for v in idx)

    def _init_v3(self, meta):
        from .codecs import codec_from_v3

        if meta.get("zarr_format") != 3 or meta.get("node_type") != "array":
            raise ValueError(f"{self.path}: zarr.json is not a v3 array")
        self.zarr_format = 3
        self.shape = tuple(meta["shape"])
        if meta["data_type"] not in _V3_DTYPES:
            raise NotImplementedError(f"{self.path}: data type {meta['data_type']!r}")
        dt = np.dtype(_V3_DTYPES[meta["data_type"]])
        grid = meta["chunk_grid"]
        if grid.get("name") != "regular":
            raise NotImplementedError(f"{self.path}: chunk grid {grid.get('name')!r}")
        self.chunks = tuple(grid["configuration"]["chunk_shape"])
        enc = meta.get("chunk_key_encoding", {"name": "default"})
        sep = (enc.get("configuration") or {}).get("separator", "/" if enc.get("name") == "default" else ".")
        if enc.get("name") == "default":
            self._key = lambda idx: "c" + sep + sep.join(str(v) for v in idx)
        elif enc.get("name") == "v2":
            self._key = lambda idx: sep.join(str(v) for v in idx)
        else:
            raise NotImplementedError(f"{self.path}: chunk key encoding {enc.get('name')!r}")
        self.fill_value = meta.get("fill_value", 0)
        self.compressor = None

        def split(codecs):  # -> (endianness of the array->bytes stage or the sharding entry, bytes->bytes codecs)
            a2b, b2b = None, []
            for e in codecs:
                name = e.get("name")
                if name == "transpose":
                    order = list((e.get("configuration") or {}).get("order", []))
                    if order != sorted(order):
                        raise NotImplementedError(f"{self.path}: transposed chunks are not supported")
                elif name in ("bytes", "sharding_indexed"):
                    a2b = e
                else:
                    b2b.append(codec_from_v3(e, dt.itemsize))
            if a2b is None:
                raise ValueError(f"{self.path}: codec list without an array->bytes codec")
            return a2b, b2b

        a2b, outer_b2b = split(meta["codecs"])
        self.sharded = a2b["name"] == "sharding_indexed"
        if self.sharded:
            if outer_b2b:
                raise NotImplementedError(f"{self.path}: codecs after sharding_indexed are not supported")
            cfg = a2b["configuration"]
            self.inner = tuple(cfg["chunk_shape"])
            inner_a2b, self.codecs = split(cfg["codecs"])
            if inner_a2b["name"] != "bytes":
                raise NotImplementedError(f"{self.path}: nested sharding is not supported")
            idx_a2b, idx_b2b = split(cfg.get("index_codecs", [{"name": "bytes"}, {"name": "crc32c"}]))
            if idx_a2b["name"] != "bytes" or (idx_a2b.get("configuration") or {}).get("endian", "little") != "little" \
                    or [c.kind for c in idx_b2b] not in ([], ["crc32c"]):
                raise NotImplementedError(f"{self.path}: shard index codecs other than bytes(+crc32c) are not supported")
            self.index_crc = bool(idx_b2b)
            self.index_at_end = cfg.get("index_location", "end") == "end"
            a2b = inner_a2b
        else:
            self.inner = self.chunks
            self.codecs = outer_b2b
        endian = (a2b.get("configuration") or {}).get("endian", "little")
        self.store_dtype = dt.newbyteorder(">") if endian == "big" and dt.itemsize > 1 else dt

    # ---- chunk bytes ----------------------------------------------------------------------------------------
    def _chunk_path(self, idx) -> Path:
        return self.path / self._key(idx)

    def _lock(self, f: Path):
        with self._locks_mu:
            return self._locks.setdefault(str(f), threading.Lock())

    def _decode(self, buf) -> np.ndarray:
        n = int(np.prod(self.inner)) * self.store_dtype.itemsize
        raw = buf
        for c in reversed(self.codecs):
            raw = c.decode(raw, n)
        arr = np.frombuffer(raw, dtype=self.store_dtype, count=int(np.prod(self.inner))).reshape(self.inner)
        return arr

    def _encode(self, arr5: np.ndarray) -> bytes:
        raw = np.ascontiguousarray(arr5, dtype=self.store_dtype).view(np.uint8).reshape(-1)
        for c in self.codecs:
            raw = c.encode(raw if isinstance(raw, np.ndarray) else np.frombuffer(raw, np.uint8))
        return raw if isinstance(raw, (bytes, bytearray)) else raw.tobytes()

    def _index_nbytes(self) -> int:
        return 16 * int(np.prod([c // i for c, i in zip(self.chunks, self.inner)])) + (4 if self.index_crc else 0)

    def _read_index(self, fh, size: int) -> np.ndarray:
        from .codecs import crc32c

        n = self._index_nbytes()
        if size < n:
            raise OSError(f"{fh.name}: shard shorter than its index")
        fh.seek(size - n if self.index_at_end else 0)
        raw = fh.read(n)
        if self.index_crc:
            if crc32c(raw[:-4]) != int.from_bytes(raw[-4:], "little"):
                raise OSError(f"{fh.name}: shard index checksum mismatch")
            raw = raw[:-4]
        return np.frombuffer(raw, "<u8").reshape(tuple(c // i for c, i in zip(self.chunks, self.inner)) + (2,))

    def _write_file(self, f: Path, payload) -> None:
        f.parent.mkdir(parents=True, exist_ok=True)
        tmp = f.with_name(f.name + f".tmp{threading.get_ident()}")
        with open(tmp, "wb") as fh:
            for part in payload if isinstance(payload, list) else [payload]:
                fh.write(part)
        os.replace(tmp, f)

    def _write_shard(self, f: Path, pieces: dict) -> None:
        """pieces: inner-chunk grid index (5-tuple) -> encoded bytes; inner chunks not listed are absent."""
        from .codecs import crc32c

        grid = tuple(c // i for c, i in zip(self.chunks, self.inner))
        index = np.full(grid + (2,), _MISSING, dtype="<u8")
        pos = 0 if self.index_at_end else self._index_nbytes()
        body = []
        for key in sorted(pieces):
            index[key] = (pos, len(pieces[key]))
            body.append(pieces[key])
            pos += len(pieces[key])
        raw = index.tobytes()
        if self.index_crc:
            raw += crc32c(raw).to_bytes(4, "little")
        self._write_file(f, body + [raw] if self.index_at_end else [raw] + body)

    def _spatial_blocks(self):
        Z, Y, X = self.shape[2:]
        cz, cy, cx = self.chunks[2:]
        return [(zi, yi, xi) for zi in range(-(-Z // cz)) for yi in range(-(-Y // cy)) for xi in range(-(-X // cx))]

    # ---- volumes --------------------------------------------------------------------------------------------
    def read_volume(self, t: int, c: int) -> np.ndarray:
        """One (t, c) volume; its chunk files are read (and decoded) concurrently, straight into the result."""
        T, C, Z, Y, X = self.shape
        if not (0 <= t < T and 0 <= c < C):
            raise IndexError(f"(t, c) = ({t}, {c}) outside {self.shape[:2]}")
        ct, cc, cz, cy, cx = self.chunks
        it, ic, iz, iy, ix = self.inner
        out = _host_volume((Z, Y, X), self.dtype)
        plain = not self.sharded and not self.codecs and ct == 1 and cc == 1 and (cy, cx) == (Y, X) \
            and self.store_dtype == self.dtype

        def place(arr5, tt, cc_, z0, y0, x0):  # arr5: a decoded (inner) chunk whose corner sits at (z0, y0, x0)
            z1, y1, x1 = min(Z, z0 + arr5.shape[2]), min(Y, y0 + arr5.shape[3]), min(X, x0 + arr5.shape[4])
            out[z0:z1, y0:y1, x0:x1] = arr5[tt, cc_, : z1 - z0, : y1 - y0, : x1 - x0]

        def one(blk):
            zi, yi, xi = blk
            z0, y0, x0 = zi * cz, yi * cy, xi * cx
            z1, y1, x1 = min(Z, z0 + cz), min(Y, y0 + cy), min(X, x0 + cx)
            f = self._chunk_path((t // ct, c // cc, zi, yi, xi))
            try:
                fh = open(f, "rb")
            except FileNotFoundError:
                out[z0:z1, y0:y1, x0:x1] = self.fill_value
                return
            with fh:
                if plain and z1 - z0 == cz:
                    view = memoryview(out[z0:z1]).cast("B")
                    if fh.readinto(view) != len(view):
                        raise OSError(f"{f}: truncated chunk")
                    return
                if not self.sharded:
                    place(self._decode(fh.read()), t % ct, c % cc, z0, y0, x0)
                    return
                index = self._read_index(fh, os.fstat(fh.fileno()).st_size)
                st, sc = (t % ct) // it, (c % cc) // ic
                for kz in range(-(-(z1 - z0) // iz)):
                    for ky in range(-(-(y1 - y0) // iy)):
                        for kx in range(-(-(x1 - x0) // ix)):
                            off, nb = (int(v) for v in index[st, sc, kz, ky, kx])
                            q0, r0, s0 = z0 + kz * iz, y0 + ky * iy, x0 + kx * ix
                            if off == _MISSING and nb == _MISSING:
                                out[q0:min(Z, q0 + iz), r0:min(Y, r0 + iy), s0:min(X, s0 + ix)] = self.fill_value
                                continue
                            fh.seek(off)
                            place(self._decode(fh.read(nb)), (t % ct) % it, (c % cc) % ic, q0, r0, s0)

        _io_map(one, self._spatial_blocks())
        return out

    def write_volume(self, t: int, c: int, vol: np.ndarray) -> None:
        T, C, Z, Y, X = self.shape
        if vol.shape != (Z, Y, X):
            raise ValueError(f"volume shape {vol.shape} does not match array {(Z, Y, X)}")
        if not (0 <= t < T and 0 <= c < C):
            raise IndexError(f"(t, c) = ({t}, {c}) outside {self.shape[:2]}")
        ct, cc, cz, cy, cx = self.chunks
        it, ic, iz, iy, ix = self.inner
        vol = np.ascontiguousarray(vol, dtype=self.dtype)
        plain = not self.sharded and not self.codecs and ct == 1 and cc == 1 and self.store_dtype == self.dtype

        def padded(z0, y0, x0, ez, ey, ex):  # the (ez, ey, ex) box at that corner, fill value outside the array
            z1, y1, x1 = min(Z, z0 + ez), min(Y, y0 + ey), min(X, x0 + ex)
            sub = vol[z0:z1, y0:y1, x0:x1]
            if sub.shape == (ez, ey, ex):
                return sub
            box = np.full((ez, ey, ex), self.fill_value, dtype=self.dtype)
            box[: z1 - z0, : y1 - y0, : x1 - x0] = sub
            return box

        def one(blk):
            zi, yi, xi = blk
            z0, y0, x0 = zi * cz, yi * cy, xi * cx
            f = self._chunk_path((t // ct, c // cc, zi, yi, xi))
            if not self.sharded:
                box = padded(z0, y0, x0, cz, cy, cx)
                if ct == 1 and cc == 1:
                    self._write_file(f, memoryview(np.ascontiguousarray(box)).cast("B") if plain else self._encode(box[None, None]))
                    return
                with self._lock(f):  # the chunk file holds other (t, c) too: read, modify, write
                    try:
                        arr5 = self._decode(f.read_bytes()).astype(self.dtype)
                    except FileNotFoundError:
                        arr5 = np.full(self.chunks, self.fill_value, dtype=self.dtype)
                    arr5[t % ct, c % cc] = box
                    self._write_file(f, self._encode(arr5))
                return
            st, sc = (t % ct) // it, (c % cc) // ic
            alone = ct == 1 and cc == 1  # the shard holds this (t, c) only: no need to look at what is on disk
            with self._lock(f):
                pieces, old = {}, None
                if not alone and f.exists():
                    with open(f, "rb") as fh:
                        index = self._read_index(fh, os.fstat(fh.fileno()).st_size)
                        for key in np.ndindex(*index.shape[:-1]):
                            off, nb = (int(v) for v in index[key])
                            if not (off == _MISSING and nb == _MISSING):
                                fh.seek(off)
                                pieces[key] = fh.read(nb)
                for kz in range(cz // iz):
                    for ky in range(cy // iy):
                        for kx in range(cx // ix):
                            q0, r0, s0 = z0 + kz * iz, y0 + ky * iy, x0 + kx * ix
                            if q0 >= Z or r0 >= Y or s0 >= X:
                                continue
                            box = padded(q0, r0, s0, iz, iy, ix)
                            key = (st, sc, kz, ky, kx)
                            if it == 1 and ic == 1:
                                pieces[key] = self._encode(box[None, None])
                            else:
                                old = pieces.get(key)
                                arr5 = self._decode(old).astype(self.dtype) if old is not None else \
                                    np.full(self.inner, self.fill_value, dtype=self.dtype)
                                arr5[(t % ct) % it, (c % cc) % ic] = box
                                pieces[key] = self._encode(arr5)
                self._write_shard(f, pieces)

        _io_map(one, self._spatial_blocks())

    # ---- volumes straight to / from the GPU ----------------------------------------------------------------
    def _plane_chunks(self):
        """[(file index, inner key or None, z0, planes)] when every (inner) chunk is a stack of whole planes of one
        (t, c) and the only codec is Blosc (or none) — the layout iohub writes — else None."""
        Z, Y, X = self.shape[2:]
        if self.inner[:2] != (1, 1) or self.inner[3:] != (Y, X) or self.chunks[3:] != (Y, X):
            return None
        if len(self.codecs) > 1 or (self.codecs and self.codecs[0].kind != "blosc") or self.store_dtype != self.dtype:
            return None
        cz, iz = self.chunks[2], self.inner[2]
        out = []
        for zi in range(-(-Z // cz)):
            for kz in range(cz // iz):
                z0 = zi * cz + kz * iz
                if z0 < Z:
                    out.append((zi, kz if self.sharded else None, z0, iz))
        return out

    def read_volume_device(self, t: int, c: int, device=None):
        """One (t, c) volume as a device tensor.  Blosc chunks are entropy-decoded on the I/O threads into one pinned
        staging block, uploaded still shuffled, and un-shuffled by ``bh_blosc_unfilter`` straight into the volume; other
        layouts are read on the host and uploaded.  = ``stage_volume`` (host half) + ``upload_staged`` (device half)."""
        import torch

        from .device import resolve_device, volume_pool

        dev = resolve_device("cuda" if device is None else device)
        staged = self.stage_volume(t, c)
        if staged is None:
            with volume_pool(dev):  # the volume gets the library's page layout (device.volume_pool)
                return torch.from_numpy(self.read_volume(t, c)).to(dev)
        return self.upload_staged(staged, dev)

    def stage_volume(self, t: int, c: int):
        """The HOST half of ``read_volume_device``, safe on any thread (no GPU call): the (t, c) volume's Blosc chunks are read
        and entropy-decoded on the I/O threads into one pinned staging block, still permuted (lz4 frames stay compressed: their
        block codec runs on the GPU).  Returns an opaque object for ``upload_staged``, or None when the layout has no device
        path (not plane-stack Blosc chunks)."""
        import torch

        from . import codecs

        T, C, Z, Y, X = self.shape
        if not (0 <= t < T and 0 <= c < C):
            raise IndexError(f"(t, c) = ({t}, {c}) outside {self.shape[:2]}")
        plan = self._plane_chunks()
        if plan is None or _torch_dtype(self.dtype) is None or not self.codecs:
            return None
        ct, cc = self.chunks[:2]
        cbytes = self.inner[2] * Y * X * self.dtype.itemsize
        stage = torch.empty(len(plan) * cbytes, dtype=torch.uint8, pin_memory=True)
        stage_np = stage.numpy()
        heads: list = [None] * len(plan)
        frames: dict = {}  # chunks whose LZ4 blocks are decoded on the GPU: the compressed frame travels instead of the raw bytes
        lz4_dev = os.environ.get("BH_LZ4_DEVICE", "1") != "0"

        def one(i):
            zi, kz, z0, _ = plan[i]
            f = self._chunk_path((t // ct, c // cc, zi, 0, 0))
            try:
                fh = open(f, "rb")
            except FileNotFoundError:
                return
            with fh:
                if kz is None:
                    buf = fh.read()
                else:
                    index = self._read_index(fh, os.fstat(fh.fileno()).st_size)
                    off, nb = (int(v) for v in index[(t % ct), (c % cc), kz, 0, 0])
                    if off == _MISSING and nb == _MISSING:
                        return
                    fh.seek(off)
                    buf = fh.read(nb)
            h0 = codecs.BloscHeader(buf)
            if lz4_dev and h0.codec == "lz4" and not h0.memcpyed and h0.nbytes == cbytes:
                heads[i] = h0
                frames[i] = bytes(buf[: h0.cbytes])
                return
            h, _ = codecs.blosc_decode_blocks(buf, out=stage_np[i * cbytes:(i + 1) * cbytes])
            heads[i] = h

        _io_map(one, range(len(plan)))
        return {"stage": stage, "heads": heads, "frames": frames, "plan": plan, "cbytes": cbytes}

    def upload_staged(self, staged, device=None):
        """The DEVICE half of ``read_volume_device`` (on the thread that owns the GPU context): upload the staging block, decode
        the LZ4 frames, un-shuffle into the volume.  Returns the (Z, Y, X) device tensor."""
        import torch

        from . import codecs
        from .device import empty as device_empty, resolve_device

        dev = resolve_device("cuda" if device is None else device)
        T, C, Z, Y, X = self.shape
        tdt = _torch_dtype(self.dtype)
        stage, heads, frames, plan, cbytes = (staged[k] for k in ("stage", "heads", "frames", "plan", "cbytes"))
        out = device_empty((Z, Y, X), tdt, dev)
        out8 = out.view(torch.uint8).reshape(-1)
        dstage = stage.to(dev, non_blocking=True)
        plane = Y * X * self.dtype.itemsize
        tmp = None
        if frames:  # LZ4 blocks -> permuted bytes on the device (csrc/lz4.hip), in the slots the host decoder would have filled:
            ks = sorted(frames)  # every chunk of the volume in one upload and one launch
            codecs.blosc_lz4_decode_frames_device([frames[k] for k in ks], dstage, [k * cbytes for k in ks])
        for i, (zi, kz, z0, iz) in enumerate(plan):
            h = heads[i]
            dst = out8[z0 * plane:min(Z, z0 + iz) * plane]
            if h is None:
                out[z0:z0 + iz] = self.fill_value
                continue
            mode = codecs.BLOSC_NOSHUFFLE if h.memcpyed else h.shuffle_mode
            src = dstage[i * cbytes:(i + 1) * cbytes]
            if dst.numel() == cbytes:
                codecs.unfilter_device(src, dst, h.blocksize, h.typesize, mode)
            else:  # the last chunk overhangs the array: un-shuffle all of it, keep the planes inside
                tmp = torch.empty(cbytes, dtype=torch.uint8, device=dev) if tmp is None else tmp
                codecs.unfilter_device(src, tmp, h.blocksize, h.typesize, mode)
                dst.copy_(tmp[: dst.numel()])
        torch.cuda.current_stream(dev).synchronize()  # the pinned block may be recycled once we return
        return out

    def write_volume_device(self, t: int, c: int, vol) -> None:
        """Store a device tensor as the (t, c) volume.  For Blosc plane-stack chunks the shuffle runs on the GPU
        (``bh_blosc_filter``), with the lz4 inner codec the block codec too (``bh_blosc_lz4_compress``: only compressed frames
        cross PCIe); the download lands in pinned memory and the I/O threads run the entropy coder (zstd / zlib) or just write."""
        self.encode_volume_device(t, c, vol)()

    def encode_volume_device(self, t: int, c: int, vol):
        """The device half of ``write_volume_device``, on the CALLING thread (a ``bh_ctx`` is not thread-safe): permute, compress
        where the device can, download.  Returns a callable that does the host half — entropy coding where left, file writes —
        and may run on another thread (``process_single_position`` hands it to its writer thread)."""
        import torch

        from . import codecs
        from .device import to_host

        T, C, Z, Y, X = self.shape
        if tuple(vol.shape) != (Z, Y, X):
            raise ValueError(f"volume shape {tuple(vol.shape)} does not match array {(Z, Y, X)}")
        plan = self._plane_chunks()
        cfg = self.codecs[0].cfg if self.codecs else {}
        def host_path():
            h = to_host(vol) if vol.is_cuda else np.asarray(vol)
            return lambda: self.write_volume(t, c, h)

        if plan is None or not self.codecs or not vol.is_cuda or cfg.get("cname", "zstd") not in ("zstd", "lz4", "zlib") \
                or codecs.default_blocksize(self.dtype.itemsize) > self.inner[2] * Y * X * self.dtype.itemsize:
            return host_path()
        ct, cc = self.chunks[:2]
        if (ct, cc) != (1, 1) or self.fill_value != 0:
            return host_path()
        want = _torch_dtype(self.dtype)
        if want is None:
            return host_path()
        v = vol.contiguous()
        if v.dtype != want:
            v = v.to(torch.float32).to(want) if want in (torch.uint16,) else v.to(want)
        ts = self.dtype.itemsize
        iz = self.inner[2]
        plane = Y * X * ts
        cbytes = iz * plane
        bsz = int(cfg.get("blocksize", 0)) or codecs.default_blocksize(ts)
        mode = int(cfg.get("shuffle", codecs.BLOSC_BITSHUFFLE))
        v8 = v.view(torch.uint8).reshape(-1)
        prof = os.environ.get("BH_PIPE_TIMING") == "2"  # stage seconds of this call on stderr (synchronises between stages)
        if prof:
            import sys as _sys
            import time as _time

            torch.cuda.synchronize(v.device)
            _tp = [_time.perf_counter()]

            def _mark(name):
                torch.cuda.synchronize(v.device)
                _tp.append(_time.perf_counter())
                print(f"   encode_volume_device {name}: {_tp[-1] - _tp[-2]:.3f} s", file=_sys.stderr, flush=True)
        dstage = torch.empty(len(plan) * cbytes, dtype=torch.uint8, device=v.device)
        if prof:
            _mark("(operator's queued work +) staging allocation")
        for i, (zi, kz, z0, _) in enumerate(plan):
            src = v8[z0 * plane:min(Z, z0 + iz) * plane]
            if src.numel() != cbytes:  # the overhanging chunk is padded (fill value 0: all-zero bytes)
                full = torch.zeros(cbytes, dtype=torch.uint8, device=v.device)
                full[: src.numel()] = src
                src = full
            codecs.filter_device(src, dstage[i * cbytes:(i + 1) * cbytes], bsz, ts, mode)
        per_file: dict = {}
        if prof:
            _mark("permutation")
        if cfg.get("cname", "zstd") == "lz4" and cbytes >= 128 and os.environ.get("BH_LZ4_DEVICE", "1") != "0":
            # the block codec runs on the GPU too (csrc/lz4.hip): only the finished frames cross PCIe, the I/O threads just write
            packed, offs = codecs.blosc_lz4_compress_device(dstage, len(plan), cbytes, bsz, ts, mode)
            if prof:
                _mark("lz4 + frames")
            hostp = to_host(packed[: offs[-1]])
            if prof:
                _mark("download")

            def put(i):
                zi, kz, z0, _ = plan[i]
                nb = int(np.frombuffer(hostp[offs[i] + 12: offs[i] + 16], np.uint32)[0])  # the frame's own length (header)
                blob = hostp[offs[i]: offs[i] + nb]  # (a view of the pinned block: a bytes copy of it would hold the GIL for ~0.1 s per frame,
                # stalling the operator's thread, which runs Python between its GPU calls)
                if kz is None:
                    self._write_file(self._chunk_path((t, c, zi, 0, 0)), blob)
                else:
                    per_file.setdefault(zi, {})[(0, 0, kz, 0, 0)] = blob

            def commit_frames():
                _io_map(put, range(len(plan)))
                for zi, pieces in per_file.items():
                    self._write_shard(self._chunk_path((t, c, zi, 0, 0)), pieces)

            return commit_frames
        host = to_host(dstage)

        def one(i):
            zi, kz, z0, _ = plan[i]
            blob = codecs.blosc_compress(host[i * cbytes:(i + 1) * cbytes], ts, cfg.get("cname", "zstd"), cfg.get("clevel", 1),
                                         mode, bsz, prefiltered=True)
            if kz is None:
                self._write_file(self._chunk_path((t, c, zi, 0, 0)), blob)
            else:
                per_file.setdefault(zi, {})[(0, 0, kz, 0, 0)] = blob

        def commit():
            _io_map(one, range(len(plan)))
            for zi, pieces in per_file.items():
                self._write_shard(self._chunk_path((t, c, zi, 0, 0)), pieces)

        return commit

    def __getitem__(self, key) -> np.ndarray:
        if not isinstance(key, tuple):
            key = (key,)
        if len(key) == 1:  # arr[t] -> (C, Z, Y, X), like the dask / zarr arrays of the reference
            return np.stack([self.read_volume(int(key[0]), ci) for ci in range(self.shape[1])])
        t, c = key[0], key[1]
        rest = key[2:] if len(key) > 2 else ()
        if isinstance(c, (list, tuple, np.ndarray)):
            vol = np.stack([self.read_volume(int(t), int(ci)) for ci in c])
            return vol[(slice(None),) + tuple(rest)] if rest else vol
        vol = self.read_volume(int(t), int(c))
        return vol[tuple(rest)] if rest else vol

    def __setitem__(self, key, value) -> None:
        t, c = key[0], key[1]
        if isinstance(c, (list, tuple, np.ndarray)):
            for ci, v in zip(c, value):
                self.write_volume(int(t), int(ci), v)
        else:
            self.write_volume(int(t), int(c), np.asarray(value))


_OME_KEYS = ("multiscales", "omero", "plate", "well", "bioformats2raw.layout", "labels", "image-label")


def _group_format(path: Path) -> int | None:
    if (path / "zarr.json").exists():
        return 3
    if (path / ".zgroup").exists():
        return 2
    return None


def _read_group_attrs(path: Path) -> dict:
    """Attributes of a group with the NGFF keys at top level whatever the layout: v2 keeps them in ``.zattrs``, NGFF 0.5
    nests them under ``attributes["ome"]`` of ``zarr.json``."""
    fmt = _group_format(path)
    if fmt == 3:
        meta = json.loads((path / "zarr.json").read_text())
        if meta.get("node_type") != "group":
            raise FileNotFoundError(f"{path} is not a zarr group")
        attrs = dict(meta.get("attributes", {}))
        ome = attrs.pop("ome", {})
        flat = {k: v for k, v in ome.items() if k != "version"}
        if "version" in ome:
            flat["_ome_version"] = ome["version"]
        flat.update(attrs)
        return flat
    if fmt == 2:
        return json.loads((path / ".zattrs").read_text()) if (path / ".zattrs").exists() else {}
    raise FileNotFoundError(f"{path} is not a zarr group")


def _write_group(path: Path, fmt: int, attrs: dict | None = None, version: str = "0.5") -> None:
    path.mkdir(parents=True, exist_ok=True)
    attrs = dict(attrs or {})
    attrs.pop("_ome_version", None)
    if fmt == 2:
        _write_json(path / ".zgroup", {"zarr_format": 2})
        if attrs or (path / ".zattrs").exists():
            _write_json(path / ".zattrs", attrs)
        return
    ome = {k: attrs.pop(k) for k in list(attrs) if k in _OME_KEYS}
    out = dict(attrs)
    if ome:
        out["ome"] = {"version": version, **ome}
    _write_json(path / "zarr.json", {"zarr_format": 3, "node_type": "group", "attributes": out})


class Position:
    """One FOV group: ``<store>/<row>/<col>/<fov>`` with array "0" (what ``open_ome_zarr(position_path)`` yields)."""

    def __init__(self, path):
        self.path = Path(path)
        self.zarr_format = _group_format(self.path)
        if self.zarr_format is None:
            raise FileNotFoundError(f"{path} is not a zarr group")
        self.zattrs = _read_group_attrs(self.path)
        # a transfer-function store (waveorder's "fov" layout, biahub/compute_transfer_function.py:36) holds named arrays
        # and no "0"
        self.data = ZarrArray(self.path / "0") if self._is_array(self.path / "0") else None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def close(self) -> None:
        """Nothing is held open between calls; present because callers of iohub's nodes close them."""

    @staticmethod
    def _is_array(p: Path) -> bool:
        return (p / ".zarray").exists() or ((p / "zarr.json").exists() and
                                            json.loads((p / "zarr.json").read_text()).get("node_type") == "array")

    def array_keys(self) -> list[str]:
        return sorted(c.name for c in self.path.iterdir() if c.is_dir() and self._is_array(c))

    def __getitem__(self, name):
        if str(name) == "0" and self.data is not None:
            return self.data
        if not self._is_array(self.path / str(name)):
            raise KeyError(name)
        return ZarrArray(self.path / str(name))

    def create_image(self, name: str, data: np.ndarray, chunks=None, compressor=None) -> "ZarrArray":
        """Write a named 5-D array next to (or instead of) "0" — iohub's ``Position.create_image`` as waveorder's
        compute-tf uses it for the transfer functions."""
        data = np.asarray(data)
        if data.ndim != 5:
            raise ValueError("create_image expects a (T, C, Z, Y, X) array")
        T, C, Z, Y, X = data.shape
        if chunks is None:
            chunks = (1, 1, max(1, min(Z, (64 << 20) // max(1, Y * X * data.dtype.itemsize))), Y, X)
        _create_array(self.path / str(name), data.shape, tuple(int(c) for c in chunks), data.dtype, self.zarr_format,
                      dict(BLOSC_DEFAULT) if compressor == "blosc" else compressor, None)
        arr = ZarrArray(self.path / str(name))
        for t in range(T):
            for c in range(C):
                arr.write_volume(t, c, data[t, c])
        return arr

    @property
    def channel_names(self) -> list[str]:
        return [ch["label"] for ch in self.zattrs.get("omero", {}).get("channels", [])]

    @property
    def scale(self) -> list[float]:
        ds = self.zattrs["multiscales"][0]["datasets"][0]
        for tr in ds.get("coordinateTransformations", []):
            if tr.get("type") == "scale":
                return [float(v) for v in tr["scale"]]
        return [1.0] * 5

    @property
    def version(self) -> str:
        if "_ome_version" in self.zattrs:
            return str(self.zattrs["_ome_version"])
        return str(self.zattrs.get("multiscales", [{}])[0].get("version", "0.5" if self.zarr_format == 3 else "0.4"))

    def update_zattrs(self, extra: dict) -> None:
        self.zattrs.update(extra)
        _write_group(self.path, self.zarr_format, self.zattrs, self.version)


def open_ome_zarr(path, mode: str = "r", layout: str = "auto") -> Position:
    """Open one position (``layout="fov"`` and HCS positions look the same on disk below the FOV group)."""
    return Position(path)


def read_fov_array(path) -> ZarrArray:
    return Position(path).data


def _position_zattrs(channel_names, scale, version, extra=None):
    ms = {
        "axes": _AXES,
        "datasets": [{"path": "0", "coordinateTransformations": [{"type": "scale", "scale": [float(s) for s in scale]}]}],
        "name": "0",
    }
    omero = {"channels": [{"label": n, "active": True, "color": "FFFFFF",
                           "window": {"start": 0, "end": 65535, "min": 0, "max": 65535}} for n in channel_names]}
    if version == "0.4":  # 0.5 carries the version once, on the enclosing "ome" object
        ms = {"version": version, **ms}
        omero = {"version": version, **omero}
    z = {"multiscales": [ms], "omero": omero}
    if extra:
        z.update(extra)
    return z


def _v3_codecs(compressor, itemsize):
    """bytes->bytes codec entries of a zarr v3 array for a numcodecs-style ``compressor`` configuration."""
    if compressor is None:
        return []
    cid = compressor.get("id")
    if cid == "blosc":
        sh = compressor.get("shuffle", 1)
        sh = {0: "noshuffle", 1: "shuffle", 2: "bitshuffle", -1: "bitshuffle" if itemsize == 1 else "shuffle"}[int(sh)]
        return [{"name": "blosc", "configuration": {"cname": compressor.get("cname", "zstd"), "clevel": compressor.get("clevel", 1),
                                                    "shuffle": sh, "typesize": itemsize,
                                                    "blocksize": compressor.get("blocksize", 0)}}]
    if cid == "zstd":
        return [{"name": "zstd", "configuration": {"level": compressor.get("level", 1), "checksum": False}}]
    if cid in ("gzip", "zlib"):
        return [{"name": "gzip", "configuration": {"level": compressor.get("level", 1)}}]
    raise NotImplementedError(f"compressor {cid!r} has no zarr v3 codec here")


def create_empty_position(path, channel_names, shape, chunks=None, scale=(1, 1, 1, 1, 1), dtype=np.float32,
                          version="0.4", compressor=None, metadata=None, shards_ratio=None) -> None:
    """Idempotent: an existing position with the same shape is left alone (reference: deskew.py:608-610).
    ``version`` "0.4" writes a zarr v2 hierarchy, "0.5" zarr v3; ``shards_ratio`` (v3) groups that many chunks per axis
    into one shard file (`biahub/settings.py:460`).  ``compressor``: None, a numcodecs-style dict, or "blosc" for
    what iohub writes (Blosc, zstd level 1, bit shuffle)."""
    path = Path(path)
    version = str(version)
    if version not in ("0.4", "0.5"):
        raise ValueError(f"OME-Zarr version {version!r} (0.4 or 0.5)")
    T, C, Z, Y, X = (int(s) for s in shape)
    if len(channel_names) != C:
        raise ValueError(f"{len(channel_names)} channel names for C={C}")
    if (path / "0" / ".zarray").exists() or (path / "0" / "zarr.json").exists():
        if ZarrArray(path / "0").shape == (T, C, Z, Y, X):
            return
        raise ValueError(f"{path} exists with a different shape")
    if compressor == "blosc":
        compressor = dict(BLOSC_DEFAULT)
    dt = np.dtype(dtype)
    if chunks is None:
        zc = max(1, min(Z, (64 << 20) // max(1, Y * X * dt.itemsize)))
        chunks = (1, 1, zc, Y, X)
    chunks = tuple(int(c) for c in chunks)
    fmt = 2 if version == "0.4" else 3
    _write_group(path, fmt, _position_zattrs(channel_names, scale, version, metadata), version)
    _create_array(path / "0", (T, C, Z, Y, X), chunks, dt, fmt, compressor, shards_ratio)


def create_empty_fov(path, channel_names, scale=(1, 1, 1, 1, 1), version="0.4", metadata=None) -> "Position":
    """A position group without an array "0": the container of a transfer-function store (``open_ome_zarr(path,
    layout="fov", mode="w", channel_names=...)`` in waveorder's compute-tf), filled with ``Position.create_image``."""
    path = Path(path)
    version = str(version)
    fmt = 2 if version == "0.4" else 3
    path.mkdir(parents=True, exist_ok=True)
    _write_group(path, fmt, _position_zattrs(channel_names, scale, version, metadata), version)
    return Position(path)


def _create_array(apath: Path, shape, chunks, dt, fmt: int, compressor, shards_ratio) -> None:
    """Metadata of one 5-D array (zarr v2 ``.zarray`` or v3 ``zarr.json``); chunks appear as they are written."""
    apath = Path(apath)
    dt = np.dtype(dt)
    T, C, Z, Y, X = (int(v) for v in shape)
    apath.mkdir(parents=True, exist_ok=True)
    if fmt == 2:
        if shards_ratio:
            raise ValueError("sharding needs OME-Zarr 0.5 (zarr v3)")
        _write_json(apath / ".zarray", {
            "zarr_format": 2, "shape": [T, C, Z, Y, X], "chunks": list(chunks),
            "dtype": dt.str, "compressor": compressor, "fill_value": 0, "filters": None, "order": "C",
            "dimension_separator": "/"})
        return
    if dt not in _V3_NAMES:
        raise NotImplementedError(f"dtype {dt} has no zarr v3 name here")
    inner = [{"name": "bytes", "configuration": {"endian": "little"}}] + _v3_codecs(compressor, dt.itemsize)
    if shards_ratio:
        ratio = tuple(int(r) for r in shards_ratio)
        if len(ratio) != 5 or min(ratio) < 1:
            raise ValueError("shards_ratio needs five positive integers (T, C, Z, Y, X)")
        outer = tuple(c * r for c, r in zip(chunks, ratio))
        codecs = [{"name": "sharding_indexed", "configuration": {
            "chunk_shape": list(chunks), "codecs": inner,
            "index_codecs": [{"name": "bytes", "configuration": {"endian": "little"}}, {"name": "crc32c"}],
            "index_location": "end"}}]
    else:
        outer, codecs = chunks, inner
    _write_json(apath / "zarr.json", {
        "zarr_format": 3, "node_type": "array", "shape": [T, C, Z, Y, X], "data_type": _V3_NAMES[dt],
        "chunk_grid": {"name": "regular", "configuration": {"chunk_shape": list(outer)}},
        "chunk_key_encoding": {"name": "default", "configuration": {"separator": "/"}},
        "fill_value": 0, "codecs": codecs, "attributes": {}, "dimension_names": ["T", "C", "Z", "Y", "X"],
        "storage_transformers": []})


def create_empty_plate(store_path, position_keys, channel_names, shape, chunks=None, scale=(1, 1, 1, 1, 1),
                       dtype=np.float32, version="0.4", compressor=None, metadata=None, shards_ratio=None,
                       metadata_sources=None, metadata_keys=None) -> None:
    """HCS plate with empty positions — argument names follow iohub's ``create_empty_plate`` as the reference calls
    it (biahub/deskew.py:629-640, register.py:488-504).  ``metadata_sources`` (a plate holding the same row/col/fov
    positions) and ``metadata_keys`` (names or ``fnmatch`` patterns, e.g. ``PROVENANCE_METADATA_KEYS``) copy the matching
    per-position attributes of the source into the new positions, as the reference's apply-inv-tf asks for
    (biahub/apply_inverse_transfer_function.py:66-72)."""
    from fnmatch import fnmatchcase

    def carried(key):
        if not (metadata_sources and metadata_keys):
            return metadata
        src = Path(metadata_sources).joinpath(*(str(k) for k in key))
        if _group_format(src) is None:
            return metadata
        found = {k: v for k, v in _read_group_attrs(src).items() if any(fnmatchcase(k, pat) for pat in metadata_keys)}
        return {**found, **(metadata or {})} if found else metadata

    store = Path(store_path)
    version = str(version)
    fmt = _group_format(store) or (2 if version == "0.4" else 3)
    if (fmt == 2) != (version == "0.4"):
        raise ValueError(f"{store} is a zarr v{fmt} store; cannot add OME-Zarr {version} positions to it")
    attrs = _read_group_attrs(store) if _group_format(store) else {}
    plate = attrs.get("plate", {"rows": [], "columns": [], "wells": []})
    if version == "0.4":
        plate.setdefault("version", version)
    for key in position_keys:
        row, col, fov = (str(k) for k in key)
        if {"name": row} not in plate["rows"]:
            plate["rows"].append({"name": row})
        if {"name": col} not in plate["columns"]:
            plate["columns"].append({"name": col})
        wpath = f"{row}/{col}"
        if not any(w["path"] == wpath for w in plate["wells"]):
            plate["wells"].append({"path": wpath, "rowIndex": [r["name"] for r in plate["rows"]].index(row),
                                   "columnIndex": [c["name"] for c in plate["columns"]].index(col)})
        if _group_format(store / row) is None:
            _write_group(store / row, fmt, None, version)
        well = store / row / col
        wattrs = _read_group_attrs(well) if _group_format(well) else {}
        wmeta = wattrs.get("well", {"images": []})
        if version == "0.4":
            wmeta.setdefault("version", version)
        if not any(i["path"] == fov for i in wmeta["images"]):
            wmeta["images"].append({"path": fov})
        wattrs["well"] = wmeta
        _write_group(well, fmt, wattrs, version)
        create_empty_position(well / fov, channel_names, shape, chunks, scale, dtype, version, compressor, carried(key),
                              shards_ratio)
    attrs["plate"] = plate
    _write_group(store, fmt, attrs, version)


def _is_device_tensor(x) -> bool:
    try:
        import torch
    except ImportError:  # pragma: no cover
        return False
    return isinstance(x, torch.Tensor) and x.is_cuda


def process_single_position(func, input_position_path, output_position_path, input_channel_indices=None,
                            output_channel_indices=None, input_time_indices=None, output_time_indices=None,
                            num_workers: int = 1, resume: bool = False, resume_token: str | None = None, **kwargs):
    """Apply ``func(czyx, **kwargs) -> czyx`` to every (time, channel-group) unit of one position.

    Mirrors how the reference drives its operators (call sites biahub/deskew.py:738-749,
    register.py:556-574, stabilize.py:287-300): one channel per call by default, all-zero/NaN inputs skipped,
    ``input_time_index`` injected when ``func`` declares it, ``extra_metadata`` moved into the output zattrs,
    finished units recorded for ``resume`` under the ``resume_token``.  The operator runs serially in this process
    (the GPU is the parallel resource, ``num_workers`` is accepted and ignored); reading the next unit and writing
    the previous result overlap it on I/O threads.
    """
    extra_metadata = kwargs.pop("extra_metadata", None)
    src, dst = Position(input_position_path), Position(output_position_path)
    T, C = src.data.shape[:2]
    in_t = list(range(T)) if input_time_indices is None else list(input_time_indices)
    out_t = in_t if output_time_indices is None else list(output_time_indices)
    in_c = [[c] for c in range(C)] if input_channel_indices is None else [list(g) for g in input_channel_indices]
    out_c = in_c if output_channel_indices is None else [list(g) for g in output_channel_indices]
    if len(in_t) != len(out_t) or len(in_c) != len(out_c):
        raise ValueError("input and output index lists must have the same length")
    wants_t = "input_time_index" in inspect.signature(func).parameters
    done_file = dst.path / ".biahub_amd_done.json"
    done = {}
    if resume and done_file.exists():
        rec = json.loads(done_file.read_text())
        if rec.get("token") == resume_token:
            done = rec.get("units", {})
    units = []
    for ti, to in zip(in_t, out_t):
        for gi, go in zip(in_c, out_c):
            key = f"{to}:{','.join(map(str, go))}"
            if not (resume and done.get(key)):
                units.append((ti, to, gi, go, key))

    # Three-stage pipeline: the next unit is read (chunks in parallel) while this one is on the GPU and the previous
    # result is being written; the operator itself runs on this thread only (a bh_ctx is not thread-safe).
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor

    import time as _time

    # BH_PIPE_TIMING=1: one line per position on stderr with the seconds each unit spent in each stage (load and store run on
    # their own threads; wait = the operator's thread idle for the next unit's load)
    timing = {"load": [], "wait": [], "func": [], "encode": [], "store": []} if os.environ.get("BH_PIPE_TIMING") else None

    # An operator that takes device tensors (`device_input`, e.g. the deskew adapter of the CLI) gets its volumes through the
    # store's device read path: the reader thread does the host half (file reads + entropy decoding into a pinned block), this
    # thread the device half (upload, LZ4 blocks, un-shuffle) — the host un-shuffle and the stacking copy drop out of the read.
    dev_in = None
    if getattr(func, "device_input", False) and os.environ.get("BH_PIPE_DEVICE_INPUT", "1") != "0":
        try:
            import torch as _torch

            if _torch.cuda.is_available() and not str(kwargs.get("device", "cuda")).startswith("cpu"):
                from .device import resolve_device

                dev_in = resolve_device(kwargs.get("device", "cuda"))
        except ImportError:
            dev_in = None

    def load(u):
        t0 = _time.perf_counter()
        r = None
        if dev_in is not None:
            staged = [src.data.stage_volume(u[0], c) for c in u[2]]
            if all(s_ is not None for s_ in staged):
                r = ("staged", staged)
        if r is None:
            vols = [src.data.read_volume(u[0], c) for c in u[2]]
            r = vols[0][None] if len(vols) == 1 else np.stack(vols)  # the common one-channel unit stays in its pinned block
        if timing is not None:
            timing["load"].append(_time.perf_counter() - t0)
        return r

    def _all_nan_or_zero(x) -> bool:
        if isinstance(x, np.ndarray):
            return _check_nan_n_zeros(x)
        import torch as _torch

        if x.dtype.is_floating_point:
            return bool(_torch.isnan(x).all()) or not bool((x != 0).any())
        return not bool(x.view(_torch.int16 if x.element_size() == 2 else _torch.uint8).any())  # (integer: all zeros)

    def store(u, res):
        t0 = _time.perf_counter()
        _store(u, res)
        if timing is not None:
            timing["store"].append(_time.perf_counter() - t0)

    def _store(u, res):
        if res is not None:
            for c, vol in zip(u[3], res):
                if callable(vol):
                    vol()  # the host half of a device-resident result (encode_volume_device ran on the operator's thread)
                else:
                    dst.data.write_volume(u[1], c, vol)
        done[u[4]] = True
        _write_json(done_file, {"token": resume_token, "units": done})

    n_run = 0
    with ThreadPoolExecutor(1, thread_name_prefix="bh-pipe-r") as reader, \
            ThreadPoolExecutor(1, thread_name_prefix="bh-pipe-w") as writer:
        nxt = reader.submit(load, units[0]) if units else None
        pending = deque()
        for i, u in enumerate(units):
            t0 = _time.perf_counter()
            czyx = nxt.result()
            t1 = _time.perf_counter()
            nxt = reader.submit(load, units[i + 1]) if i + 1 < len(units) else None
            if isinstance(czyx, tuple):  # staged volumes: their device half, here
                import torch as _torch

                vols = [src.data.upload_staged(s_, dev_in) for s_ in czyx[1]]
                czyx = vols[0][None] if len(vols) == 1 else _torch.stack(vols)
                del vols
            res = None
            if not _all_nan_or_zero(czyx):
                call_kw = dict(kwargs)
                if wants_t:
                    call_kw["input_time_index"] = u[0]
                res = func(czyx, **call_kw)
                t2 = _time.perf_counter()
                if timing is not None:
                    timing["wait"].append(t1 - t0)
                    timing["func"].append(t2 - t1)
                if _is_device_tensor(res):
                    # an operator marked `device_resident` hands its result over in HBM: permutation and (lz4 stores) block
                    # codec run on the GPU, on this thread; the writer thread gets the host half
                    if res.shape[0] != len(u[3]):
                        raise ValueError(f"operator returned {res.shape[0]} channels for {len(u[3])} output channels")
                    res = [dst.data.encode_volume_device(u[1], c, res[k]) for k, c in enumerate(u[3])]
                    if timing is not None:
                        timing["encode"].append(_time.perf_counter() - t2)
                else:
                    res = np.asarray(res)
                    if res.shape[0] != len(u[3]):
                        raise ValueError(f"operator returned {res.shape[0]} channels for {len(u[3])} output channels")
                n_run += 1
            pending.append(writer.submit(store, u, res))
            while len(pending) > 2:  # bound the results held in host memory
                pending.popleft().result()
        while pending:
            pending.popleft().result()
    if timing is not None:
        import sys as _sys

        print("pipe timing " + str(output_position_path) + ": " + ", ".join(
            f"{k} " + "/".join(f"{x:.3f}" for x in v) for k, v in timing.items() if v) + " s", file=_sys.stderr, flush=True)
    if extra_metadata:
        merged = dict(dst.zattrs.get("extra_metadata", {}))
        merged.update(extra_metadata)
        dst.update_zattrs({"extra_metadata": merged})
    return n_run
