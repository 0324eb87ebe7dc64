"""Minimal OME-Zarr (NGFF 0.4, zarr v2 directory store) reader/writer and the per-position driver.

iohub / zarr / numcodecs are not available in this image, and the reference reaches them only through
``open_ome_zarr``, ``create_empty_plate`` and ``process_single_position`` (biahub/deskew.py:608-640,738-749).
This module provides those three entry points with the argument names the reference uses, over plain files:
HCS layout ``plate.zarr/<row>/<col>/<fov>/0`` with one 5-D ``(T,C,Z,Y,X)`` array per position, chunks
``(1,1,zc,Y,X)``, "/" dimension separator, uncompressed or stdlib-zlib chunks.  (blosc/zstd stores written by
iohub need numcodecs and are refused with a clear error — the codec pipeline is I/O, out of scope here.)
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
    tmp = path.with_suffix(path.suffix + ".tmp")
    tmp.write_text(json.dumps(obj, indent=1))
    os.replace(tmp, path)


_IO_POOL = None


def _io_pool():
    """Shared pool for chunk I/O (file reads/writes and zlib release the GIL); size via BH_IO_THREADS (default 8)."""
    global _IO_POOL
    if _IO_POOL is None:
        from concurrent.futures import ThreadPoolExecutor

        _IO_POOL = ThreadPoolExecutor(max_workers=max(1, int(os.environ.get("BH_IO_THREADS", "8"))),
                                      thread_name_prefix="bh-io")
    return _IO_POOL


def _io_map(fn, items):
    items = list(items)
    if len(items) <= 1 or threading.current_thread().name.startswith("bh-io"):
        for it in items:  # already on a pool thread (read-ahead / write-behind): do not nest
            fn(it)
        return
    for r in _io_pool().map(fn, items):
        pass


_TORCH_DT = None


def _host_volume(shape, dtype) -> np.ndarray:
    """Destination of a volume read: pinned host memory from torch's caching host allocator when a GPU is present (the
    upload that follows is then a direct DMA, and a plate's equally shaped volumes reuse the blocks), plain numpy else."""
    global _TORCH_DT
    dtype = np.dtype(dtype)
    if int(np.prod(shape)) * dtype.itemsize >= (8 << 20):
        try:
            import torch

            if torch.cuda.is_available():
                if _TORCH_DT is None:
                    _TORCH_DT = {np.dtype(np.uint8): torch.uint8, np.dtype(np.uint16): torch.uint16,
                                 np.dtype(np.int16): torch.int16, np.dtype(np.float32): torch.float32,
                                 np.dtype(np.float64): torch.float64, np.dtype(np.int32): torch.int32}
                if dtype in _TORCH_DT:
                    return torch.empty(tuple(int(v) for v in shape), dtype=_TORCH_DT[dtype], pin_memory=True).numpy()
        except Exception:  # no torch / no pinned memory: pageable is always correct
            pass
    return np.empty(shape, dtype=dtype)


class ZarrArray:
    """5-D zarr v2 array, whole (t, c) volumes in and out."""

    def __init__(self, path: Path):
        self.path = Path(path)
        meta = json.loads((self.path / ".zarray").read_text())
        if meta.get("zarr_format") != 2:
            raise ValueError(f"{path}: only zarr v2 arrays are supported")
        comp = meta.get("compressor")
        if comp is not None and comp.get("id") != "zlib":
            raise NotImplementedError(
                f"{path}: compressor {comp.get('id')!r} needs numcodecs; this reader handles uncompressed and zlib chunks"
            )
        if meta.get("filters"):
            raise NotImplementedError(f"{path}: zarr filters are not supported")
        self.shape = tuple(meta["shape"])
        self.chunks = tuple(meta["chunks"])
        self.dtype = np.dtype(meta["dtype"])
        self.fill_value = meta.get("fill_value", 0) or 0
        self.compressor = comp
        self.sep = meta.get("dimension_separator", ".")
        if len(self.shape) != 5 or self.chunks[0] != 1 or self.chunks[1] != 1 or self.chunks[3:] != self.shape[3:]:
            raise NotImplementedError(f"{path}: expected a (T,C,Z,Y,X) array chunked (1,1,zc,Y,X), got {self.chunks}")

    def _chunk_path(self, t, c, zi) -> Path:
        return self.path / self.sep.join(str(v) for v in (t, c, zi, 0, 0))

    def read_volume(self, t: int, c: int) -> np.ndarray:
        """One (t, c) volume; its z-chunks are read (and inflated) concurrently, straight into the result."""
        T, C, Z, Y, X = self.shape
        zc = self.chunks[2]
        out = _host_volume((Z, Y, X), self.dtype)

        def one(zi):
            f = self._chunk_path(t, c, zi)
            z0, z1 = zi * zc, min(Z, (zi + 1) * zc)
            try:
                fh = open(f, "rb")
            except FileNotFoundError:
                out[z0:z1] = self.fill_value
                return
            with fh:
                if self.compressor is None and z1 - z0 == zc:
                    view = memoryview(out[z0:z1]).cast("B")
                    if fh.readinto(view) != len(view):
                        raise OSError(f"{f}: truncated chunk")
                    return
                raw = fh.read()
            if self.compressor is not None:
                raw = zlib.decompress(raw)  # releases the GIL
            out[z0:z1] = np.frombuffer(raw, dtype=self.dtype).reshape(zc, Y, X)[: z1 - z0]

        _io_map(one, range(-(-Z // zc)))
        return out

    def write_volume(self, t: int, c: int, vol: np.ndarray) -> None:
        T, C, Z, Y, X = self.shape
        if vol.shape != (Z, Y, X):
            raise ValueError(f"volume shape {vol.shape} does not match array {(Z, Y, X)}")
        zc = self.chunks[2]
        vol = np.ascontiguousarray(vol, dtype=self.dtype)

        def one(zi):
            z0, z1 = zi * zc, min(Z, (zi + 1) * zc)
            chunk = vol[z0:z1]
            if z1 - z0 < zc:  # zarr stores full chunks
                pad = np.full((zc - (z1 - z0), Y, X), self.fill_value, dtype=self.dtype)
                chunk = np.concatenate([chunk, pad])
            raw = memoryview(chunk).cast("B")
            if self.compressor is not None:
                raw = zlib.compress(raw, self.compressor.get("level", 1))
            f = self._chunk_path(t, c, zi)
            f.parent.mkdir(parents=True, exist_ok=True)
            tmp = f.with_name(f.name + ".tmp")
            with open(tmp, "wb") as fh:
                fh.write(raw)
            os.replace(tmp, f)

        _io_map(one, range(-(-Z // zc)))

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


class Position:
    """One FOV group: ``<store>/<row>/<col>/<fov>`` with array "0" (what ``open_ome_zarr(position_path)`` yields)."""

    def __init__(self, path):
        self.path = Path(path)
        if not (self.path / ".zgroup").exists():
            raise FileNotFoundError(f"{path} is not a zarr group")
        self.zattrs = json.loads((self.path / ".zattrs").read_text()) if (self.path / ".zattrs").exists() else {}
        self.data = ZarrArray(self.path / "0")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def __getitem__(self, name):
        if str(name) != "0":
            raise KeyError(name)
        return self.data

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
        return str(self.zattrs.get("multiscales", [{}])[0].get("version", "0.4"))

    def update_zattrs(self, extra: dict) -> None:
        self.zattrs.update(extra)
        _write_json(self.path / ".zattrs", self.zattrs)


def open_ome_zarr(path, mode: str = "r", layout: str = "auto") -> Position:
    """Open one position (``layout="fov"`` and HCS positions look the same on disk below the FOV group)."""
    return Position(path)


def read_fov_array(path) -> ZarrArray:
    return Position(path).data


def _position_zattrs(channel_names, scale, version, extra=None):
    z = {
        "multiscales": [{
            "version": version,
            "axes": _AXES,
            "datasets": [{"path": "0", "coordinateTransformations": [{"type": "scale", "scale": [float(s) for s in scale]}]}],
            "name": "0",
        }],
        "omero": {"version": version, "channels": [{"label": n, "active": True, "color": "FFFFFF",
                                                   "window": {"start": 0, "end": 65535, "min": 0, "max": 65535}}
                                                  for n in channel_names]},
    }
    if extra:
        z.update(extra)
    return z


def create_empty_position(path, channel_names, shape, chunks=None, scale=(1, 1, 1, 1, 1), dtype=np.float32,
                          version="0.4", compressor=None, metadata=None) -> None:
    """Idempotent: an existing position with the same shape is left alone (reference: deskew.py:608-610)."""
    path = Path(path)
    T, C, Z, Y, X = (int(s) for s in shape)
    if len(channel_names) != C:
        raise ValueError(f"{len(channel_names)} channel names for C={C}")
    if (path / "0" / ".zarray").exists():
        if tuple(json.loads((path / "0" / ".zarray").read_text())["shape"]) == (T, C, Z, Y, X):
            return
        raise ValueError(f"{path} exists with a different shape")
    if chunks is None:
        zc = max(1, min(Z, (64 << 20) // max(1, Y * X * np.dtype(dtype).itemsize)))
        chunks = (1, 1, zc, Y, X)
    (path / "0").mkdir(parents=True, exist_ok=True)
    _write_json(path / ".zgroup", {"zarr_format": 2})
    _write_json(path / ".zattrs", _position_zattrs(channel_names, scale, version, metadata))
    _write_json(path / "0" / ".zarray", {
        "zarr_format": 2, "shape": [T, C, Z, Y, X], "chunks": [int(c) for c in chunks],
        "dtype": np.dtype(dtype).str, "compressor": compressor, "fill_value": 0, "filters": None, "order": "C",
        "dimension_separator": "/"})


def create_empty_plate(store_path, position_keys, channel_names, shape, chunks=None, scale=(1, 1, 1, 1, 1),
                       dtype=np.float32, version="0.4", compressor=None, metadata=None) -> None:
    """HCS plate with empty positions — argument names follow iohub's ``create_empty_plate`` as the reference calls
    it (biahub/deskew.py:629-640, register.py:488-504)."""
    store = Path(store_path)
    store.mkdir(parents=True, exist_ok=True)
    _write_json(store / ".zgroup", {"zarr_format": 2})
    attrs = json.loads((store / ".zattrs").read_text()) if (store / ".zattrs").exists() else {}
    plate = attrs.get("plate", {"version": version, "rows": [], "columns": [], "wells": []})
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
        (store / row).mkdir(exist_ok=True)
        _write_json(store / row / ".zgroup", {"zarr_format": 2})
        well = store / row / col
        well.mkdir(exist_ok=True)
        _write_json(well / ".zgroup", {"zarr_format": 2})
        wattrs = json.loads((well / ".zattrs").read_text()) if (well / ".zattrs").exists() else {"well": {"version": version, "images": []}}
        if not any(i["path"] == fov for i in wattrs["well"]["images"]):
            wattrs["well"]["images"].append({"path": fov})
        _write_json(well / ".zattrs", wattrs)
        create_empty_position(well / fov, channel_names, shape, chunks, scale, dtype, version, compressor, metadata)
    attrs["plate"] = plate
    _write_json(store / ".zattrs", attrs)


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

    def load(u):
        vols = [src.data.read_volume(u[0], c) for c in u[2]]
        return vols[0][None] if len(vols) == 1 else np.stack(vols)  # the common one-channel unit stays in its pinned block

    def store(u, res):
        if res is not None:
            for c, vol in zip(u[3], res):
                dst.data.write_volume(u[1], c, vol)
        done[u[4]] = True
        _write_json(done_file, {"token": resume_token, "units": done})

    n_run = 0
    with ThreadPoolExecutor(1, thread_name_prefix="bh-pipe-r") as reader, \
            ThreadPoolExecutor(1, thread_name_prefix="bh-pipe-w") as writer:
        nxt = reader.submit(load, units[0]) if units else None
        pending = deque()
        for i, u in enumerate(units):
            czyx = nxt.result()
            nxt = reader.submit(load, units[i + 1]) if i + 1 < len(units) else None
            res = None
            if not _check_nan_n_zeros(czyx):
                call_kw = dict(kwargs)
                if wants_t:
                    call_kw["input_time_index"] = u[0]
                res = np.asarray(func(czyx, **call_kw))
                if res.shape[0] != len(u[3]):
                    raise ValueError(f"operator returned {res.shape[0]} channels for {len(u[3])} output channels")
                n_run += 1
            pending.append(writer.submit(store, u, res))
            while len(pending) > 2:  # bound the results held in host memory
                pending.popleft().result()
        while pending:
            pending.popleft().result()
    if extra_metadata:
        merged = dict(dst.zattrs.get("extra_metadata", {}))
        merged.update(extra_metadata)
        dst.update_zattrs({"extra_metadata": merged})
    return n_run
