"""Chunk codecs of the OME-Zarr stores on either side of the hot path.

The reference never touches chunk bytes itself: it reaches them through iohub 0.3.11 -> zarr / numcodecs 0.15.1
(reference uv.lock:1910-1911, 3160-3161), whose default chunk compressor is Blosc (c-blosc 1.21, zstd inside, bit- or
byte-shuffle) for NGFF 0.4 stores and the zarr-v3 ``blosc`` / ``zstd`` / ``sharding_indexed`` codecs for NGFF 0.5
(`biahub/settings.py:43`, `output_ome_zarr_version`).  None of those libraries is in this image, so this module
restates the published formats:

* the Blosc-1 container (16-byte header, block start table, optional per-byte-plane splits, byte shuffle, bit shuffle)
  with the zstd / lz4 / zlib / blosclz inner codecs — pinned against streams written by the real c-blosc 1.21.0
  (tests/golden/blosc_streams.npz, made by tests/golden/make_codec_golden.py);
* plain zstd / gzip / zlib / lz4 (numcodecs framing) chunk compressors;
* CRC-32C (Castagnoli), which zarr v3's sharding index carries.

The entropy coders themselves (zstd, lz4) come from pyarrow's bundled copies; zlib from the standard library.  The byte
permutations (shuffle / bit shuffle) are native code in libbhcore: HIP kernels (csrc/codec.hip: ``bh_blosc_unfilter`` /
``bh_blosc_filter``) so that a volume headed for the GPU is un-shuffled there, at HBM speed, instead of on a host core,
and the same loops as host code (``bh_host_blosc_*``) for volumes that stay on the host.  Their NumPy restatement is
oracle/codec_np.py (tests only).
"""

from __future__ import annotations

import struct
import zlib

import numpy as np

BLOSC_NOSHUFFLE, BLOSC_SHUFFLE, BLOSC_BITSHUFFLE = 0, 1, 2
_BLOSC_MIN_BUFFERSIZE = 128  # c-blosc: buffers below this are stored, blocks below this many elements are not split
_BLOSC_MAX_SPLITS = 16
_BLOSC_FORMATS = {0: "blosclz", 1: "lz4", 2: "snappy", 3: "zlib", 4: "zstd"}
_BLOSC_CNAMES = {"blosclz": 0, "lz4": 1, "lz4hc": 1, "snappy": 2, "zlib": 3, "zstd": 4}

_PA_CODECS: dict = {}


def _pa_codec(name: str, level=None):
    key = (name, level)
    if key not in _PA_CODECS:
        import pyarrow as pa

        _PA_CODECS[key] = pa.Codec(name, compression_level=level) if level is not None else pa.Codec(name)
    return _PA_CODECS[key]


def zstd_decompress(buf, nbytes: int) -> np.ndarray:
    out = _pa_codec("zstd").decompress(memoryview(buf), decompressed_size=int(nbytes))
    return np.frombuffer(out, np.uint8)


def zstd_compress(buf, level: int = 1) -> bytes:
    return _pa_codec("zstd", int(level)).compress(memoryview(buf), asbytes=True)


def lz4_block_decompress(buf, nbytes: int) -> np.ndarray:
    out = _pa_codec("lz4_raw").decompress(memoryview(buf), decompressed_size=int(nbytes))
    return np.frombuffer(out, np.uint8)


def lz4_block_compress(buf) -> bytes:
    return _pa_codec("lz4_raw").compress(memoryview(buf), asbytes=True)


def zstd_frame_content_size(buf) -> int | None:
    """Decompressed size from a zstd frame header (RFC 8878 §3.1.1.1), None when the frame does not record it."""
    b = bytes(memoryview(buf)[:18])
    if len(b) < 5 or b[:4] != b"\x28\xb5\x2f\xfd":
        raise ValueError("not a zstd frame")
    fhd = b[4]
    fcs_flag, single, did_flag = fhd >> 6, (fhd >> 5) & 1, fhd & 3
    pos = 5 + (0 if single else 1) + (0, 1, 2, 4)[did_flag]
    if fcs_flag == 0:
        return b[pos] if single else None
    if fcs_flag == 1:
        return struct.unpack_from("<H", b, pos)[0] + 256
    if fcs_flag == 2:
        return struct.unpack_from("<I", b, pos)[0]
    return struct.unpack_from("<Q", b, pos)[0]


# ---------------------------------------------------------------------------------------------------------------
# CRC-32C (Castagnoli, reflected polynomial 0x82F63B78): the checksum of zarr v3's `crc32c` codec
# ---------------------------------------------------------------------------------------------------------------
_CRC32C_TABLE = None


def _crc32c_table() -> np.ndarray:
    global _CRC32C_TABLE
    if _CRC32C_TABLE is None:
        t = np.arange(256, dtype=np.uint32)
        for _ in range(8):
            t = np.where(t & 1, (t >> 1) ^ np.uint32(0x82F63B78), t >> 1).astype(np.uint32)
        _CRC32C_TABLE = t
    return _CRC32C_TABLE


def crc32c(data, value: int = 0) -> int:
    """CRC-32C of ``data`` (bytes-like), continuing from ``value``.  Table driven, one byte per step: meant for shard
    indexes (16 bytes per inner chunk), not for bulk data."""
    table = _crc32c_table().tolist()
    crc = (~value) & 0xFFFFFFFF
    for byte in bytes(memoryview(data)):
        crc = table[(crc ^ byte) & 0xFF] ^ (crc >> 8)
    return (~crc) & 0xFFFFFFFF


# ---------------------------------------------------------------------------------------------------------------
# blosclz (c-blosc's own LZ77 coder, format version 1 — the FastLZ level-2 token stream)
# ---------------------------------------------------------------------------------------------------------------
def blosclz_decompress(src, nbytes: int) -> np.ndarray:
    """Token stream: control byte c < 32 -> c + 1 literals follow; else a match of length (c >> 5) + 2 (7 -> extended
    by following bytes, 255 continues) at distance ((c & 31) << 8 | next) + 1; distance field 8191 escapes to a 16-bit
    far distance.  A Python loop per token: rare codec (iohub writes zstd), kept for completeness."""
    ip = bytes(memoryview(src))
    n_in = len(ip)
    out = bytearray(nbytes)
    op = 0
    i = 0
    ctrl = ip[0] & 31
    i = 1
    while True:
        if ctrl >= 32:
            length = (ctrl >> 5) - 1
            ofs = (ctrl & 31) << 8
            if length == 6:
                while True:
                    code = ip[i]
                    i += 1
                    length += code
                    if code != 255:
                        break
            code = ip[i]
            i += 1
            dist = ofs + code
            if code == 255 and ofs == (31 << 8):
                dist = ((ip[i] << 8) | ip[i + 1]) + 8191
                i += 2
            length += 3
            ref = op - dist - 1
            if ref < 0 or op + length > nbytes:
                raise ValueError("corrupt blosclz stream")
            if dist + 1 >= length:
                out[op:op + length] = out[ref:ref + length]
            else:  # overlapping match: the pattern of dist + 1 bytes repeats
                pat = bytes(out[ref:op])
                reps = -(-length // len(pat))
                out[op:op + length] = (pat * reps)[:length]
            op += length
            if i < n_in:
                ctrl = ip[i]
                i += 1
            else:
                break
        else:
            run = ctrl + 1
            if op + run > nbytes or i + run > n_in:
                raise ValueError("corrupt blosclz stream")
            out[op:op + run] = ip[i:i + run]
            op += run
            i += run
            if i < n_in:
                ctrl = ip[i]
                i += 1
            else:
                break
    if op != nbytes:
        raise ValueError(f"blosclz stream decoded to {op} bytes, expected {nbytes}")
    return np.frombuffer(bytes(out), np.uint8)


# ---------------------------------------------------------------------------------------------------------------
# Blosc-1 container
# ---------------------------------------------------------------------------------------------------------------
class BloscHeader:
    __slots__ = ("version", "versionlz", "flags", "typesize", "nbytes", "blocksize", "cbytes")

    def __init__(self, buf):
        mv = memoryview(buf)
        if len(mv) < 16:
            raise ValueError("blosc stream shorter than its 16-byte header")
        self.version, self.versionlz, self.flags, self.typesize = mv[0], mv[1], mv[2], mv[3]
        self.nbytes, self.blocksize, self.cbytes = struct.unpack_from("<III", mv, 4)
        if self.cbytes > len(mv):
            raise ValueError(f"blosc stream truncated: header says {self.cbytes} bytes, have {len(mv)}")

    @property
    def memcpyed(self) -> bool:
        return bool(self.flags & 0x2)

    @property
    def shuffle_mode(self) -> int:
        if self.flags & 0x1 and self.typesize > 1:
            return BLOSC_SHUFFLE
        if self.flags & 0x4:
            return BLOSC_BITSHUFFLE
        return BLOSC_NOSHUFFLE

    @property
    def codec(self) -> str:
        return _BLOSC_FORMATS.get(self.flags >> 5, f"format {self.flags >> 5}")


def _inner_decompress(codec: str, buf, nbytes: int) -> np.ndarray:
    if codec == "zstd":
        return zstd_decompress(buf, nbytes)
    if codec == "lz4":
        return lz4_block_decompress(buf, nbytes)
    if codec == "zlib":
        return np.frombuffer(zlib.decompress(buf), np.uint8)
    if codec == "blosclz":
        return blosclz_decompress(buf, nbytes)
    raise NotImplementedError(f"blosc inner codec {codec!r} is not supported")


def blosc_decode_blocks(buf, out: np.ndarray | None = None) -> tuple[BloscHeader, np.ndarray]:
    """Entropy-decode a Blosc-1 stream WITHOUT undoing its shuffle: returns the header and the ``nbytes`` still-permuted
    bytes (block after block).  ``unfilter`` below — or ``bh_blosc_unfilter`` on the GPU — finishes the job."""
    h = BloscHeader(buf)
    mv = memoryview(buf)
    if out is None:
        out = np.empty(h.nbytes, np.uint8)
    elif out.size != h.nbytes or out.dtype != np.uint8:
        raise ValueError("output buffer does not match the stream")
    if h.nbytes == 0:
        return h, out
    if h.memcpyed:
        out[:] = np.frombuffer(mv, np.uint8, h.nbytes, 16)
        return h, out
    nblocks = -(-h.nbytes // h.blocksize)
    bstarts = struct.unpack_from(f"<{nblocks}i", mv, 16)
    dont_split = bool(h.flags & 0x10)
    codec = h.codec
    for b in range(nblocks):
        o0 = b * h.blocksize
        bsize = min(h.blocksize, h.nbytes - o0)
        leftover = bsize != h.blocksize
        split = (not dont_split and h.typesize <= _BLOSC_MAX_SPLITS
                 and h.blocksize // h.typesize >= _BLOSC_MIN_BUFFERSIZE and not leftover)
        nsplits = h.typesize if split else 1
        ne = bsize // nsplits
        pos = bstarts[b]
        for j in range(nsplits):
            (cb,) = struct.unpack_from("<i", mv, pos)
            pos += 4
            if cb < 0 or pos + cb > len(mv):
                raise ValueError("corrupt blosc stream")
            dst = out[o0 + j * ne: o0 + (j + 1) * ne]
            if cb == ne:
                dst[:] = np.frombuffer(mv, np.uint8, ne, pos)
            else:
                dst[:] = _inner_decompress(codec, mv[pos:pos + cb], ne)
            pos += cb
    return h, out


def _native_filter(name: str, src: np.ndarray, out: np.ndarray, blocksize: int, typesize: int, mode: int) -> None:
    """The per-block permutation in libbhcore's host code (csrc/codec.hip, ``bh_host_blosc_*``; releases the GIL).  Like
    every other operator of this package it needs the built library: there is no NumPy fallback (the NumPy restatement
    lives in oracle/codec_np.py and is test infrastructure)."""
    from . import _lib

    lib = _lib.load()
    src = np.ascontiguousarray(src)
    _lib.check(getattr(lib, name)(src.ctypes.data, out.ctypes.data, src.size, int(blocksize), int(typesize), int(mode)))


def unfilter(shuffled: np.ndarray, nbytes: int, blocksize: int, typesize: int, mode: int,
             out: np.ndarray | None = None) -> np.ndarray:
    """Undo the per-block permutation of a Blosc stream on the host."""
    if out is None or not out.flags.c_contiguous:
        res = np.empty(nbytes, np.uint8)
    else:
        res = out
    if mode == BLOSC_NOSHUFFLE or nbytes == 0:
        res[:] = shuffled
    else:
        _native_filter("bh_host_blosc_unfilter", shuffled, res, blocksize, typesize, mode)
    if out is not None and res is not out:
        out[:] = res
        return out
    return res


def filter_host(raw: np.ndarray, blocksize: int, typesize: int, mode: int) -> np.ndarray:
    """The per-block permutation a Blosc writer applies (inverse of ``unfilter``)."""
    raw = np.asarray(raw, np.uint8).reshape(-1)
    out = np.empty_like(raw)
    if raw.size:
        _native_filter("bh_host_blosc_filter", raw, out, max(1, blocksize), typesize, mode)
    return out


def blosc_decompress(buf, out: np.ndarray | None = None) -> np.ndarray:
    """Full Blosc-1 decode to ``nbytes`` uint8 (what ``numcodecs.Blosc.decode`` returns)."""
    h, raw = blosc_decode_blocks(buf)
    if h.memcpyed or h.shuffle_mode == BLOSC_NOSHUFFLE:
        if out is None:
            return raw
        out[:] = raw
        return out
    return unfilter(raw, h.nbytes, h.blocksize, h.typesize, h.shuffle_mode, out)


def blosc_compress(data, typesize: int, cname: str = "zstd", clevel: int = 1, shuffle_mode: int = BLOSC_BITSHUFFLE,
                   blocksize: int = 0, prefiltered: bool = False) -> bytes:
    """Write a Blosc-1 stream any c-blosc >= 1.15 reads.  Blocks are never split (flag 0x10, what c-blosc itself does
    for zstd / lz4hc / zlib).  ``blocksize == 0`` picks 256 KiB rounded to 8 elements.  ``prefiltered``: ``data`` is
    already permuted block by block (by ``bh_blosc_filter`` on the GPU) with exactly this blocksize."""
    raw = np.frombuffer(memoryview(data).cast("B"), np.uint8) if not isinstance(data, np.ndarray) else \
        np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    nbytes = raw.size
    if not 1 <= typesize <= 255:
        typesize = 1
    if cname not in _BLOSC_CNAMES or cname in ("snappy", "blosclz", "lz4hc"):
        raise NotImplementedError(f"blosc writer: inner codec {cname!r} is not supported (zstd, lz4, zlib are)")
    flags = 0x10 | (_BLOSC_CNAMES[cname] << 5)
    if shuffle_mode == BLOSC_SHUFFLE:
        flags |= 0x1
    elif shuffle_mode == BLOSC_BITSHUFFLE:
        flags |= 0x4
    if blocksize <= 0:
        blocksize = default_blocksize(typesize)
    blocksize = min(blocksize, nbytes) if nbytes else blocksize
    versionlz = 1

    def stored() -> bytes:
        if prefiltered:
            raise ValueError("a prefiltered buffer cannot be stored raw")
        return struct.pack("<BBBBIII", 2, versionlz, flags | 0x2, typesize, nbytes, blocksize, nbytes + 16) + raw.tobytes()

    if nbytes < _BLOSC_MIN_BUFFERSIZE or clevel == 0:
        if prefiltered and nbytes:
            raise ValueError("buffers below 128 bytes are stored raw: do not prefilter them")
        return stored()
    nblocks = -(-nbytes // blocksize)
    parts, bstarts = [], []
    pos = 16 + 4 * nblocks
    work = raw  # `raw` stays the caller's bytes: an incompressible buffer is stored unpermuted
    if not prefiltered and shuffle_mode != BLOSC_NOSHUFFLE:
        work = filter_host(raw, blocksize, typesize, shuffle_mode)
    for b in range(nblocks):
        blk = work[b * blocksize: min(nbytes, (b + 1) * blocksize)]
        if cname == "zstd":
            comp = zstd_compress(blk, clevel)
        elif cname == "lz4":
            comp = lz4_block_compress(blk)
        else:
            comp = zlib.compress(blk, clevel)
        if len(comp) >= blk.size:  # incompressible: stored, which the reader recognises by cbytes == block bytes
            comp = blk.tobytes()
        bstarts.append(pos)
        parts.append(struct.pack("<i", len(comp)))
        parts.append(comp)
        pos += 4 + len(comp)
    if pos >= nbytes + 16 and not prefiltered:
        return stored()
    head = struct.pack("<BBBBIII", 2, versionlz, flags, typesize, nbytes, blocksize, pos)
    return head + struct.pack(f"<{nblocks}i", *bstarts) + b"".join(parts)


def default_blocksize(typesize: int) -> int:
    unit = 8 * max(1, typesize)
    return max(unit, ((256 << 10) // unit) * unit)


# ---------------------------------------------------------------------------------------------------------------
# the permutations on the GPU (csrc/codec.hip); torch uint8 tensors on the device, distinct buffers
# ---------------------------------------------------------------------------------------------------------------
def _device_filter(fn_name: str, src, dst, blocksize: int, typesize: int, mode: int) -> None:
    import torch

    from . import _lib
    from .device import get_context, ptr

    if src.dtype != torch.uint8 or dst.dtype != torch.uint8 or not src.is_cuda or src.device != dst.device:
        raise ValueError("device (un)filter needs two uint8 tensors on the same GPU")
    if src.numel() != dst.numel() or not src.is_contiguous() or not dst.is_contiguous():
        raise ValueError("device (un)filter needs contiguous tensors of equal size")
    ctx = get_context(src.device)
    _lib.check(getattr(ctx.lib, fn_name)(ctx.handle, ptr(src), ptr(dst), src.numel(), int(blocksize), int(typesize), int(mode)))


def unfilter_device(src, dst, blocksize: int, typesize: int, mode: int) -> None:
    """dst <- the blocks of src un-shuffled (``bh_blosc_unfilter``); the device half of ``blosc_decode_blocks``."""
    _device_filter("bh_blosc_unfilter", src, dst, blocksize, typesize, mode)


def filter_device(src, dst, blocksize: int, typesize: int, mode: int) -> None:
    """dst <- src shuffled block by block (``bh_blosc_filter``), ready for ``blosc_compress(..., prefiltered=True)``."""
    _device_filter("bh_blosc_filter", src, dst, blocksize, typesize, mode)


def blosc_lz4_compress_device(filtered, nframes: int, cbytes: int, blocksize: int, typesize: int, mode: int):
    """Blosc-1 frames (inner codec lz4) of ``nframes`` chunks of ``cbytes`` bytes each, ON the GPU (``bh_blosc_lz4_compress``).
    ``filtered``: uint8 device tensor of ``nframes * cbytes`` bytes already permuted block by block (``filter_device`` with this
    ``blocksize``).  Returns ``(packed, offsets)``: the frames in one uint8 device tensor and a host list of ``nframes + 1``
    offsets (frame f = ``packed[offsets[f]: ...]``, its own length in its header; the last entry is the total) — only the
    compressed bytes need to cross PCIe.  Synchronises once (the offsets are read back)."""
    import ctypes

    import torch

    from . import _lib
    from .device import get_context, ptr

    if filtered.dtype != torch.uint8 or not filtered.is_cuda or not filtered.is_contiguous() or filtered.numel() != nframes * cbytes:
        raise ValueError("need a contiguous uint8 device tensor of nframes * cbytes bytes")
    if cbytes < _BLOSC_MIN_BUFFERSIZE:
        raise ValueError("chunks below 128 bytes are stored raw by Blosc: compress them on the host")
    ctx = get_context(filtered.device)
    bound = int(ctx.lib.bh_blosc_lz4_bound(int(nframes), int(cbytes), int(blocksize)))
    packed = torch.empty(bound, dtype=torch.uint8, device=filtered.device)
    foff = torch.empty(nframes + 1, dtype=torch.int64, device=filtered.device)
    with torch.cuda.device(filtered.device):
        _lib.check(ctx.lib.bh_blosc_lz4_compress(ctx.handle, ptr(filtered), int(nframes), int(cbytes), int(blocksize), int(typesize),
                                                 int(mode), ptr(packed), ptr(foff)))
    return packed, [int(v) for v in foff.cpu().tolist()]


def blosc_lz4_stream_table(buf) -> tuple["BloscHeader", np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """The LZ4 streams of a Blosc-1 frame with the lz4 inner codec: ``(header, soff, csize, doff, dlen)`` — stream i is
    ``csize[i]`` bytes at ``buf[soff[i]:]`` and decodes to ``dlen[i]`` still-permuted bytes at offset ``doff[i]`` of the chunk
    (a block is one stream, or ``typesize`` streams when its writer split it).  Host-side parsing only."""
    h = BloscHeader(buf)
    if h.codec != "lz4" or h.memcpyed:
        raise ValueError("not a compressed Blosc frame with the lz4 inner codec")
    mv = memoryview(buf)
    nblocks = -(-h.nbytes // h.blocksize)
    bstarts = struct.unpack_from(f"<{nblocks}i", mv, 16)
    dont_split = bool(h.flags & 0x10)
    soff, csize, doff, dlen = [], [], [], []
    for b in range(nblocks):
        o0 = b * h.blocksize
        bsize = min(h.blocksize, h.nbytes - o0)
        split = (not dont_split and h.typesize <= _BLOSC_MAX_SPLITS and h.blocksize // h.typesize >= _BLOSC_MIN_BUFFERSIZE
                 and bsize == h.blocksize)
        nsplits = h.typesize if split else 1
        ne = bsize // nsplits
        pos = bstarts[b]
        for j in range(nsplits):
            (cb,) = struct.unpack_from("<i", mv, pos)
            pos += 4
            if cb < 0 or pos + cb > len(mv):
                raise ValueError("corrupt blosc stream")
            soff.append(pos)
            csize.append(cb)
            doff.append(o0 + j * ne)
            dlen.append(ne)
            pos += cb
    return h, np.asarray(soff, np.uint64), np.asarray(csize, np.uint32), np.asarray(doff, np.uint64), np.asarray(dlen, np.uint32)


def blosc_lz4_decode_blocks_device(frame, out_dev, device=None) -> "BloscHeader":
    """The device half of reading a Blosc-lz4 frame: ``frame`` (bytes-like, the whole frame as read from the store) is uploaded
    COMPRESSED and its LZ4 streams are decoded by ``bh_lz4_decompress_streams`` into ``out_dev`` (uint8 device tensor of
    ``header.nbytes`` still-permuted bytes; ``unfilter_device`` finishes).  Raises on a corrupt stream."""
    import torch

    from . import _lib
    from .device import get_context, ptr

    h, soff, csize, doff, dlen = blosc_lz4_stream_table(frame)
    if out_dev.dtype != torch.uint8 or not out_dev.is_cuda or out_dev.numel() != h.nbytes or not out_dev.is_contiguous():
        raise ValueError("output must be a contiguous uint8 device tensor of header.nbytes bytes")
    dev = out_dev.device
    fr = torch.frombuffer(bytearray(memoryview(frame)[: h.cbytes]), dtype=torch.uint8).to(dev)
    tabs = [torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a.view(np.int32)).to(dev) for a in (soff, csize, doff, dlen)]
    ctx = get_context(dev)
    with torch.cuda.device(dev):
        _lib.check(ctx.lib.bh_lz4_decompress_streams(ctx.handle, ptr(fr), ptr(tabs[0]), ptr(tabs[1]), ptr(tabs[2]), ptr(tabs[3]),
                                                     int(len(csize)), ptr(out_dev)))
    return h


def blosc_lz4_decode_frames_device(frames, out_dev, out_offsets) -> list["BloscHeader"]:
    """Several Blosc-lz4 frames in ONE upload and ONE launch of ``bh_lz4_decompress_streams``: frame ``k`` decodes to
    ``out_dev[out_offsets[k]: out_offsets[k] + header.nbytes]`` (uint8 device tensor, still-permuted bytes).  A volume's chunks
    decoded one by one keep a few hundred wavefronts busy each and synchronise per chunk; together they fill the device."""
    import torch

    from . import _lib
    from .device import get_context, ptr

    if out_dev.dtype != torch.uint8 or not out_dev.is_cuda or not out_dev.is_contiguous():
        raise ValueError("output must be a contiguous uint8 device tensor")
    heads, tabs, pos = [], [[], [], [], []], 0
    sizes = []
    for fr, o in zip(frames, out_offsets):
        h, soff, csize, doff, dlen = blosc_lz4_stream_table(fr)
        if o < 0 or o + h.nbytes > out_dev.numel():
            raise ValueError("frame does not fit the output tensor")
        heads.append(h)
        tabs[0].append(soff + np.uint64(pos))
        tabs[1].append(csize)
        tabs[2].append(doff + np.uint64(o))
        tabs[3].append(dlen)
        sizes.append(h.cbytes)
        pos += (h.cbytes + 15) & ~15
    if not heads:
        return heads
    host = torch.empty(pos, dtype=torch.uint8, pin_memory=True)
    hv, q = host.numpy(), 0
    for fr, nb in zip(frames, sizes):
        hv[q:q + nb] = np.frombuffer(memoryview(fr)[:nb], np.uint8)
        q += (nb + 15) & ~15
    dev = out_dev.device
    src = host.to(dev, non_blocking=True)
    cat = [np.concatenate(t) for t in tabs]
    dtabs = [torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a.view(np.int32)).to(dev) for a in cat]
    ctx = get_context(dev)
    with torch.cuda.device(dev):
        _lib.check(ctx.lib.bh_lz4_decompress_streams(ctx.handle, ptr(src), ptr(dtabs[0]), ptr(dtabs[1]), ptr(dtabs[2]), ptr(dtabs[3]),
                                                     int(len(cat[1])), ptr(out_dev)))
    return heads


# ---------------------------------------------------------------------------------------------------------------
# zarr compressor configurations -> (decode, encode)
# ---------------------------------------------------------------------------------------------------------------
_SHUFFLE_NAMES = {"noshuffle": BLOSC_NOSHUFFLE, "shuffle": BLOSC_SHUFFLE, "bitshuffle": BLOSC_BITSHUFFLE}


class ChunkCodec:
    """bytes <-> bytes stage of a chunk: ``decode(buf, nbytes) -> uint8 array``, ``encode(uint8 array) -> bytes``."""

    def __init__(self, kind: str, **cfg):
        self.kind = kind
        self.cfg = cfg

    def decode(self, buf, nbytes: int) -> np.ndarray:
        k = self.kind
        if k == "blosc":
            return blosc_decompress(buf)
        if k == "zstd":
            n = zstd_frame_content_size(buf)
            return zstd_decompress(buf, nbytes if n is None else n)
        if k in ("zlib", "gzip"):
            return np.frombuffer(zlib.decompress(buf, 47), np.uint8)  # 47: zlib or gzip wrapper, auto-detected
        if k == "lz4":  # numcodecs LZ4: int32 little-endian decompressed size, then one raw block
            (n,) = struct.unpack_from("<i", memoryview(buf), 0)
            return lz4_block_decompress(memoryview(buf)[4:], n)
        if k == "crc32c":
            mv = memoryview(buf)
            (want,) = struct.unpack_from("<I", mv, len(mv) - 4)
            if crc32c(mv[:-4]) != want:
                raise ValueError("crc32c mismatch")
            return np.frombuffer(mv[:-4], np.uint8)
        raise NotImplementedError(f"chunk codec {k!r}")

    def encode(self, raw: np.ndarray) -> bytes:
        k, c = self.kind, self.cfg
        if k == "blosc":
            return blosc_compress(raw, c.get("typesize", 1), c.get("cname", "zstd"), c.get("clevel", 1),
                                  c.get("shuffle", BLOSC_BITSHUFFLE), c.get("blocksize", 0))
        if k == "zstd":
            return zstd_compress(raw, c.get("level", 1))
        if k == "zlib":
            return zlib.compress(raw, c.get("level", 1))
        if k == "gzip":
            co = zlib.compressobj(c.get("level", 1), zlib.DEFLATED, 31)
            return co.compress(raw) + co.flush()
        if k == "lz4":
            return struct.pack("<i", raw.size) + lz4_block_compress(raw)
        if k == "crc32c":
            return bytes(raw) + struct.pack("<I", crc32c(raw))
        raise NotImplementedError(f"chunk codec {k!r}")


def codec_from_v2(comp: dict | None, itemsize: int) -> ChunkCodec | None:
    """zarr v2 ``compressor`` entry (numcodecs configuration) -> codec."""
    if comp is None:
        return None
    cid = comp.get("id")
    if cid == "blosc":
        sh = comp.get("shuffle", 1)
        if sh == -1:  # numcodecs AUTOSHUFFLE
            sh = BLOSC_BITSHUFFLE if itemsize == 1 else BLOSC_SHUFFLE
        return ChunkCodec("blosc", cname=comp.get("cname", "lz4"), clevel=comp.get("clevel", 5), shuffle=int(sh),
                          blocksize=comp.get("blocksize", 0), typesize=itemsize)
    if cid in ("zstd", "zlib", "gzip"):
        return ChunkCodec(cid, level=comp.get("level", 1))
    if cid == "lz4":
        return ChunkCodec("lz4")
    raise NotImplementedError(f"zarr v2 compressor {cid!r} is not supported (blosc, zstd, zlib, gzip, lz4 are)")


def codec_from_v3(entry: dict, itemsize: int) -> ChunkCodec:
    """One bytes->bytes entry of a zarr v3 ``codecs`` list."""
    name, cfg = entry.get("name"), entry.get("configuration", {}) or {}
    if name == "blosc":
        sh = cfg.get("shuffle", "noshuffle")
        return ChunkCodec("blosc", cname=cfg.get("cname", "zstd"), clevel=cfg.get("clevel", 5),
                          shuffle=_SHUFFLE_NAMES[sh] if isinstance(sh, str) else int(sh),
                          blocksize=cfg.get("blocksize", 0), typesize=cfg.get("typesize", itemsize))
    if name == "zstd":
        return ChunkCodec("zstd", level=cfg.get("level", 0) or 1)
    if name == "gzip":
        return ChunkCodec("gzip", level=cfg.get("level", 1))
    if name == "crc32c":
        return ChunkCodec("crc32c")
    raise NotImplementedError(f"zarr v3 codec {name!r} is not supported (bytes, blosc, zstd, gzip, crc32c, sharding_indexed are)")
