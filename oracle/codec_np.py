"""NumPy restatement of the Blosc-1 byte permutations — TEST INFRASTRUCTURE (oracle), not the product path.

Follows c-blosc 1.21 (the library numcodecs 0.15.1 wraps; reference uv.lock:3160-3161): shuffle.c `shuffle` / `unshuffle`
/ `bitshuffle` / `bitunshuffle`, shuffle-generic.h, bitshuffle-generic.c (`bshuf_trans_bit_elem_scal`).  Pinned by the
streams the real library wrote (tests/golden/blosc_streams.npz): tests/test_codecs.py un-permutes every one of them with
these functions.  The product (biahub_amd/codecs.py, csrc/codec.hip) is checked against this file; only tests import it.
"""

from __future__ import annotations

import numpy as np

BLOSC_NOSHUFFLE, BLOSC_SHUFFLE, BLOSC_BITSHUFFLE = 0, 1, 2


def shuffle(block: np.ndarray, typesize: int) -> np.ndarray:
    """Byte shuffle of one block: byte j of element i goes to plane j; the ``len % typesize`` tail is copied."""
    block = np.asarray(block, np.uint8)
    n = block.size // typesize
    if typesize <= 1 or n == 0:
        return block.copy()
    out = np.empty_like(block)
    out[: n * typesize] = block[: n * typesize].reshape(n, typesize).T.reshape(-1)
    out[n * typesize:] = block[n * typesize:]
    return out


def unshuffle(block: np.ndarray, typesize: int) -> np.ndarray:
    block = np.asarray(block, np.uint8)
    n = block.size // typesize
    if typesize <= 1 or n == 0:
        return block.copy()
    out = np.empty_like(block)
    out[: n * typesize] = block[: n * typesize].reshape(typesize, n).T.reshape(-1)
    out[n * typesize:] = block[n * typesize:]
    return out


def bitshuffle(block: np.ndarray, typesize: int) -> np.ndarray:
    """Bit shuffle of one block: bit k of byte j of element i goes to bit-plane 8 j + k, element i at bit i % 8 (LSB
    first) of byte i // 8 of the plane.  c-blosc 1.x applies it only when the block holds a multiple of 8 elements
    and otherwise stores the block unpermuted; a ``len % typesize`` tail is copied."""
    block = np.asarray(block, np.uint8)
    n = block.size // typesize
    if n == 0 or n % 8:
        return block.copy()
    out = np.empty_like(block)
    bits = np.unpackbits(block[: n * typesize].reshape(n, typesize), axis=1, bitorder="little")  # (n, 8 ts)
    out[: n * typesize] = np.packbits(bits.T, axis=1, bitorder="little").reshape(-1)            # (8 ts, n / 8)
    out[n * typesize:] = block[n * typesize:]
    return out


def bitunshuffle(block: np.ndarray, typesize: int) -> np.ndarray:
    block = np.asarray(block, np.uint8)
    n = block.size // typesize
    if n == 0 or n % 8:
        return block.copy()
    out = np.empty_like(block)
    planes = np.unpackbits(block[: n * typesize].reshape(8 * typesize, n // 8), axis=1, bitorder="little")  # (8 ts, n)
    out[: n * typesize] = np.packbits(planes.T, axis=1, bitorder="little").reshape(-1)                      # (n, ts)
    out[n * typesize:] = block[n * typesize:]
    return out


def unfilter(shuffled: np.ndarray, nbytes: int, blocksize: int, typesize: int, mode: int) -> np.ndarray:
    """Undo the per-block permutation of a Blosc stream (c-blosc blosc.c `blosc_d`, the part after the entropy decoder)."""
    shuffled = np.asarray(shuffled, np.uint8).reshape(-1)
    out = np.empty(nbytes, np.uint8)
    for o0 in range(0, nbytes, max(1, blocksize)):
        blk = shuffled[o0:o0 + blocksize]
        if mode == BLOSC_SHUFFLE and typesize > 1:
            blk = unshuffle(blk, typesize)
        elif mode == BLOSC_BITSHUFFLE and blk.size >= typesize:
            blk = bitunshuffle(blk, typesize)
        out[o0:o0 + blk.size] = blk
    return out


def filter_blocks(raw: np.ndarray, blocksize: int, typesize: int, mode: int) -> np.ndarray:
    """The per-block permutation a Blosc writer applies (c-blosc blosc.c `blosc_c`, the part before the entropy coder)."""
    raw = np.asarray(raw, np.uint8).reshape(-1)
    out = np.empty_like(raw)
    for o0 in range(0, raw.size, max(1, blocksize)):
        blk = raw[o0:o0 + blocksize]
        if mode == BLOSC_SHUFFLE and typesize > 1:
            blk = shuffle(blk, typesize)
        elif mode == BLOSC_BITSHUFFLE and blk.size >= typesize:
            blk = bitshuffle(blk, typesize)
        out[o0:o0 + blk.size] = blk
    return out
