#!/usr/bin/env python3
"""Executable model of the register-stage column passes (csrc/fftconv_colw.inc): a 512-thread workgroup owns a tile of N rows
x W complex columns (N W = 16384), every thread keeps 16 rows x 2 adjacent columns in registers, the length-N column transform
is radix-16 over the top four row-index bits, radix-16 over the next four and radix-N/256 over the rest, joined by two
exchanges through the LDS tile.  Checks the index maps and the transform against numpy and prices the LDS bank conflicts of
every exchange access (MI355X_MICROARCH.md rules).  Run: python tools/colw_model.py"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))
from xw_model import XW, bank_conflicts, brev  # noqa: E402

NT = 512


class ColW:
    def __init__(self, logn, pad_rule=None):
        self.logn = logn
        self.N = 1 << logn
        self.W = 16384 // self.N            # complex columns per tile
        self.cpn = self.W // 2              # column pairs (float4) per row
        self.lowbits = logn - 8
        self.RC = 1 << self.lowbits         # radix of the last stage (1: none)
        assert self.cpn * (self.N // 16) == NT
        self.pad_rule = pad_rule or (lambda row: 0)

    # thread id -> (column pair, row id) per distribution; rows held: 16 per thread
    def rows_a(self, tid):
        cp, q = tid % self.cpn, tid // self.cpn          # q < N/16
        return cp, [r * (self.N // 16) + q for r in range(16)]

    def rows_b(self, tid):
        cp, rest = tid % self.cpn, tid // self.cpn
        nq = self.N // 256
        qq, t = rest % nq, rest // nq                    # t < 16
        return cp, [t * (self.N // 16) + r * nq + qq for r in range(16)]

    def rows_c(self, tid):
        cp, u = tid % self.cpn, tid // self.cpn          # 16 consecutive rows
        return cp, [16 * u + r for r in range(16)]

    def addr(self, row, cp):                              # float4 index in the LDS tile
        return row * self.cpn + self.pad_rule(row) + cp

    def transform_column(self, x):
        """The three stages on one column, array-index semantics (in place): returns X[brev(i)] at position i."""
        N = self.N
        a = np.array(x, dtype=np.complex128)
        # stage A: radix-16 over the top 4 bits, external twiddle w_N^(q k)
        n16 = N // 16
        for q in range(n16):
            idx = [r * n16 + q for r in range(16)]
            y = XW.dif(list(a[idx]))
            for rp in range(16):
                a[idx[rp]] = y[rp] * np.exp(-2j * np.pi * q * brev(rp, 4) / N)
        # stage B: radix-16 over the next 4 bits inside each block of N/16, external twiddle w_{N/16}^(qq k)
        nq = N // 256
        for t in range(16):
            for qq in range(nq):
                idx = [t * n16 + r * nq + qq for r in range(16)]
                y = XW.dif(list(a[idx]))
                for rp in range(16):
                    a[idx[rp]] = y[rp] * np.exp(-2j * np.pi * qq * brev(rp, 4) / n16)
        # stage C: radix-N/256 over the low bits, no external twiddle
        if self.RC > 1:
            for base in range(0, N, self.RC):
                idx = list(range(base, base + self.RC))
                y = XW.dif(list(a[idx]))
                for k in range(self.RC):
                    a[idx[k]] = y[k]
        return a


def check(logn, pad_rule=None, verbose=True):
    m = ColW(logn, pad_rule)
    rng = np.random.default_rng(logn)
    x = rng.standard_normal(m.N) + 1j * rng.standard_normal(m.N)
    got = m.transform_column(x)
    want = np.fft.fft(x)
    assert max(abs(got[i] - want[brev(i, logn)]) for i in range(m.N)) < 1e-9 * np.abs(want).max()
    for dist in (m.rows_a, m.rows_b, m.rows_c):
        seen = set()
        for tid in range(NT):
            cp, rows = dist(tid)
            for r in rows:
                seen.add((r, cp))
        assert len(seen) == m.N * m.cpn
    tot = {}
    for name, dist in (("A", m.rows_a), ("B", m.rows_b), ("C", m.rows_c)):
        for kind in ("r128", "w128"):
            c = i = 0
            for wave in range(NT // 64):
                for r in range(16):
                    addrs = []
                    for lane in range(64):
                        cp, rows = dist(wave * 64 + lane)
                        addrs.append(16 * m.addr(rows[r], cp))
                    cc, ii = bank_conflicts(addrs, kind)
                    c += cc
                    i += ii
            tot[name + " " + kind] = (c, i)
    if verbose:
        print(f"N = {m.N} W = {m.W}: transform OK; LDS cycles (actual, ideal) per tile:", tot)
    return tot


if __name__ == "__main__":
    for logn in (8, 9, 10):
        check(logn)
