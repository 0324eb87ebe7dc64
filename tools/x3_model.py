#!/usr/bin/env python3
"""Executable model of the radix-3 wave-private X pass (csrc/fftconv_x3.inc): rows of 1536 or 3072 voxels, i.e. packed complex
transforms of M = 3 L points, L = 256 or 512.

One wavefront owns a row pair; LG = L / 8 lanes (32 or 64) hold a row as 3 thirds x 8 registers.  The transform is a
decimation-in-frequency radix-3 step across the thirds (elements m, m + L, m + 2 L sit in one lane), then three in-register
stages per third — radix-8 over the top three index bits, radix-8 (L = 512) or radix-4 (L = 256) over the middle bits, radix-8
over the low three — joined by two exchanges through a wave-private LDS buffer.  Third t then holds the frequencies 3 j + t at
the bit-reversed position of j (the order the tile kernels of fftconv_xpass.inc leave such rows in).  At the spectrum end a
lane holds ONE 8-point group of each third; the untangle of the packed real transform needs the mirrored frequency M - f:
thirds 1 and 2 mirror into each other (position p of third 1 <-> position L - 1 - p of third 2), so a lane that holds group g
of third 1 holds group LG - 1 - g of third 2; third 0 mirrors into itself, and its groups are dealt out so that the mirrored
group sits in the NEIGHBOURING lane (lane ^ 1: one DPP quad-permute per register), lanes 0 and 1 holding the two self-mirrored
groups 0 and 1.

The model works lane by lane and register by register with the kernel's index maps, LDS addresses and twiddle tables, and is
checked against numpy's FFT; `bank_conflicts` (tools/xw_model.py) prices each exchange.  Run: python tools/x3_model.py
"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))
from xw_model import bank_conflicts, brev  # noqa: E402


class X3:
    def __init__(self, logl, swz=None, g1=None):
        self.logl = logl
        self.L = 1 << logl
        self.M = 3 * self.L
        self.lg = self.L // 8            # lanes per row pair
        self.blk = self.L // 8           # elements per top-3-bit block ( == lg: lane l <-> element l of a block)
        self.midbits = logl - 6          # 3 (L = 512) or 2 (L = 256)
        self.R2 = 1 << self.midbits
        self.nrep = 8 // self.R2         # middle-stage groups per lane and third (1 or 2)
        self.pad = 8                     # complex words of padding per block
        self.bstride = self.blk + self.pad
        self.rowwords = 8 * self.bstride  # LDS words of one third of one row
        self._swz = swz or (self.swz_l256 if logl == 8 else self.swz_l512)
        self._g1 = g1 or (lambda l: l)

    # ---- array index (within a third) <-> (lane, reg) in the three distributions ----
    def da(self, l, rho):                # real side / stage A: registers = top three bits
        return rho * self.blk + l

    def db(self, l, rho):                # stage B: registers = middle bits (+ the top bit when R2 = 4)
        n0 = l & 7
        lh = l >> 3                      # the top-3 bits that are not in registers
        mid = rho & (self.R2 - 1)
        xh = rho >> self.midbits
        nx = 3 - self.midbits            # top bits held in registers: 0 (L = 512) or 1 (L = 256)
        top = (xh << (3 - nx)) | lh
        return top * self.blk + mid * 8 + n0

    # lane -> 8-point group of each third at the spectrum end
    @staticmethod
    def mirror_group(g):
        if g < 2:
            return g
        v = g.bit_length() - 1
        return 3 * (1 << v) - 1 - g

    def group0(self, l):                 # third 0: (g, mirror(g)) in lanes (2k, 2k + 1), k >= 1; groups 0, 1 in lanes 0, 1
        if l < 2:
            return l
        k = l >> 1
        v = k.bit_length()               # floor(log2 k) + 1
        low = k - (1 << (v - 1))
        g = (1 << v) + low
        return g if (l & 1) == 0 else (2 << v) - 1 - low

    def group(self, t, l):
        if t == 0:
            return self.group0(l)
        g1 = self._g1(l)
        return g1 if t == 1 else self.lg - 1 - g1

    def dc(self, t, l, n):
        return 8 * self.group(t, l) + n

    # ---- LDS address (complex words, within one third's buffer) of array index i ----
    # XOR swizzle of the four 16-B chunks of an 8-point group: two parities of the group index, found per L by exhaustive
    # search over the mask pairs with `lds_cycles` (every DA / DB access conflict-free; DC within 1.0-1.5x of conflict-free)
    @staticmethod
    def swz_l256(g):
        return (g & 1) | (((g >> 1) & 1) << 1)

    @staticmethod
    def swz_l512(g):
        par = lambda v: bin(v).count("1") & 1
        return par(g & 2) | (par(g & 20) << 1)

    def addr(self, i):
        blk, w = divmod(i, self.blk)
        g, n = divmod(w, 8)
        gg = blk * (self.blk // 8) + g
        return blk * self.bstride + g * 8 + 2 * ((n >> 1) ^ self._swz(gg)) + (n & 1)

    # ---- stored spectrum column of position q = t L + p (what xw_col holds) ----
    def col_of_pos(self, q):
        t, p = divmod(q, self.L)
        g, n = divmod(p, 8)
        lane = [l for l in range(self.lg) if self.group(t, l) == g][0]
        return 2 * ((4 * t + (n >> 1)) * self.lg + lane) + (n & 1)

    @staticmethod
    def dif(x):
        n = len(x)
        x = list(x)
        h = n // 2
        while h >= 1:
            for i in range(n):
                if not i & h:
                    a, b = x[i], x[i + h]
                    j = i % h
                    x[i], x[i + h] = a + b, (a - b) * np.exp(-2j * np.pi * j / (2 * h))
            h //= 2
        return x

    @staticmethod
    def dit(x):
        n = len(x)
        x = list(x)
        h = 1
        while h < n:
            for i in range(n):
                if not i & h:
                    j = i % h
                    a, b = x[i], x[i + h] * np.exp(+2j * np.pi * j / (2 * h))
                    x[i], x[i + h] = a + b, a - b
            h *= 2
        return x

    # ---- forward: regs[l][t][rho] in DA (element t L + da(l, rho)) -> regs[l][t][n] in DC ----
    def forward(self, regs):
        L, M, lg, blk = self.L, self.M, self.lg, self.blk
        w3 = np.exp(-2j * np.pi / 3)
        # radix-3 across the thirds + twiddle w_M^(k m)
        for l in range(lg):
            for rho in range(8):
                m = self.da(l, rho)
                a, b, c = regs[l][0][rho], regs[l][1][rho], regs[l][2][rho]
                regs[l][0][rho] = a + b + c
                regs[l][1][rho] = (a + w3 * b + w3 ** 2 * c) * np.exp(-2j * np.pi * m / M)
                regs[l][2][rho] = (a + w3 ** 2 * b + w3 ** 4 * c) * np.exp(-2j * np.pi * 2 * m / M)
        lds = [np.zeros(self.rowwords, dtype=np.complex128) for _ in range(3)]
        for t in range(3):
            # stage A: radix-8 over the top three bits; twiddle w_L^(l k), k = brev3(register)
            for l in range(lg):
                y = self.dif(regs[l][t])
                regs[l][t] = [y[rp] * np.exp(-2j * np.pi * l * brev(rp, 3) / L) for rp in range(8)]
            for l in range(lg):
                for rho in range(8):
                    lds[t][self.addr(self.da(l, rho))] = regs[l][t][rho]
            for l in range(lg):
                regs[l][t] = [lds[t][self.addr(self.db(l, rho))] for rho in range(8)]
            # stage B: radix-R2 over the middle bits; twiddle w_BLK^(n0 k1)
            R2 = self.R2
            for l in range(lg):
                n0 = l & 7
                for rep in range(self.nrep):
                    y = self.dif([regs[l][t][rep * R2 + m] for m in range(R2)])
                    for mp in range(R2):
                        regs[l][t][rep * R2 + mp] = y[mp] * np.exp(-2j * np.pi * n0 * brev(mp, self.midbits) / blk)
            for l in range(lg):
                for rho in range(8):
                    lds[t][self.addr(self.db(l, rho))] = regs[l][t][rho]
            for l in range(lg):
                regs[l][t] = [lds[t][self.addr(self.dc(t, l, n))] for n in range(8)]
            # stage C: radix-8 over the low three bits
            for l in range(lg):
                regs[l][t] = self.dif(regs[l][t])
        return regs

    def inverse(self, regs):
        L, M, lg, blk = self.L, self.M, self.lg, self.blk
        lds = [np.zeros(self.rowwords, dtype=np.complex128) for _ in range(3)]
        R2 = self.R2
        for t in range(3):
            for l in range(lg):
                regs[l][t] = self.dit(regs[l][t])
            for l in range(lg):
                for n in range(8):
                    lds[t][self.addr(self.dc(t, l, n))] = regs[l][t][n]
            for l in range(lg):
                regs[l][t] = [lds[t][self.addr(self.db(l, rho))] for rho in range(8)]
            for l in range(lg):
                n0 = l & 7
                for rep in range(self.nrep):
                    x = [regs[l][t][rep * R2 + mp] * np.exp(+2j * np.pi * n0 * brev(mp, self.midbits) / blk) for mp in range(R2)]
                    y = self.dit(x)
                    for m in range(R2):
                        regs[l][t][rep * R2 + m] = y[m]
            for l in range(lg):
                for rho in range(8):
                    lds[t][self.addr(self.db(l, rho))] = regs[l][t][rho]
            for l in range(lg):
                regs[l][t] = [lds[t][self.addr(self.da(l, rho))] for rho in range(8)]
            for l in range(lg):
                x = [regs[l][t][rp] * np.exp(+2j * np.pi * l * brev(rp, 3) / L) for rp in range(8)]
                regs[l][t] = self.dit(x)
        w3 = np.exp(+2j * np.pi / 3)
        for l in range(lg):
            for rho in range(8):
                m = self.da(l, rho)
                y0 = regs[l][0][rho]
                y1 = regs[l][1][rho] * np.exp(+2j * np.pi * m / M)
                y2 = regs[l][2][rho] * np.exp(+2j * np.pi * 2 * m / M)
                regs[l][0][rho] = y0 + y1 + y2
                regs[l][1][rho] = y0 + w3 * y1 + w3 ** 2 * y2
                regs[l][2][rho] = y0 + w3 ** 2 * y1 + w3 ** 4 * y2
        return regs

    # ---- frequency held at (third t, position p), and the untangle twiddle w_X^f, X = 2 M ----
    def freq(self, t, p):
        return 3 * brev(p, self.logl) + t

    def ut(self, t, p):
        return np.exp(-2j * np.pi * self.freq(t, p) / (2 * self.M))

    @staticmethod
    def own_fwd(a, b, w):      # own element a (frequency f), mirrored element b (frequency M - f), w = w_X^f -> new a
        E = 0.5 * (a + np.conj(b))
        Dm = 0.5 * (a - np.conj(b))
        return E + w * (-1j * Dm)

    @staticmethod
    def own_inv(a, b, w):
        E = 0.5 * (a + np.conj(b))
        Dm = 0.5 * (a - np.conj(b))
        return E + 1j * (Dm * np.conj(w))

    def untangle(self, regs, inverse, nyq_in=None):
        """In place on DC registers; every lane computes its OWN elements from (own, mirrored partner).  Third 0: the partner
        register 7 - n of lane l ^ 1 (lanes 0 / 1: in-lane pairs); thirds 1 <-> 2: register 7 - n of the other third."""
        own = self.own_inv if inverse else self.own_fwd
        nyq = None
        old = [[list(regs[l][t]) for t in range(3)] for l in range(self.lg)]
        for l in range(self.lg):
            for n in range(8):
                # thirds 1 / 2
                regs[l][1][n] = own(old[l][1][n], old[l][2][7 - n], self.ut(1, self.dc(1, l, n)))
                regs[l][2][n] = own(old[l][2][n], old[l][1][7 - n], self.ut(2, self.dc(2, l, n)))
            if l >= 2:
                for n in range(8):
                    assert self.mirror_pos(self.dc(0, l, n)) == self.dc(0, l ^ 1, 7 - n)
                    regs[l][0][n] = own(old[l][0][n], old[l ^ 1][0][7 - n], self.ut(0, self.dc(0, l, n)))
            elif l == 0:      # positions 0..7: DC/Nyquist, M/2, (2,3), (4,7), (5,6)
                z0 = old[0][0][0]
                if not inverse:
                    regs[0][0][0] = z0.real + z0.imag
                    nyq = z0.real - z0.imag
                else:
                    regs[0][0][0] = 0.5 * (z0.real + nyq_in.real) + 0.5j * (z0.real - nyq_in.real)
                regs[0][0][1] = np.conj(old[0][0][1])
                for n, pn in ((2, 3), (3, 2), (4, 7), (7, 4), (5, 6), (6, 5)):
                    regs[0][0][n] = own(old[0][0][n], old[0][0][pn], self.ut(0, n))
            else:             # positions 8..15: (8,15), (9,14), (10,13), (11,12)
                for n in range(8):
                    regs[1][0][n] = own(old[1][0][n], old[1][0][7 - n], self.ut(0, 8 + n))
        return regs, nyq

    @staticmethod
    def mirror_pos(p):
        if p < 2:
            return p
        top = p.bit_length() - 1
        return 3 * (1 << top) - 1 - p

    # ---- whole row: real row (2 M) -> half spectrum in stored column order (+ Nyquist) and back ----
    def row_forward(self, x):
        z = x[0::2] + 1j * x[1::2]
        regs = [[[z[t * self.L + self.da(l, rho)] for rho in range(8)] for t in range(3)] for l in range(self.lg)]
        regs = self.forward(regs)
        regs, nyq = self.untangle(regs, False)
        out = np.zeros(self.M + 1, dtype=np.complex128)
        for l in range(self.lg):
            for t in range(3):
                for n in range(8):
                    out[2 * ((4 * t + (n >> 1)) * self.lg + l) + (n & 1)] = regs[l][t][n]
        out[self.M] = nyq
        return out

    def row_inverse(self, S):
        regs = [[[S[2 * ((4 * t + (n >> 1)) * self.lg + l) + (n & 1)] for n in range(8)] for t in range(3)] for l in range(self.lg)]
        regs, _ = self.untangle(regs, True, S[self.M])
        regs = self.inverse(regs)
        z = np.zeros(self.M, dtype=np.complex128)
        for l in range(self.lg):
            for t in range(3):
                for rho in range(8):
                    z[t * self.L + self.da(l, rho)] = regs[l][t][rho]
        x = np.zeros(2 * self.M)
        x[0::2], x[1::2] = z.real, z.imag
        return x

    # ---- LDS cycles of every exchange instruction of one wavefront (both rows of PAIRS pairs; thirds use disjoint buffers) ----
    def lds_cycles(self):
        m = self
        npairs = 64 // m.lg
        rowbytes = 8 * m.rowwords

        def lanes(fn):  # byte address per lane of the whole wave: lane group p works in its own pair of row buffers
            return [8 * fn(l % m.lg) + (l // m.lg) * 6 * rowbytes for l in range(64)]

        tot = {}

        def add(name, a, kind):
            c, i = bank_conflicts(a, kind)
            e = tot.setdefault(name + " " + kind, [0, 0])
            e[0] += c
            e[1] += i

        for rho in range(8):
            a = lanes(lambda l: m.addr(m.da(l, rho)))
            add("DA", a, "w64")
            add("DA", a, "r64")
            a = lanes(lambda l: m.addr(m.db(l, rho)))
            add("DB", a, "r64")
            add("DB", a, "w64")
        for t in range(3):
            for q in range(4):
                a = lanes(lambda l: m.addr(m.dc(t, l, 2 * q)))
                assert all(m.addr(m.dc(t, l, 2 * q + 1)) == m.addr(m.dc(t, l, 2 * q)) + 1 for l in range(m.lg))
                add(f"DC{t}", a, "r128")
                add(f"DC{t}", a, "w128")
        return {k: tuple(v) for k, v in tot.items()}, npairs


def check(logl, verbose=True):
    m = X3(logl)
    M, L = m.M, m.L
    rng = np.random.default_rng(logl)
    for dist in (m.da, m.db):
        assert sorted(dist(l, r) for l in range(m.lg) for r in range(8)) == list(range(L))
    for t in range(3):
        assert sorted(m.dc(t, l, n) for l in range(m.lg) for n in range(8)) == list(range(L))
    assert len({m.addr(i) for i in range(L)}) == L and max(m.addr(i) for i in range(L)) < m.rowwords
    assert sorted(m.col_of_pos(q) for q in range(M)) == list(range(M))
    # complex transform: third t, position p holds frequency 3 brev(p) + t
    z = rng.standard_normal(M) + 1j * rng.standard_normal(M)
    regs = [[[z[t * L + m.da(l, rho)] for rho in range(8)] for t in range(3)] for l in range(m.lg)]
    regs = m.forward(regs)
    Z = np.fft.fft(z)
    err = max(abs(regs[l][t][n] - Z[m.freq(t, m.dc(t, l, n))]) for l in range(m.lg) for t in range(3) for n in range(8))
    assert err < 1e-9 * np.abs(Z).max(), err
    regs = m.inverse(regs)
    err = max(abs(regs[l][t][rho] - M * z[t * L + m.da(l, rho)]) for l in range(m.lg) for t in range(3) for rho in range(8))
    assert err < 1e-9 * M, err
    # real rows: the stored row holds rfft(x) at the columns col_of_pos names, positions in the tile kernels' order
    x = rng.standard_normal(2 * M)
    S = m.row_forward(x)
    R = np.fft.rfft(x)
    for q in range(M):
        t, p = divmod(q, L)
        assert abs(S[m.col_of_pos(q)] - R[m.freq(t, p)]) < 1e-9 * np.abs(R).max(), q
    assert abs(S[M] - R[M]) < 1e-9 * np.abs(R).max()
    back = m.row_inverse(S)
    assert np.abs(back - M * x).max() < 1e-9 * M
    if verbose:
        cyc, npairs = m.lds_cycles()
        print(f"M = {M} (L = {L}, {m.lg} lanes per pair, {npairs} pair(s) per wave): model OK; LDS cycles (actual, ideal):", cyc)


if __name__ == "__main__":
    for logl in (8, 9):
        check(logl)
