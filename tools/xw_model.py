#!/usr/bin/env python3
"""Executable model of the wave-private X pass (csrc/fftconv_xw.inc): one wavefront owns a row pair (y, y + Y/2), every
lane keeps 16 complex points of each row in registers, the length-M transform is three in-register stages
(radix-8 over the top 3 index bits, radix-M/64 over the middle bits, radix-8 over the low 3 bits) joined by two wave-private
LDS exchanges, and the untangle / Y radix-2 step run in registers.  The model works lane by lane and register by register
with the kernel's index maps, LDS addresses and twiddle tables, and is checked against numpy's FFT; `bank_conflicts`
prices each exchange with the LDS rules of MI355X_MICROARCH.md.  Run: python tools/xw_model.py
"""
import numpy as np


def brev(v, bits):
    r = 0
    for b in range(bits):
        if v >> b & 1:
            r |= 1 << (bits - 1 - b)
    return r


class XW:
    def __init__(self, logm):
        self.logm = logm
        self.M = 1 << logm
        self.lg = self.M // 16          # lanes per row pair
        self.loglg = logm - 4
        self.midbits = logm - 6         # 4, 3, 2
        self.R2 = 1 << self.midbits
        self.nrep = 16 // self.R2       # middle-stage groups per lane (1, 2, 4)
        self.blk = self.M // 8          # elements per top-3-bit block
        self.pad = 0 if logm == 8 else 8  # complex words of padding per block (M = 256: 64 row buffers per workgroup, no room)
        self.bstride = self.blk + self.pad

    # ---- array index <-> (lane, reg) in the three distributions ----
    def dr(self, l, rho):               # real side: rho = 2 r + b
        r, b = rho >> 1, rho & 1
        return r * self.blk + 2 * l + b

    def d2(self, l, rho):
        # lanes: [extra high bits not in regs][low 3 bits]; regs: middle bits (+ extra high bits when R2 < 16)
        n0 = l & 7
        lh = l >> 3                      # high lane bits: the top-3 bits that are not in registers
        mid = rho & (self.R2 - 1)
        xh = rho >> self.midbits         # extra register bits = LOW part of the top-3 field? choose: high part
        nx = {16: 0, 8: 1, 4: 2}[self.R2]  # number of top bits held in registers
        top = (xh << (3 - nx)) | lh      # xh are the most significant of the top-3 bits
        return top * self.blk + mid * 8 + n0

    def group_of(self, lam, sigma):
        if lam == 0:
            return sigma
        u = lam.bit_length() - 1
        low = lam - (1 << u)
        g0 = (2 << u) | low
        if sigma == 0:
            return g0
        return (2 << u) | (1 << u) | (~low & ((1 << u) - 1))

    def d3(self, l, rho):
        sigma, n = rho >> 3, rho & 7
        return 8 * self.group_of(l, sigma) + n

    # ---- LDS address (in complex words) of array index i: 8 words of padding per top-3-bit block, and the four 16-B chunks of
    # every 8-point group XOR-swizzled by two parities of the group index (found by search with `bank_conflicts`: every
    # DR / D2 access conflict-free, the D3 ones within 1.1x (stores) and 1.5x (loads) of conflict-free)
    @staticmethod
    def swz(g):
        par = lambda v: bin(v).count("1") & 1
        return par(g & 6) | (par(g & 59) << 1)

    def addr(self, i):
        blk, w = divmod(i, self.blk)
        g, n = divmod(w, 8)
        gg = blk * (self.blk // 8) + g
        return blk * self.bstride + g * 8 + 2 * ((n >> 1) ^ self.swz(gg)) + (n & 1)

    # ---- physical spectrum column of position p (the order the new kernels store) ----
    def col_of_pos(self, p):
        g, n = divmod(p, 8)
        for lam in range(self.lg):
            for sigma in (0, 1):
                if self.group_of(lam, sigma) == g:
                    return 2 * ((4 * sigma + (n >> 1)) * self.lg + lam) + (n & 1)
        raise AssertionError

    # ---- in-register radix-2^k DIF / DIT on a list of complex values (bit-reversed output for DIF) ----
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
                    w = np.exp(-2j * np.pi * j / (2 * h))
                    x[i], x[i + h] = a + b, (a - b) * w
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
                    w = np.exp(+2j * np.pi * j / (2 * h))
                    a, b = x[i], x[i + h] * w
                    x[i], x[i + h] = a + b, a - b
            h *= 2
        return x

    # ---- forward transform of one row of M complex points held as regs[l][rho] in DR; returns regs in D3 ----
    def forward(self, regs):
        M, lg, blk = self.M, self.lg, self.blk
        lds = np.zeros(8 * self.bstride, dtype=np.complex128)
        # F1: radix-8 over r for each b; external twiddle w_M^(t k), k = brev3(register position)
        for l in range(lg):
            for b in (0, 1):
                t = 2 * l + b
                y = self.dif([regs[l][2 * r + b] for r in range(8)])
                for rp in range(8):
                    k = brev(rp, 3)
                    regs[l][2 * rp + b] = y[rp] * np.exp(-2j * np.pi * t * k / M)
        # E1: DR -> LDS -> D2
        for l in range(lg):
            for rho in range(16):
                lds[self.addr(self.dr(l, rho))] = regs[l][rho]
        for l in range(lg):
            for rho in range(16):
                regs[l][rho] = lds[self.addr(self.d2(l, rho))]
        # F2: radix-R2 over the middle bits; external twiddle w_{M/8}^(n0 k1)
        R2 = self.R2
        for l in range(lg):
            n0 = l & 7
            for rep in range(self.nrep):
                y = self.dif([regs[l][rep * R2 + m] for m in range(R2)])
                for mp in range(R2):
                    k1 = brev(mp, self.midbits)
                    regs[l][rep * R2 + mp] = y[mp] * np.exp(-2j * np.pi * n0 * k1 / blk)
        # E2: D2 -> LDS -> D3
        for l in range(lg):
            for rho in range(16):
                lds[self.addr(self.d2(l, rho))] = regs[l][rho]
        for l in range(lg):
            for rho in range(16):
                regs[l][rho] = lds[self.addr(self.d3(l, rho))]
        # F3: radix-8 over n0 in each slot
        for l in range(lg):
            for s in (0, 1):
                y = self.dif([regs[l][8 * s + n] for n in range(8)])
                for n in range(8):
                    regs[l][8 * s + n] = y[n]
        return regs

    def inverse(self, regs):
        M, lg, blk = self.M, self.lg, self.blk
        lds = np.zeros(8 * self.bstride, dtype=np.complex128)
        for l in range(lg):
            for s in (0, 1):
                y = self.dit([regs[l][8 * s + n] for n in range(8)])
                for n in range(8):
                    regs[l][8 * s + n] = y[n]
        for l in range(lg):
            for rho in range(16):
                lds[self.addr(self.d3(l, rho))] = regs[l][rho]
        for l in range(lg):
            for rho in range(16):
                regs[l][rho] = lds[self.addr(self.d2(l, rho))]
        R2 = self.R2
        for l in range(lg):
            n0 = l & 7
            for rep in range(self.nrep):
                x = [regs[l][rep * R2 + mp] * np.exp(+2j * np.pi * n0 * brev(mp, self.midbits) / blk) for mp in range(R2)]
                y = self.dit(x)
                for m in range(R2):
                    regs[l][rep * R2 + m] = y[m]
        for l in range(lg):
            for rho in range(16):
                lds[self.addr(self.d2(l, rho))] = regs[l][rho]
        for l in range(lg):
            for rho in range(16):
                regs[l][rho] = lds[self.addr(self.dr(l, rho))]
        for l in range(lg):
            for b in (0, 1):
                t = 2 * l + b
                x = [regs[l][2 * rp + b] * np.exp(+2j * np.pi * t * brev(rp, 3) / M) for rp in range(8)]
                y = self.dit(x)
                for r in range(8):
                    regs[l][2 * r + b] = y[r]
        return regs

    # ---- untangle in D3 registers (returns Nyquist value from lane 0) ----
    def mirror(self, p):
        if p < 2:
            return p
        top = p.bit_length() - 1
        return 3 * (1 << top) - 1 - p

    def ut(self, p):  # w_X^{brev(p)} = w_{2M}^f
        return np.exp(-2j * np.pi * brev(p, self.logm) / (2 * self.M))

    def untangle_fwd(self, regs):
        nyq = None
        for l in range(self.lg):
            done = set()
            for rho in range(16):
                p = self.d3(l, rho)
                if p in done:
                    continue
                if p == 0:
                    z0 = regs[l][rho]
                    regs[l][rho] = z0.real + z0.imag
                    nyq = z0.real - z0.imag
                    done.add(0)
                    continue
                if p == 1:
                    regs[l][rho] = np.conj(regs[l][rho])
                    done.add(1)
                    continue
                pm = self.mirror(p)
                pp = min(p, pm)
                pm = max(p, pm)
                # both must be in this lane
                rp = [r for r in range(16) if self.d3(l, r) == pp][0]
                rm = [r for r in range(16) if self.d3(l, r) == pm][0]
                a, b = regs[l][rp], regs[l][rm]
                E = 0.5 * (a + np.conj(b))
                Dm = 0.5 * (a - np.conj(b))
                wO = self.ut(pp) * (-1j * Dm)
                regs[l][rp] = E + wO
                regs[l][rm] = np.conj(E - wO)
                done.update((pp, pm))
        return regs, nyq

    def untangle_inv(self, regs, nyq):
        for l in range(self.lg):
            done = set()
            for rho in range(16):
                p = self.d3(l, rho)
                if p in done:
                    continue
                if p == 0:
                    x0 = regs[l][rho].real
                    regs[l][rho] = 0.5 * (x0 + nyq.real) + 0.5j * (x0 - nyq.real)
                    done.add(0)
                    continue
                if p == 1:
                    regs[l][rho] = np.conj(regs[l][rho])
                    done.add(1)
                    continue
                pm = self.mirror(p)
                pp, pm = min(p, pm), max(p, pm)
                rp = [r for r in range(16) if self.d3(l, r) == pp][0]
                rm = [r for r in range(16) if self.d3(l, r) == pm][0]
                a, b = regs[l][rp], regs[l][rm]
                E = 0.5 * (a + np.conj(b))
                Dm = 0.5 * (a - np.conj(b))
                iO = 1j * (Dm * np.conj(self.ut(pp)))
                regs[l][rp] = E + iO
                regs[l][rm] = np.conj(E - iO)
                done.update((pp, pm))
        return regs

    # ---- whole row: real row (2M) -> half spectrum in stored column order (+ Nyquist) and back ----
    def row_forward(self, x):
        z = x[0::2] + 1j * x[1::2]
        regs = [[z[self.dr(l, rho)] for rho in range(16)] for l in range(self.lg)]
        regs = self.forward(regs)
        regs, nyq = self.untangle_fwd(regs)
        out = np.zeros(self.M + 1, dtype=np.complex128)
        for l in range(self.lg):
            for rho in range(16):
                sigma, n = rho >> 3, rho & 7
                col = 2 * ((4 * sigma + (n >> 1)) * self.lg + l) + (n & 1)
                out[col] = regs[l][rho]
        out[self.M] = nyq
        return out

    def row_inverse(self, S):
        regs = [[S[2 * ((4 * (rho >> 3) + ((rho & 7) >> 1)) * self.lg + l) + (rho & 1)] for rho in range(16)]
                for l in range(self.lg)]
        regs = self.untangle_inv(regs, S[self.M])
        regs = self.inverse(regs)
        z = np.zeros(self.M, dtype=np.complex128)
        for l in range(self.lg):
            for rho in range(16):
                z[self.dr(l, rho)] = regs[l][rho]
        x = np.zeros(2 * self.M)
        x[0::2], x[1::2] = z.real, z.imag
        return x


# ---- LDS bank-conflict pricing (MI355X_MICROARCH.md §LDS): extra cycles of one wave64 instruction ----
def bank_conflicts(byte_addrs, kind):
    """byte_addrs: 64 per-lane byte addresses; kind in {r64, r128, w64, w128}.  Returns (cycles, ideal cycles)."""
    if kind == "r64":
        groups, nbanks, width = [range(0, 32), range(32, 64)], 64, 2
    elif kind == "r128":
        groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
                  [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
        groups = groups + [[x + 32 for x in g] for g in groups]
        nbanks, width = 64, 4
    elif kind == "w64":
        groups, nbanks, width = [range(16 * k, 16 * k + 16) for k in range(4)], 32, 2
    else:  # w128
        groups, nbanks, width = [range(8 * k, 8 * k + 8) for k in range(8)], 32, 4
    total = 0
    for g in groups:
        per_bank = {}
        for lane in g:
            a = byte_addrs[lane]
            if a is None:
                continue
            for d in range(width):
                dw = a // 4 + d
                per_bank.setdefault(dw % nbanks, set()).add(dw)
        total += max([len(v) for v in per_bank.values()] or [1])
    return total, len(groups)


def check(logm):
    m = XW(logm)
    M = m.M
    rng = np.random.default_rng(logm)
    # distributions are bijections
    for dist in (m.dr, m.d2, m.d3):
        assert sorted(dist(l, r) for l in range(m.lg) for r in range(16)) == list(range(M))
    assert sorted(m.addr(i) for i in range(M)) == sorted(set(m.addr(i) for i in range(M)))
    assert sorted(m.col_of_pos(p) for p in range(M)) == list(range(M))
    # complex transform: positions = bit-reversed frequencies
    z = rng.standard_normal(M) + 1j * rng.standard_normal(M)
    regs = [[z[m.dr(l, rho)] for rho in range(16)] for l in range(m.lg)]
    regs = m.forward(regs)
    Z = np.fft.fft(z)
    err = max(abs(regs[l][rho] - Z[brev(m.d3(l, rho), logm)]) for l in range(m.lg) for rho in range(16))
    assert err < 1e-9 * np.abs(Z).max(), err
    regs = m.inverse(regs)
    err = max(abs(regs[l][rho] - M * z[m.dr(l, rho)]) for l in range(m.lg) for rho in range(16))
    assert err < 1e-9 * M, err
    # real rows
    x = rng.standard_normal(2 * M)
    S = m.row_forward(x)
    R = np.fft.rfft(x)
    for p in range(M):
        assert abs(S[m.col_of_pos(p)] - R[brev(p, logm)]) < 1e-9 * np.abs(R).max(), p
    assert abs(S[M] - R[M]) < 1e-9 * np.abs(R).max()
    back = m.row_inverse(S)
    assert np.abs(back - M * x).max() < 1e-9 * M
    # bank conflicts of every exchange instruction (one row pair group per wave is the worst case: M = 1024 fills a wave)
    npairs = 64 // m.lg
    rep = {}

    def lanes(fn):  # byte address per lane for the whole wave (groups use disjoint buffers)
        return [8 * (fn(l % m.lg) + (l // m.lg) * 2 * 8 * m.bstride) for l in range(64)]

    tot = {}
    # E1 write: ds_write_b128 of (rho = 2r, 2r+1); E1' read the same with ds_read_b128
    for r in range(8):
        a = lanes(lambda l: m.addr(m.dr(l, 2 * r)))
        assert all(m.addr(m.dr(l, 2 * r + 1)) == m.addr(m.dr(l, 2 * r)) + 1 for l in range(m.lg))
        for kind in ("w128", "r128"):
            c, i = bank_conflicts(a, kind)
            tot.setdefault("E1 DR " + kind, [0, 0])
            tot["E1 DR " + kind][0] += c
            tot["E1 DR " + kind][1] += i
    for rho in range(16):
        a = lanes(lambda l: m.addr(m.d2(l, rho)))
        for kind in ("r64", "w64"):
            c, i = bank_conflicts(a, kind)
            tot.setdefault("D2 " + kind, [0, 0])
            tot["D2 " + kind][0] += c
            tot["D2 " + kind][1] += i
    for s in (0, 1):
        for q in range(4):
            # logical chunk q (n = 2q, 2q + 1) of the lane's group: its physical 16-B chunk is rotated by the group index
            a = lanes(lambda l: m.addr(m.d3(l, 8 * s + 2 * q)))
            assert all(m.addr(m.d3(l, 8 * s + 2 * q + 1)) == m.addr(m.d3(l, 8 * s + 2 * q)) + 1 for l in range(m.lg))
            for kind in ("r128", "w128"):
                c, i = bank_conflicts(a, kind)
                tot.setdefault("D3 " + kind, [0, 0])
                tot["D3 " + kind][0] += c
                tot["D3 " + kind][1] += i
    print(f"M = {M}: model OK; LDS cycles (actual, ideal) per row:", {k: tuple(v) for k, v in tot.items()}, f"pairs/wave {npairs}")


if __name__ == "__main__":
    for logm in (8, 9, 10):
        check(logm)
