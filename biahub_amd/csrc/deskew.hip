// Fused oblique-light-sheet deskew for gfx950 (MI355X).
//
// Replaces, in one kernel and one pass over HBM, the reference's chain
//   permute/flip copy (biahub/deskew.py:110) -> edge pad (:517-519) -> grid build (:113-154)
//   -> F.grid_sample (:531-533) -> mean over N (:536)
//
//   out[a, yo, xo] = (1/N) * sum_{k<N} lerp( in[:, Y-1-min(aN+k, Y-1), X-1-yo], ix(xo, aN+k) )
//
// Data layout / access pattern
//   in  (Z, Y, X)  : X contiguous.  For fixed (a, k) the kernel needs the (Z x X) plane
//                    in[:, yin, :]; a workgroup stages a [z-window][TX] tile of it with
//                    row-contiguous (coalesced) reads and stores it TRANSPOSED in LDS as
//                    [k][x][z] with an odd z-stride, so that the compute phase — lanes along
//                    the output x axis, which walks input z at px_to_scan_ratio per step —
//                    reads consecutive LDS banks.
//   out (Za, X, Xp): Xp contiguous.  Each wave writes 64 consecutive floats per store.
//   Every input voxel is read once (plus a 2-3 row overlap between neighbouring x-chunks)
//   and every output voxel written once: algorithmic bytes 4*(V_in + V_out).
//
// Coordinates reproduce the reference's float32 arithmetic operation by operation
// (oracle/oracle_np.py:deskew_coords), so sample positions are bit-identical to torch's.
#include "common.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace bh {

struct DeskewGeom {
    int Z, Y, X;     // input
    int Za, Xp;      // output (Za, X, Xp)
    int N;           // average_n_slices
    float px, pxct, offset, zm1;
    int XC;          // output-x chunk per workgroup (multiple of 256)
    int ZS;          // LDS z stride (odd)
    int ZC;          // max z-window length (<= ZS)
    // fused overhang-fill prologue (FILL kernels): zero-mask bits + per-block sums
    uint32_t* mask0;  // [Za*X][W32] one bit per output voxel (1 = exact zero)
    double* psum;     // per-block partial sums of the outputs
    int W32;          // mask words per output row (even)
    const int* enable;  // FILL kernels, may be null: device flag, 0 = nothing to do (the conditional pass behind the one-pass path)
    // one-pass fill (deskew_pers_kernel<NK, 2>): the geometry's zero pattern and its dilation, one bit per (a, x'), WB words per a;
    // the fill value (already final) and the fallback flag live in *st
    const uint32_t* gbits;
    const uint32_t* dgbits;
    int WB;
    FillStats* st;
};

// The reference's sample position along the scan axis, in its float32 operation order:
//   in_z = px*x - (px*ct)*zo + offset ; g = 2*in_z/(Z-1) - 1 ; ix = ((g+1)/2)*(Z-1)
__host__ __device__ inline float deskew_ix(float px, float pxct, float offset, float zm1, int xo, int zo) {
#pragma clang fp contract(off)
    float t1 = px * (float)xo;
    float t2 = pxct * (float)zo;
    float in_z = (t1 - t2) + offset;
    float g = (2.0f * in_z) / zm1 - 1.0f;
    float ix = ((g + 1.0f) / 2.0f) * zm1;
    return ix;
}

template <typename T>
__device__ __forceinline__ float to_f32(T v) {
    return (float)v;
}

template <typename T>
struct Vec4;
template <>
struct Vec4<float> {
    typedef float type __attribute__((ext_vector_type(4)));
};
template <>
struct Vec4<uint16_t> {
    typedef unsigned short type __attribute__((ext_vector_type(4)));
};
template <>
struct Vec4<int16_t> {
    typedef short type __attribute__((ext_vector_type(4)));
};
template <>
struct Vec4<uint8_t> {
    typedef unsigned char type __attribute__((ext_vector_type(4)));
};

// a / N, N a small positive integer: q = a*(1/N) corrected by one fused residual step
// (r = a - q*N is exact in fma), which is the correctly rounded quotient for these operands.
__device__ __forceinline__ float div_small(float a, float n, float rn) {
    const float q = a * rn;
    const float r = __builtin_fmaf(-q, n, a);
    return __builtin_fmaf(r, rn, q);
}

// NK > 0: N == NK known at compile time (interpolation plan kept in registers).
// NK == 0: generic N, plan recomputed per row.
// TX = input-x columns per workgroup (row segment TX*sizeof(TIN) bytes, loaded 4 elements per lane);
// J = outputs per lane per row, so one wave covers the whole XC = 64*J output chunk and the NT/64
// waves take different rows.  LDS tile is [k][z][TX+1]: the odd row pitch makes both the staging
// stores (lanes along x) and the compute loads (lanes along z) bank-conflict free.
// FILLM: 0 no fill; 1 the fused prologue of the mask pipeline (zero-mask bits + block sums, zeros not stored); 2 the one-pass
// fill (see deskew_pers_kernel below: the fill value is final in *g.st, whole rows are written, the dilated geometric zero
// pattern g.dgbits says where the fill value goes, an exact zero outside g.gbits raises g.st->fallback).
template <typename TIN, int TX, int J, int NK, int NT, bool DMA, int FILLM>
__global__ __launch_bounds__(NT) void deskew_kernel(const TIN* __restrict__ in, float* __restrict__ out,
                                                    DeskewGeom g) {
#pragma clang fp contract(off)
    constexpr bool FILL = FILLM == 1, ROWS = FILLM == 2;
    constexpr int XC = 64 * J;
    constexpr int PITCH = TX + 1;
    extern __shared__ __attribute__((aligned(16))) float tile[];  // [N][ZC][PITCH]
    if (FILL && g.enable != nullptr && *g.enable == 0) return;
    const int tid = threadIdx.x;
    const int xt0 = blockIdx.x * TX;
    const int xo0 = blockIdx.y * XC;
    const int a = blockIdx.z;
    const int N = NK > 0 ? NK : g.N;
    const int xoN = min(XC, g.Xp - xo0);
    const int zo0 = a * N;

    // z-window covering every sample of this (a, xo-chunk): ix is monotone in xo and zo
    const float ix_min = deskew_ix(g.px, g.pxct, g.offset, g.zm1, xo0, zo0 + N - 1);
    const float ix_max = deskew_ix(g.px, g.pxct, g.offset, g.zm1, xo0 + xoN - 1, zo0);
    const int zlo = (int)floorf(ix_min);
    int zcnt = (int)floorf(ix_max) + 2 - zlo;
    zcnt = min(zcnt, g.ZC);  // host guarantees zcnt <= ZC; clamp is a memory-safety net only
    const int kstride = g.ZC * PITCH;

    // Overhang chunk: every sample lies outside the scanned range, so the output is exactly 0
    // (grid_sample zero padding) — stream zeros, touch neither the input nor LDS.
    if (zlo + zcnt <= 0 || zlo >= g.Z) {
        const int lane0 = tid & 63;
        for (int xl = tid >> 6; xl < TX; xl += NT / 64) {
            const int x = xt0 + xl;
            if (x >= g.X) break;
            const size_t orow_i = (size_t)a * g.X + (g.X - 1 - x);
            float* orow = out + orow_i * g.Xp;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int xo = xo0 + lane0 + 64 * j;
                if (FILL) {
                    // every voxel here is an exact zero: set the mask bits, leave the data to the fill pass
                    const unsigned long long bits = __ballot(xo < g.Xp);
                    if (lane0 == 0 && xo0 + 64 * j < g.Xp)
                        *reinterpret_cast<unsigned long long*>(g.mask0 + orow_i * g.W32 + (xo0 + 64 * j) / 32) = bits;
                } else if (xo < g.Xp) {
                    orow[xo] = ROWS ? g.st->fill : 0.0f;
                }
            }
        }
        if (FILL && tid == 0)
            g.psum[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = 0.0;
        return;
    }

    // ---- stage: global rows (TX contiguous x) -> LDS [k][z][x] ------------------------------
    // Fast path (float32, TX == 64, tile inside the volume): one LDS-DMA per row — every lane
    // fetches 4 B of a 256-B row segment straight into LDS (wave-uniform row base + lane*4, so the
    // odd row pitch is free), nothing is staged in VGPRs and all of a wave's rows are in flight
    // at once.  Rows outside [0, Z) are zero-filled with ordinary LDS stores.
    // Other dtypes / ragged tiles take the register path: unconditional (clamped address + select)
    // vector loads, because predicated loads inside an unrolled loop serialise on gfx950 and cost
    // ~30 % of HBM throughput (tools/membench*.hip).
    const size_t plane = (size_t)g.Y * g.X;
    if (DMA && TX == 64 && sizeof(TIN) == 4 && (xt0 + TX <= g.X)) {
        const int lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        constexpr int NW = NT / 64;
        // rows of the window that exist in the volume: [za, zb) relative to zlo
        const int za = max(0, -zlo), zb = min(zcnt, g.Z - zlo);
        for (int k = 0; k < N; ++k) {
            const int yin = g.Y - 1 - min(zo0 + k, g.Y - 1);
            float* dk = tile + k * kstride;
            for (int zz = wave; zz < za; zz += NW) dk[zz * PITCH + lane] = 0.0f;
            for (int zz = max(zb, 0) + wave; zz < zcnt; zz += NW) dk[zz * PITCH + lane] = 0.0f;
            const int z0 = za + wave;
            const TIN* src = in + (size_t)(zlo + z0) * plane + (size_t)yin * g.X + xt0 + lane;
            unsigned lds_dst = (unsigned)(size_t)(dk + z0 * PITCH);  // LDS byte address (wave-uniform)
            for (int zz = z0; zz < zb; zz += NW) {
                unsigned keep;
                asm volatile(
                    "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                    : "=&s"(keep)
                    : "v"(src), "s"(lds_dst)
                    : "memory");
                src += (size_t)NW * plane;
                lds_dst += NW * PITCH * 4;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the compiler does not count asm LDS-DMA
    } else {
        typedef typename Vec4<TIN>::type V4;
        constexpr int LPR = TX / 4;        // lanes per row
        constexpr int ZSTEP = NT / LPR;    // rows per pass of the workgroup
        constexpr int U = 8;
        const int xq = (tid % LPR) * 4;
        const int zz0 = tid / LPR;
        const bool vec_ok = (xt0 + TX <= g.X) && ((g.X & 3) == 0);  // whole tile inside, rows 4-aligned
        for (int k = 0; k < N; ++k) {
            const int yin = g.Y - 1 - min(zo0 + k, g.Y - 1);
            const TIN* src = in + (size_t)yin * g.X + xt0 + xq;
            float* dst = tile + k * kstride + xq;
            if (vec_ok) {
                for (int zb = zz0; zb < zcnt; zb += ZSTEP * U) {
                    V4 v[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int z = zlo + zb + u * ZSTEP;
                        const int zc = max(0, min(z, g.Z - 1));
                        v[u] = *reinterpret_cast<const V4*>(src + (size_t)zc * plane);
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int zz = zb + u * ZSTEP;
                        const int z = zlo + zz;
                        const bool ok = z >= 0 && z < g.Z;
                        if (zz < zcnt) {
                            float* d = dst + zz * PITCH;
                            d[0] = ok ? to_f32(v[u].x) : 0.0f;
                            d[1] = ok ? to_f32(v[u].y) : 0.0f;
                            d[2] = ok ? to_f32(v[u].z) : 0.0f;
                            d[3] = ok ? to_f32(v[u].w) : 0.0f;
                        }
                    }
                }
            } else {  // ragged tile: scalar, clamped
                for (int zz = zz0; zz < zcnt; zz += ZSTEP) {
                    const int z = zlo + zz;
                    const int zc = max(0, min(z, g.Z - 1));
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int x = xt0 + xq + c;
                        const float t = to_f32(in[(size_t)zc * plane + (size_t)yin * g.X + min(x, g.X - 1)]);
                        dst[zz * PITCH + c] = (x < g.X && z == zc) ? t : 0.0f;
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- compute: lanes along xo, J outputs per lane spaced by 64; one wave per output row ----
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int xbase = xo0 + lane;

    constexpr int NKK = NK > 0 ? NK : 1;
    int i0[NKK][J];
    float w0[NKK][J], w1[NKK][J];
    if (NK > 0) {
#pragma unroll
        for (int k = 0; k < NKK; ++k)
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const float ix = deskew_ix(g.px, g.pxct, g.offset, g.zm1, xbase + 64 * j, zo0 + k);
                const float fl = floorf(ix);
                w1[k][j] = ix - fl;
                w0[k][j] = (fl + 1.0f) - ix;
                int rel = (int)fl - zlo;
                rel = max(0, min(rel, g.ZC - 2));  // lanes past Xp may fall outside the window
                i0[k][j] = k * kstride + rel * PITCH;
            }
    }
    const float fN = (float)N;
    const float rN = 1.0f / fN;
    double tsum = 0.0;
    // ROWS: this lane's bits of the geometric zero pattern and of its dilation (the same for every row of the plane)
    bool gz[J], dgz[J];
    float fillv = 0.0f, zmin = 1.0f;
    if (ROWS) {
        fillv = g.st->fill;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int xo = min(xbase + 64 * j, g.Xp - 1);
            const size_t wi = (size_t)a * g.WB + (xo >> 5);
            gz[j] = (g.gbits[wi] >> (xo & 31)) & 1u;
            dgz[j] = (g.dgbits[wi] >> (xo & 31)) & 1u;
        }
    }
    for (int xl = wave; xl < TX; xl += NT / 64) {
        const int x = xt0 + xl;
        if (x >= g.X) break;
        const int yo = g.X - 1 - x;
        const size_t orow_i = (size_t)a * g.X + yo;
        float* orow = out + orow_i * g.Xp;
        const float* tcol = tile + xl;
        float acc[J];
        if (NK > 0) {
#pragma unroll
            for (int j = 0; j < J; ++j) {
                float s = 0.0f;
#pragma unroll
                for (int k = 0; k < NKK; ++k) {
                    const float v0 = tcol[i0[k][j]];
                    const float v1 = tcol[i0[k][j] + PITCH];
                    const float val = __builtin_fmaf(v1, w1[k][j], v0 * w0[k][j]);
                    s = (k == 0) ? val : s + val;
                }
                acc[j] = s;
            }
        } else {
#pragma unroll
            for (int j = 0; j < J; ++j) {
                float s = 0.0f;
                for (int k = 0; k < N; ++k) {
                    const float ix = deskew_ix(g.px, g.pxct, g.offset, g.zm1, xbase + 64 * j, zo0 + k);
                    const float fl = floorf(ix);
                    int rel = (int)fl - zlo;
                    rel = max(0, min(rel, g.ZC - 2));
                    const float* p = tcol + k * kstride + rel * PITCH;
                    const float val = __builtin_fmaf(p[PITCH], ix - fl, p[0] * ((fl + 1.0f) - ix));
                    s = (k == 0) ? val : s + val;
                }
                acc[j] = s;
            }
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int xo = xbase + 64 * j;
            const float val = (N > 1) ? div_small(acc[j], fN, rN) : acc[j];
            if (FILL) {
                // fused prologue of the overhang fill: zero-mask bits + running sum; exact zeros are not
                // stored (the fill pass overwrites every masked voxel anyway)
                const bool inb = xo < g.Xp;
                const unsigned long long bits = __ballot(inb && val == 0.0f);
                if (lane == 0 && xo0 + 64 * j < g.Xp)
                    *reinterpret_cast<unsigned long long*>(g.mask0 + orow_i * g.W32 + (xo0 + 64 * j) / 32) = bits;
                if (inb && val != 0.0f) {
                    orow[xo] = val;
                    tsum += (double)val;
                }
            } else if (ROWS) {
                if (xo < g.Xp) {
                    if (!gz[j]) zmin = fminf(zmin, fabsf(val));  // an exact zero that geometry does not explain: a data zero
                    orow[xo] = dgz[j] ? fillv : val;
                }
            } else if (xo < g.Xp) {
                orow[xo] = val;
            }
        }
    }
    if (ROWS && __ballot(zmin == 0.0f) != 0ull && lane == 0) atomicOr(&g.st->fallback, 1);
    if (FILL) {
        __shared__ double wsum[NT / 64];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tsum += __shfl_down(tsum, o, 64);
        if (lane == 0) wsum[wave] = tsum;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < NT / 64; ++w) t += wsum[w];
            g.psum[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Persistent, double-buffered form of the kernel above for the common case (float32 input, X a multiple of 64): one
// 512-thread workgroup per CU walks the (a, x'-chunk, x-tile) tiles; while the eight wavefronts sample tile t out of one LDS
// buffer, the LDS-DMA loads of tile t + 1 are already landing in the other (the kernel above stages, waits, samples: its
// memory pipe idles while it computes and vice versa, and only a second workgroup on the CU overlaps the two).  The
// wavefronts are specialised: two LOADERS only issue LDS-DMA (and the zero rows of a window that leaves the volume), six
// SAMPLERS only read LDS and store.  Loads and stores retire through one in-order counter per wavefront (vmcnt), so a
// wavefront that did both could not wait for its tile without also waiting for every store of the previous one; a loader
// has no stores to wait for, a sampler never waits on vmcnt at all.  One workgroup barrier per tile hands the freshly landed
// buffer to the samplers and the drained one back to the loaders.  Overhang tiles (every sample outside the scanned range)
// cost a handful of mask words and no LDS.
// A lane owns FOUR CONSECUTIVE x' (one 16-byte store per row instead of four 4-byte ones); their taps fall into at most three
// consecutive z rows per averaged slice, which are read once and selected per output — 3 N LDS reads per row instead of 8 N.
// Same float32 sample positions, same per-output operation order as deskew_kernel: bit-identical results.
//
// FILLM = 0: no fill.  1: the fused prologue of the mask pipeline (zero-mask bits + block sums, exact zeros not stored; fill.hip
// finishes).  2: ONE-PASS fill — the fill value is final before this kernel starts (*g.st), and WHOLE rows are written:
//   * which outputs are exact zeros is geometry: (a, x') whose N taps all fall outside the scanned range — the same for every
//     row of a plane, so the zero mask and its radius-3 dilation are two small bit planes (g.gbits, g.dgbits: deskew_rows.inc);
//     a sampler lane replaces the outputs inside the dilated pattern by the fill value and stores its 16 bytes as usual;
//   * tiles whose whole window lies outside the volume are pure fill: the two LOADER wavefronts write them, a quota of rows per
//     tile interval in front of the next tile's LDS-DMA (they are otherwise idle, and the samplers are bound by their own
//     instruction stream, so the 9 GB of overhang stores ride beside the sampling instead of behind it);
//   * an output that is exactly zero although geometry does not say so (a zero run in the data) would have been part of the
//     reference's mask: the lane raises g.st->fallback and the conditional mask pipeline queued behind this kernel redoes the
//     volume (bh_deskew).  Camera data, flat-fielded or deconvolved volumes have no such voxels.
template <int NK, int FILLM>
__global__ __launch_bounds__(512) void deskew_pers_kernel(const float* __restrict__ in, float* __restrict__ out, DeskewGeom g,
                                                          int ntx, int nxc) {
#pragma clang fp contract(off)
    constexpr bool FILL = FILLM == 1, ROWS = FILLM == 2;
    if (FILL && g.enable != nullptr && *g.enable == 0) return;
#ifndef BH_DK_PROBE
#define BH_DK_PROBE 0  // timing probes with WRONG results: bit 0 = no output stores, bit 1 = no LDS reads in the sampler, bit 2 = no LDS-DMA
#endif
#ifndef BH_DK_PIPE
#define BH_DK_PIPE 1
#endif
#ifndef BH_DK_NLOAD
#define BH_DK_NLOAD 2  // loader wavefronts of the 8; 0: every wavefront loads and samples (and waits for its own stores once per tile)
#endif
    constexpr int TX = 64, XC = 256, PITCH = TX + 1, NT = 512, NLOAD = BH_DK_NLOAD;
    constexpr bool SPLIT = NLOAD > 0;
    constexpr int NLD = SPLIT ? NLOAD : NT / 64;           // wavefronts that stage
    constexpr int NWV = SPLIT ? NT / 64 - NLOAD : NT / 64;  // wavefronts that sample
    extern __shared__ __attribute__((aligned(16))) float tile[];  // [2][N][ZC][PITCH]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = SPLIT && wave_all < NLOAD;
    const int wave = loader ? wave_all : wave_all - (SPLIT ? NLOAD : 0);  // index among the loaders / among the samplers
    const int lwave = SPLIT ? wave : wave_all;                            // index among the staging wavefronts
    constexpr int N = NK;
    const int kstride = g.ZC * PITCH;
    const int bufstride = N * kstride;
    const size_t plane = (size_t)g.Y * g.X;
    const long ntiles = (long)g.Za * nxc * ntx;
    const float fN = (float)N, rN = 1.0f / fN;
    double tsum = 0.0;

    struct Tile {
        int a, xo0, xt0, zlo, zcnt;
        bool zero;
    };
    auto tile_of = [&](long t) {
        Tile q;
        const int xt = (int)(t % ntx);
        const long r = t / ntx;
        const int xc = (int)(r % nxc);
        q.a = (int)(r / nxc);
        q.xt0 = xt * TX;
        q.xo0 = xc * XC;
        const int xoN = min(XC, g.Xp - q.xo0);
        const int zo0 = q.a * N;
        const float ix_min = deskew_ix(g.px, g.pxct, g.offset, g.zm1, q.xo0, zo0 + N - 1);
        const float ix_max = deskew_ix(g.px, g.pxct, g.offset, g.zm1, q.xo0 + xoN - 1, zo0);
        q.zlo = (int)floorf(ix_min);
        q.zcnt = min((int)floorf(ix_max) + 2 - q.zlo, g.ZC);
        q.zero = q.zlo + q.zcnt <= 0 || q.zlo >= g.Z;
        return q;
    };
    // an overhang tile: exact zeros everywhere
    auto zero_tile = [&](const Tile& q) {
        if (loader || ROWS) return;
        for (int xl = wave; xl < TX; xl += NWV) {
            const size_t orow_i = (size_t)q.a * g.X + (g.X - 1 - (q.xt0 + xl));
            if (FILL) {
                // 256 x' = four 64-bit mask words: lanes 0..3 write one each
                const int xw0 = q.xo0 + 64 * lane;
                if (lane < 4 && xw0 < g.Xp) {
                    const int nbits = min(64, g.Xp - xw0);
                    const unsigned long long bits = nbits == 64 ? ~0ull : ((1ull << nbits) - 1ull);
                    *reinterpret_cast<unsigned long long*>(g.mask0 + orow_i * g.W32 + xw0 / 32) = bits;
                }
            } else {
                float* orow = out + orow_i * g.Xp;
                const int xo = q.xo0 + 4 * lane;
                typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
                if (xo + 3 < g.Xp) {
                    *reinterpret_cast<f4u*>(orow + xo) = f4u{0.f, 0.f, 0.f, 0.f};
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (xo + j < g.Xp) orow[xo + j] = 0.0f;
                }
            }
        }
    };
    // LDS-DMA of a tile's z window into buffer b: one 256-byte row segment per instruction, rows outside [0, Z) zero-filled
    auto stage = [&](const Tile& q, int b, bool wait = true) {
        if (SPLIT && !loader) return;
        float* base = tile + b * bufstride;
        const int za = max(0, -q.zlo), zb = min(q.zcnt, g.Z - q.zlo);
        const int zo0 = q.a * N;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int yin = g.Y - 1 - min(zo0 + k, g.Y - 1);
            float* dk = base + k * kstride;
            for (int zz = lwave; zz < za; zz += NLD) dk[zz * PITCH + lane] = 0.0f;
            for (int zz = max(zb, 0) + lwave; zz < q.zcnt; zz += NLD) dk[zz * PITCH + lane] = 0.0f;
            const int z0 = za + lwave;
            const float* src = in + (size_t)(q.zlo + z0) * plane + (size_t)yin * g.X + q.xt0 + lane;
            unsigned lds_dst = (unsigned)(size_t)(dk + z0 * PITCH);  // LDS byte address (wave-uniform)
            for (int zz = z0; zz < zb && !(BH_DK_PROBE & 4); zz += NLD) {
                unsigned keep;
                asm volatile(
                    "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                    : "=&s"(keep)
                    : "v"(src), "s"(lds_dst)
                    : "memory");
                src += (size_t)NLD * plane;
                lds_dst += NLD * PITCH * 4;
            }
        }
        // landed before this wavefront reaches the barrier that publishes the buffer (unsplit: waited for at the barrier instead,
        // behind the sampling of the current tile)
        if (SPLIT && wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    // ROWS: the loaders issue the next tile's LDS-DMA first, then their quota of overhang rows, and wait for the DMA alone: loads
    // and stores retire through one in-order counter, so "all but the n youngest" with n = the store instructions issued since
    // (a lower bound of them: waiting for a store or two too many is harmless, for a load too few would not be)
    auto wait_dma_behind = [&](int nstores) {
        if (nstores >= 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
        else if (nstores >= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else if (nstores >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    // next tile of this workgroup that needs staging; overhang tiles on the way are finished on the spot
    // A workgroup walks a CONTIGUOUS range of tiles: x tiles run fastest, so 32 consecutive tiles share their (a, x'-chunk)
    // — one interpolation plan (twelve float32 divisions per lane) serves them all, and neighbouring tiles read neighbouring
    // 256-byte segments of the same input rows.
    const long per_wg = (ntiles + gridDim.x - 1) / gridDim.x;
    const long t_begin = (long)blockIdx.x * per_wg, t_end = min(ntiles, t_begin + per_wg);
    auto advance = [&](long t) {
        for (++t; t < t_end; ++t) {
            const Tile q = tile_of(t);
            if (!q.zero) break;
            zero_tile(q);
        }
        return t;
    };

    // ROWS: the loaders' own walk over the overhang tiles of this workgroup's range (both loaders keep the same cursor and take
    // alternate rows): `emit(quota)` writes up to `quota` rows of 256 x' each — one unaligned 16-byte store per lane
    const float fillv = ROWS ? g.st->fill : 0.0f;
    long zt = t_begin - 1;
    int zrow = TX;
    Tile zq = {};
    // returns how many rows went out as ONE full 16-byte store instruction (a lower bound of the store instructions issued: a row
    // of the ragged last x' chunk issues between none and five)
    auto emit = [&](int quota) -> int {
        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
        int full = 0;
        for (int n = 0; n < quota; ++n) {
            if (zrow >= TX) {
                for (++zt; zt < t_end; ++zt) {
                    zq = tile_of(zt);
                    if (zq.zero) break;
                }
                if (zt >= t_end) {
                    zt = t_end;  // stays exhausted
                    return full;
                }
                zrow = lwave;
            }
            const size_t orow_i = (size_t)zq.a * g.X + (g.X - 1 - (zq.xt0 + zrow));
            float* orow = out + orow_i * g.Xp;
            const int xo = zq.xo0 + 4 * lane;
            if (zq.xo0 + XC <= g.Xp) {  // (wave-uniform) a whole chunk: one instruction
                *reinterpret_cast<f4u*>(orow + xo) = f4u{fillv, fillv, fillv, fillv};
                ++full;
            } else if (xo + 3 < g.Xp) {
                *reinterpret_cast<f4u*>(orow + xo) = f4u{fillv, fillv, fillv, fillv};
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (xo + j < g.Xp) orow[xo + j] = fillv;
            }
            zrow += NLD;
        }
        return full;
    };
#ifndef BH_DK_QUOTA
#define BH_DK_QUOTA 48  // rows per loader and tile interval (the overhang is ~1.2 tiles = 75 rows per sampled tile at config 2); <= 48:
                        // the wait below counts them in vmcnt (6 bits)
#endif
    static_assert(!ROWS || SPLIT, "the one-pass fill needs loader wavefronts");
    float zmin = 1.0f;  // ROWS: smallest |output| outside the geometric zero pattern seen by this lane

    long cur = advance(t_begin - 1);
    int b = 0;
    if (cur < t_end) stage(tile_of(cur), 0);
    int plan_a = -1, plan_xo0 = -1;
    int i0[N][4];
    float w0[N][4], w1[N][4];
    unsigned gnib = 0u, dgnib = 0u;  // ROWS: this lane's four bits of the geometric zero pattern and of its dilation
    while (cur < t_end) {
        const Tile q = tile_of(cur);
        if (!SPLIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // the loaders' DMA of this tile has landed; the samplers have finished with the other buffer
        const long nxt = advance(cur);
        if (ROWS && loader) {
            if (nxt < t_end) stage(tile_of(nxt), b ^ 1, false);
            wait_dma_behind(emit(BH_DK_QUOTA));
        } else if (nxt < t_end) {
            stage(tile_of(nxt), b ^ 1);
        }
        if (loader) {
            cur = nxt;
            b ^= 1;
            continue;
        }
        // ---- sample: lane owns x' = xo0 + 4 lane .. + 3
        const float* tb = tile + b * bufstride;
        const int zo0 = q.a * N;
        const int xb4 = q.xo0 + 4 * lane;
        // interpolation plan of the lane's four outputs x N slices: LDS offset of the lower tap (the upper one is one row on:
        // both arrive in one ds_read2_b32) and the two weights — kept in registers across the tile's 64 rows
        if (q.a != plan_a || q.xo0 != plan_xo0) {  // (wave-uniform) the plan of this (a, x'-chunk): shared by its x tiles
            plan_a = q.a;
            plan_xo0 = q.xo0;
#pragma unroll
            for (int k = 0; k < N; ++k) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float ix = deskew_ix(g.px, g.pxct, g.offset, g.zm1, xb4 + j, zo0 + k);
                    const float fl = floorf(ix);
                    w1[k][j] = ix - fl;
                    w0[k][j] = (fl + 1.0f) - ix;
                    const int rel = max(0, min((int)fl - q.zlo, g.ZC - 2));  // lanes past Xp may fall outside the window
                    i0[k][j] = k * kstride + rel * PITCH;
                }
            }
            if (ROWS) {  // four consecutive x' from a multiple of 4: one nibble of one word
                const int wi = min(xb4, g.Xp - 1) >> 5, sh = xb4 & 31;
                gnib = xb4 < g.Xp ? (g.gbits[(size_t)q.a * g.WB + wi] >> sh) & 15u : 15u;
                dgnib = xb4 < g.Xp ? (g.dgbits[(size_t)q.a * g.WB + wi] >> sh) & 15u : 15u;
            }
        }
        // (wave-uniform) most tiles of the scanned range lie clear of the pattern: no select per output there
        const bool edge_tile = ROWS && __ballot(dgnib != 0u) != 0ull;
        {
            // the taps of the NEXT row are requested before this row is computed and stored (BH_DK_PIPE=0: A/B switch): the
            // sampler's LDS latency is otherwise paid once per row — SQ counters put 38 % of its cycles in waits
            float tp0[N][4], tp1[N][4];
            auto load_taps = [&](int xl_) {
                const float* tc = tb + min(xl_, TX - 1);
#pragma unroll
                for (int k = 0; k < N; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        tp0[k][j] = (BH_DK_PROBE & 2) ? (float)(xl_ + k) : tc[i0[k][j]];
                        tp1[k][j] = (BH_DK_PROBE & 2) ? (float)j : tc[i0[k][j] + PITCH];
                    }
            };
            load_taps(wave);
            for (int xl = wave; xl < TX; xl += NWV) {
                const int yo = g.X - 1 - (q.xt0 + xl);
                const size_t orow_i = (size_t)q.a * g.X + yo;
                float* orow = out + orow_i * g.Xp;
                float acc[4];
                if (!BH_DK_PIPE && xl != wave) load_taps(xl);
#pragma unroll
                for (int k = 0; k < N; ++k) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float val = __builtin_fmaf(tp1[k][j], w1[k][j], tp0[k][j] * w0[k][j]);
                        acc[j] = (k == 0) ? val : acc[j] + val;
                    }
                }
                if (BH_DK_PIPE) load_taps(xl + NWV);  // clamped: the last round re-reads a row of the tile
                float val[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) val[j] = (N > 1) ? div_small(acc[j], fN, rN) : acc[j];
                typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
                if (ROWS) {
                    if (edge_tile) {
                        // outside the geometric pattern an exact zero is a data zero; inside the dilated one the fill value goes out
                        zmin = fminf(zmin, fminf(fminf((gnib & 1u) ? 1.0f : fabsf(val[0]), (gnib & 2u) ? 1.0f : fabsf(val[1])),
                                                 fminf((gnib & 4u) ? 1.0f : fabsf(val[2]), (gnib & 8u) ? 1.0f : fabsf(val[3]))));
#pragma unroll
                        for (int j = 0; j < 4; ++j) val[j] = (dgnib >> j & 1u) ? fillv : val[j];
                    } else {
                        zmin = fminf(zmin, fminf(fminf(fabsf(val[0]), fabsf(val[1])), fminf(fabsf(val[2]), fabsf(val[3]))));
                    }
                    if (BH_DK_PROBE & 1) {
                        if (val[0] + val[1] + val[2] + val[3] == -12345.0f) orow[xb4] = 1.0f;
                    } else if (xb4 + 3 < g.Xp) {
                        *reinterpret_cast<f4u*>(orow + xb4) = f4u{val[0], val[1], val[2], val[3]};
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (xb4 + j < g.Xp) orow[xb4 + j] = val[j];
                    }
                } else if (FILL) {
                    // zero-mask bits of the 256 outputs: word w (x' = xo0 + 64 w .. + 63) is the nibbles of lanes 16 w .. 16 w + 15,
                    // lane 16 w + i contributing bits 4 i .. 4 i + 3.  A DPP row is those 16 lanes: each lane places its nibble
                    // in the low (i < 8) or high half, four row_shr steps OR the row together, lane 16 w + 15 writes the word.
                    const unsigned nib = (unsigned)(xb4 < g.Xp && val[0] == 0.0f) | ((unsigned)(xb4 + 1 < g.Xp && val[1] == 0.0f) << 1) |
                                         ((unsigned)(xb4 + 2 < g.Xp && val[2] == 0.0f) << 2) |
                                         ((unsigned)(xb4 + 3 < g.Xp && val[3] == 0.0f) << 3);
                    const int i16 = lane & 15;
                    unsigned lo = 0u, hi = 0u;
                    // most rows of the scanned range hold no exact zero at all: the words are then 0 without the row reduction
                    // (a wave-uniform branch; BH_DK_MASK_SKIP=0: A/B switch)
#ifndef BH_DK_MASK_SKIP
#define BH_DK_MASK_SKIP 1
#endif
                    if (!BH_DK_MASK_SKIP || __ballot(nib != 0u) != 0ull) {
                        lo = i16 < 8 ? nib << (4 * i16) : 0u;
                        hi = i16 < 8 ? 0u : nib << (4 * (i16 - 8));
#define BH_ROW_OR(v, sh) v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + sh, 0xF, 0xF, true)  /* row_shr:sh, 0 shifted in */
                        BH_ROW_OR(lo, 1);
                        BH_ROW_OR(hi, 1);
                        BH_ROW_OR(lo, 2);
                        BH_ROW_OR(hi, 2);
                        BH_ROW_OR(lo, 4);
                        BH_ROW_OR(hi, 4);
                        BH_ROW_OR(lo, 8);
                        BH_ROW_OR(hi, 8);
#undef BH_ROW_OR
                    }
                    const int xw0 = q.xo0 + 64 * (lane >> 4);
                    if (i16 == 15 && xw0 < g.Xp)
                        *reinterpret_cast<uint2*>(g.mask0 + orow_i * g.W32 + xw0 / 32) = make_uint2(lo, hi);
                    // exact zeros are not stored (the fill pass overwrites every masked voxel anyway); a group without zeros
                    // inside the row goes out as one 16-byte store
                    if (BH_DK_PROBE & 1) {
                        tsum += (double)(val[0] + val[1] + val[2] + val[3]);
                    } else if (xb4 + 3 < g.Xp && nib == 0u) {
                        *reinterpret_cast<f4u*>(orow + xb4) = f4u{val[0], val[1], val[2], val[3]};
                        tsum += (double)val[0] + (double)val[1] + (double)val[2] + (double)val[3];
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (xb4 + j < g.Xp && val[j] != 0.0f) {
                                orow[xb4 + j] = val[j];
                                tsum += (double)val[j];
                            }
                    }
                } else {
                    if (BH_DK_PROBE & 1) {
                        if (val[0] + val[1] + val[2] + val[3] == -12345.0f) orow[xb4] = 1.0f;  // keeps the arithmetic alive
                    } else if (xb4 + 3 < g.Xp) {
                        *reinterpret_cast<f4u*>(orow + xb4) = f4u{val[0], val[1], val[2], val[3]};
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (xb4 + j < g.Xp) orow[xb4 + j] = val[j];
                    }
                }
            }
        }
        cur = nxt;
        b ^= 1;
    }
    if (ROWS) {
        if (loader) emit(1 << 30);  // what is left of the overhang tiles
        if (!loader && __ballot(zmin == 0.0f) != 0ull && lane == 0) atomicOr(&g.st->fallback, 1);
    }
    if (FILL) {
        __shared__ double wsum[NT / 64];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tsum += __shfl_down(tsum, o, 64);
        if (lane == 0) wsum[wave_all] = tsum;  // the loaders contribute 0
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < NT / 64; ++w) t += wsum[w];
            g.psum[blockIdx.x] = t;
        }
    }
}

#include "deskew_rows.inc"

static int deskew_geometry(int64_t Z, int64_t Y, int64_t X, double angle, double ratio, int keep_overhang,
                           int n, DeskewGeom* g, int64_t out_shape[3]) {
    double voxel[3];
    BH_TRY(bh_deskew_shape(Z, Y, X, angle, ratio, keep_overhang, n, 1.0, out_shape, voxel));
    // un-averaged geometry drives the shear offset (deskew.py:499-503: Z_out_full = Y)
    const double ct = std::cos(angle * M_PI / 180.0);
    const double px = ratio;
    const int64_t Xp = out_shape[2];
    const double offset = px * ct * (double)(Y - 1) / 2 - px * (double)(Xp - 1) / 2 + (double)(Z - 1) / 2;
    g->Z = (int)Z;
    g->Y = (int)Y;
    g->X = (int)X;
    g->Za = (int)out_shape[0];
    g->Xp = (int)Xp;
    g->N = n;
    g->px = (float)px;
    g->pxct = (float)(px * ct);
    g->offset = (float)offset;
    g->zm1 = (float)(Z - 1);
    return BH_OK;
}

// Exact maximum z-window over every (a, chunk) for a candidate XC, using the device formula.
static int max_window(const DeskewGeom& g, int XC) {
    int worst = 0;
    for (int a = 0; a < g.Za; ++a) {
        const int zo0 = a * g.N;
        for (int xo0 = 0; xo0 < g.Xp; xo0 += XC) {
            const int xoN = std::min(XC, g.Xp - xo0);
            const float lo = deskew_ix(g.px, g.pxct, g.offset, g.zm1, xo0, zo0 + g.N - 1);
            const float hi = deskew_ix(g.px, g.pxct, g.offset, g.zm1, xo0 + xoN - 1, zo0);
            const int cnt = (int)std::floor(hi) + 2 - (int)std::floor(lo);
            worst = std::max(worst, cnt);
        }
    }
    return worst;
}

template <typename TIN, int TX, int J, int NT, bool DMA>
static int launch_deskew_cfg(bh_ctx* ctx, const TIN* in, float* out, DeskewGeom g, int fill, int* nblocks) {  // fill: FILLM
    constexpr int XC = 64 * J;
    g.XC = XC;
    g.ZC = max_window(g, XC);
    g.ZS = TX + 1;
    const size_t lds = (size_t)g.N * g.ZC * (TX + 1) * sizeof(float);
    BH_REQUIRE(lds <= 160 * 1024,
               "deskew tile needs %zu bytes of LDS (px_to_scan_ratio=%g, average_n_slices=%d) — exceeds 160 KiB",
               lds, (double)g.px, g.N);
    dim3 grid((unsigned)ceil_div(g.X, TX), (unsigned)ceil_div(g.Xp, XC), (unsigned)g.Za);
    BH_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "deskew grid too large (%u,%u,%u)", grid.x, grid.y, grid.z);
    const size_t nblk = (size_t)grid.x * grid.y * grid.z;
    if (nblocks) *nblocks = (int)nblk;
    if (fill == 1) {
        BH_REQUIRE(nblk < (1ull << 31), "deskew grid too large for the fused fill");
        BH_TRY(get_scratch(ctx, "fill_pall", nblk * sizeof(double), (void**)&g.psum));
    }
    auto run = [&](auto kern) -> int {
        if (lds > 64 * 1024)
            BH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, grid, dim3(NT), lds, ctx->stream, in, out, g);
        BH_CHECK_HIP(hipGetLastError());
        return BH_OK;
    };
#define BH_DK_CASE(NKV)                                                                    \
    return fill == 2 ? run(deskew_kernel<TIN, TX, J, NKV, NT, DMA, 2>)                      \
                     : (fill == 1 ? run(deskew_kernel<TIN, TX, J, NKV, NT, DMA, 1>) : run(deskew_kernel<TIN, TX, J, NKV, NT, DMA, 0>))
    switch (g.N) {
        case 1: BH_DK_CASE(1);
        case 2: BH_DK_CASE(2);
        case 3: BH_DK_CASE(3);
        case 4: BH_DK_CASE(4);
        default: BH_DK_CASE(0);
    }
#undef BH_DK_CASE
}

// LDS bytes of a (TX, J) configuration for this geometry
static size_t cfg_lds(const DeskewGeom& g, int TX, int J) {
    return (size_t)g.N * max_window(g, 64 * J) * (TX + 1) * sizeof(float);
}

// the persistent double-buffered kernel: float32 input, whole 64-column tiles, N <= 4, two tile buffers within 160 KiB
template <typename TIN>
static int launch_deskew_pers(bh_ctx* ctx, const TIN* in, float* out, DeskewGeom g, bool fill, int* nblocks, bool* taken) {
    *taken = false;
    return BH_OK;
}
template <>
int launch_deskew_pers<float>(bh_ctx* ctx, const float* in, float* out, DeskewGeom g, bool fill, int* nblocks, bool* taken) {
    *taken = false;
    // BH_DESKEW_PERS: 0 never, 1 always (when the shape allows), default: with a fused fill only — measured at config 2
    // (tools/time_deskew.py, same box): 4.96 against 5.59 ms with the fill prologue, 5.89 against 5.60 ms without
    const int mode = getenv("BH_DESKEW_PERS") ? atoi(getenv("BH_DESKEW_PERS")) : 2;
    if (mode == 0 || (mode == 2 && !fill)) return BH_OK;
    if (g.N < 1 || g.N > 4 || (g.X % 64) != 0) return BH_OK;
    constexpr int XC = 256, TX = 64;
    g.XC = XC;
    g.ZC = std::max(max_window(g, XC), 3);
    g.ZS = TX + 1;
    const size_t lds = 2 * (size_t)g.N * g.ZC * (TX + 1) * sizeof(float);
    if (lds + 256 > 160 * 1024) return BH_OK;
    const int ntx = g.X / TX, nxc = (int)ceil_div(g.Xp, XC);
    const long ntiles = (long)g.Za * nxc * ntx;
    const int grid = (int)std::min<long>(ntiles, ctx->num_cus);
    if (nblocks) *nblocks = grid;
    if (fill) BH_TRY(get_scratch(ctx, "fill_pall", (size_t)grid * sizeof(double), (void**)&g.psum));
    auto run = [&](auto kern) -> int {
        BH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, ctx->stream, in, out, g, ntx, nxc);
        BH_CHECK_HIP(hipGetLastError());
        return BH_OK;
    };
    *taken = true;
    switch (g.N) {
        case 1: return fill ? run(deskew_pers_kernel<1, 1>) : run(deskew_pers_kernel<1, 0>);
        case 2: return fill ? run(deskew_pers_kernel<2, 1>) : run(deskew_pers_kernel<2, 0>);
        case 3: return fill ? run(deskew_pers_kernel<3, 1>) : run(deskew_pers_kernel<3, 0>);
        default: return fill ? run(deskew_pers_kernel<4, 1>) : run(deskew_pers_kernel<4, 0>);
    }
}

// the persistent kernel in its one-pass form (g.gbits / g.dgbits / g.st set by launch_deskew_rows)
static int launch_deskew_rows_pers(bh_ctx* ctx, const float* in, float* out, DeskewGeom g, bool* taken) {
    *taken = false;
    if (g.N < 1 || g.N > 4 || (g.X % 64) != 0) return BH_OK;
    constexpr int XC = 256, TX = 64;
    g.XC = XC;
    g.ZC = std::max(max_window(g, XC), 3);
    g.ZS = TX + 1;
    const size_t lds = 2 * (size_t)g.N * g.ZC * (TX + 1) * sizeof(float);
    if (lds + 256 > 160 * 1024) return BH_OK;
    const int ntx = g.X / TX, nxc = (int)ceil_div(g.Xp, XC);
    const long ntiles = (long)g.Za * nxc * ntx;
    const int grid = (int)std::min<long>(ntiles, ctx->num_cus);
    auto run = [&](auto kern) -> int {
        BH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, ctx->stream, in, out, g, ntx, nxc);
        BH_CHECK_HIP(hipGetLastError());
        return BH_OK;
    };
    *taken = true;
    switch (g.N) {
        case 1: return run(deskew_pers_kernel<1, 2>);
        case 2: return run(deskew_pers_kernel<2, 2>);
        case 3: return run(deskew_pers_kernel<3, 2>);
        default: return run(deskew_pers_kernel<4, 2>);
    }
}

template <typename TIN>
static int launch_deskew(bh_ctx* ctx, const TIN* in, float* out, DeskewGeom g, int fill, int* nblocks) {  // fill: FILLM of the kernels
    if (fill != 2) {
        bool taken = false;
        BH_TRY(launch_deskew_pers<TIN>(ctx, in, out, g, fill != 0, nblocks, &taken));
        if (taken) return BH_OK;
    }
    // Candidates (TX, J, threads, LDS-DMA staging) from fastest measured (profiles/, tools/tune_deskew.py:
    // 5.7 ms at 512x2048x2048 -> 683x2048x3034 for cfg 0) to smallest tile; take the first whose tile
    // lets two workgroups share a CU, else the first that fits.  BH_DESKEW_CFG=<n> forces one.
    int force = -1;
    if (const char* e = getenv("BH_DESKEW_CFG")) force = atoi(e);
    const size_t two_per_cu = 80 * 1024;
    constexpr int NC = 5;
    const int cand[NC][2] = {{64, 4}, {64, 2}, {32, 4}, {64, 1}, {32, 1}};
    // The one-pass fill (fill == 2) writes the overhang chunks from the same launch — workgroups that never touch their LDS —
    // so the smaller tile (two outputs per lane and row: half the LDS, twice the workgroups per CU) comes first there:
    // 7.2 -> 6.6 ms for the standalone pair at config 2 (tools/time_deskew.py with BH_DESKEW_CFG=0 / 1).
    const int order_rows[NC] = {1, 0, 2, 3, 4}, order_std[NC] = {0, 1, 2, 3, 4};
    const int* order = fill == 2 ? order_rows : order_std;
    int pick = -1;
    if (force >= 0 && force < NC) pick = force;
    for (int k = 0; k < NC && pick < 0; ++k)
        if (cfg_lds(g, cand[order[k]][0], cand[order[k]][1]) <= two_per_cu) pick = order[k];
    for (int k = 0; k < NC && pick < 0; ++k)
        if (cfg_lds(g, cand[order[k]][0], cand[order[k]][1]) <= 160 * 1024) pick = order[k];
    if (pick < 0) pick = NC - 1;
    switch (pick) {
        case 0: return launch_deskew_cfg<TIN, 64, 4, 256, true>(ctx, in, out, g, fill, nblocks);
        case 1: return launch_deskew_cfg<TIN, 64, 2, 256, true>(ctx, in, out, g, fill, nblocks);
        case 2: return launch_deskew_cfg<TIN, 32, 4, 256, false>(ctx, in, out, g, fill, nblocks);
        case 3: return launch_deskew_cfg<TIN, 64, 1, 256, true>(ctx, in, out, g, fill, nblocks);
        default: return launch_deskew_cfg<TIN, 32, 1, 256, false>(ctx, in, out, g, fill, nblocks);
    }
}

// The one-pass fill (deskew_rows.inc): geometry bits, the fill value from the row sums of the input, then ONE resampling kernel
// that writes whole rows — the tile kernel (any dtype, any shape) or the persistent kernel (float32, whole 64-column tiles,
// N <= 4: BH_DESKEW_ROWS_KERNEL=pers; measured slower at config 2, 6.3 against 5.2 ms: two loader wavefronts per CU write the
// 9 GB of overhang tiles there, whole workgroups here).  row_sums: optional float64 [Z * Y] row sums of `in` on the device (an
// operator that has just produced `in` can hand them over), else they are reduced here in one read of `in`.
template <typename TIN>
static int launch_deskew_rows(bh_ctx* ctx, const TIN* in, float* out, DeskewGeom g, int fill_mode, float fill_value,
                              const double* row_sums, FillStats* st, bool* taken) {
    *taken = false;
    if (getenv("BH_DESKEW_ONEPASS") && atoi(getenv("BH_DESKEW_ONEPASS")) == 0) return BH_OK;
    hipStream_t s = ctx->stream;
    const int WB = (int)ceil_div(g.Xp, 32);
    uint32_t *gb, *dgb;
    double* psum;
    unsigned long long* pcnt;
    BH_TRY(get_scratch(ctx, "dk_gbits", (size_t)g.Za * WB * 4, (void**)&gb));
    BH_TRY(get_scratch(ctx, "dk_dgbits", (size_t)g.Za * WB * 4, (void**)&dgb));
    BH_TRY(get_scratch(ctx, "dk_psum", (size_t)g.Za * sizeof(double), (void**)&psum));
    BH_TRY(get_scratch(ctx, "dk_pcnt", (size_t)g.Za * sizeof(unsigned long long), (void**)&pcnt));
    const int nw = g.Za * WB;
    hipLaunchKernelGGL(rows::geom_bits_kernel<0>, dim3((unsigned)ceil_div(nw, 256)), dim3(256), 0, s, g, gb, WB);
    hipLaunchKernelGGL(rows::dilate_bits_kernel, dim3((unsigned)ceil_div(nw, 256)), dim3(256), 0, s, gb, dgb, g.Za, g.Xp, WB, 3);
    if (fill_mode == BH_FILL_MEAN) {
        if (row_sums == nullptr) {
            double* R;
            BH_TRY(get_scratch(ctx, "dk_rowsums", (size_t)g.Z * g.Y * sizeof(double), (void**)&R));
            hipLaunchKernelGGL(rows::row_sums_kernel<TIN>, dim3((unsigned)(ctx->num_cus * 8)), dim3(256), 0, s, in, R, (long)g.Z * g.Y, g.X);
            row_sums = R;
        }
        hipLaunchKernelGGL(rows::mean_partial_kernel, dim3((unsigned)g.Za), dim3(256), 0, s, g, dgb, WB, row_sums, psum, pcnt);
    }
    hipLaunchKernelGGL(rows::mean_final_kernel, dim3(1), dim3(256), 0, s, psum, pcnt, g.Za, g.N, (long)g.X,
                       (long)g.Za * g.X * g.Xp, fill_mode, fill_value, st);
    BH_CHECK_HIP(hipGetLastError());
    g.gbits = gb;
    g.dgbits = dgb;
    g.WB = WB;
    g.st = st;
    *taken = true;
    const char* kern = getenv("BH_DESKEW_ROWS_KERNEL");
    if (kern && strcmp(kern, "pers") == 0 && std::is_same<TIN, float>::value) {
        bool pers = false;
        BH_TRY(launch_deskew_rows_pers(ctx, reinterpret_cast<const float*>(in), out, g, &pers));
        if (pers) return BH_OK;
    }
    return launch_deskew(ctx, in, out, g, 2, nullptr);
}

int fill_overhang_impl(bh_ctx* ctx, float* data, int64_t Z, int64_t Y, int64_t X, int fill_mode,
                       float fill_value, int iterations, float* mean_out, int fused_partials, int connectivity, const int* enable);
int fill_mask_buffers(bh_ctx* ctx, int64_t rows, int64_t X, uint32_t** m0, int* W32);

}  // namespace bh

extern "C" {

int bh_deskew_shape(int64_t Z, int64_t Y, int64_t X, double ls_angle_deg, double px_to_scan_ratio,
                    int keep_overhang, int average_n_slices, double pixel_size_um, int64_t out_shape[3],
                    double voxel_size[3]) {
    BH_REQUIRE(out_shape != nullptr && voxel_size != nullptr, "NULL output argument");
    BH_REQUIRE(Z > 0 && Y > 0 && X > 0, "raw_data_shape must be positive, got (%lld,%lld,%lld)", (long long)Z,
               (long long)Y, (long long)X);
    BH_REQUIRE(average_n_slices >= 1, "average_n_slices must be >= 1, got %d", average_n_slices);
    BH_REQUIRE(px_to_scan_ratio > 0, "px_to_scan_ratio must be > 0, got %g", px_to_scan_ratio);
    const double theta = ls_angle_deg * M_PI / 180.0;
    const double st = std::sin(theta), ct = std::cos(theta);
    long long Xp;
    if (keep_overhang) {
        Xp = (long long)std::ceil(((double)Z / px_to_scan_ratio) + ((double)Y * ct));
    } else {
        Xp = (long long)std::ceil(((double)Z / px_to_scan_ratio) - ((double)Y * ct));
        // message text mirrors biahub/deskew.py:263-267
        BH_REQUIRE(Xp > 0,
                   "Dataset contains only overhang when keep_overhang=False. Computed Xp=%lld <= 0. Either set "
                   "keep_overhang=True or use a dataset with non-overhang content.",
                   Xp);
    }
    out_shape[0] = (Y + average_n_slices - 1) / average_n_slices;
    out_shape[1] = X;
    out_shape[2] = Xp;
    voxel_size[0] = average_n_slices * st * pixel_size_um;
    voxel_size[1] = pixel_size_um;
    voxel_size[2] = pixel_size_um;
    return BH_OK;
}

int bh_deskew(bh_ctx* ctx, const void* in, int in_dtype, int64_t Z, int64_t Y, int64_t X, double ls_angle_deg,
              double px_to_scan_ratio, int keep_overhang, int average_n_slices, int fill_mode, float fill_value,
              float* out, float* mean_out) {
    return bh_deskew_rows(ctx, in, in_dtype, Z, Y, X, ls_angle_deg, px_to_scan_ratio, keep_overhang, average_n_slices, fill_mode,
                          fill_value, out, mean_out, nullptr);
}

int bh_deskew_rows(bh_ctx* ctx, const void* in, int in_dtype, int64_t Z, int64_t Y, int64_t X, double ls_angle_deg,
                   double px_to_scan_ratio, int keep_overhang, int average_n_slices, int fill_mode, float fill_value,
                   float* out, float* mean_out, const double* row_sums) {
    BH_REQUIRE(ctx != nullptr && in != nullptr && out != nullptr, "NULL argument");
    BH_REQUIRE(Z >= 2, "deskew needs at least 2 scan slices, got Z=%lld", (long long)Z);
    BH_REQUIRE(Z < (1 << 24) && Y < (1 << 24) && X < (1ll << 31), "volume too large for float32 coordinates");
    BH_REQUIRE(fill_mode >= BH_FILL_NONE && fill_mode <= BH_FILL_MEAN, "unknown fill_mode %d", fill_mode);
    bh::DeskewGeom g;
    int64_t os[3];
    BH_TRY(bh::deskew_geometry(Z, Y, X, ls_angle_deg, px_to_scan_ratio, keep_overhang, average_n_slices, &g, os));
    BH_REQUIRE(os[2] < (1 << 24), "deskewed X extent %lld too large", (long long)os[2]);
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    // reference :538 — fill only when keep_overhang and (fill == "mean" or fill != 0)
    const bool do_fill = keep_overhang && (fill_mode == BH_FILL_MEAN || (fill_mode == BH_FILL_CONSTANT && fill_value != 0.0f));
    int nblocks = 0;
    g.enable = nullptr;
    g.gbits = g.dgbits = nullptr;
    g.WB = 0;
    g.st = nullptr;
    if (do_fill) BH_TRY(bh::fill_mask_buffers(ctx, os[0] * os[1], os[2], &g.mask0, &g.W32));
    // One-pass fill (float32): the fill value from row sums of the input, whole rows written by the resampling kernel.  The mask
    // pipeline is queued behind it CONDITIONALLY (a device flag the kernel raises when it meets an exact zero that geometry
    // does not explain): every kernel of it returns at once otherwise.
    bool onepass = false;
    bh::FillStats* st = nullptr;
    if (do_fill) {
        BH_TRY(bh::get_scratch(ctx, "fill_stats", sizeof(bh::FillStats), (void**)&st));
        bh::ScopedTimer t(ctx, bh::T_DESKEW);
        switch (in_dtype) {
            case BH_DT_F32: BH_TRY(bh::launch_deskew_rows(ctx, (const float*)in, out, g, fill_mode, fill_value, row_sums, st, &onepass)); break;
            case BH_DT_U16: BH_TRY(bh::launch_deskew_rows(ctx, (const uint16_t*)in, out, g, fill_mode, fill_value, row_sums, st, &onepass)); break;
            case BH_DT_U8: BH_TRY(bh::launch_deskew_rows(ctx, (const uint8_t*)in, out, g, fill_mode, fill_value, row_sums, st, &onepass)); break;
            case BH_DT_I16: BH_TRY(bh::launch_deskew_rows(ctx, (const int16_t*)in, out, g, fill_mode, fill_value, row_sums, st, &onepass)); break;
            default: BH_REQUIRE(false, "unsupported input dtype code %d", in_dtype);
        }
    }
    if (onepass) g.enable = &st->fallback;
    ctx->deskew_path = onepass ? 1 : 0;
    if (!onepass) {
        {
            bh::ScopedTimer t(ctx, bh::T_DESKEW);
            switch (in_dtype) {
                case BH_DT_F32: BH_TRY(bh::launch_deskew(ctx, (const float*)in, out, g, do_fill ? 1 : 0, &nblocks)); break;
                case BH_DT_U16: BH_TRY(bh::launch_deskew(ctx, (const uint16_t*)in, out, g, do_fill ? 1 : 0, &nblocks)); break;
                case BH_DT_U8: BH_TRY(bh::launch_deskew(ctx, (const uint8_t*)in, out, g, do_fill ? 1 : 0, &nblocks)); break;
                case BH_DT_I16: BH_TRY(bh::launch_deskew(ctx, (const int16_t*)in, out, g, do_fill ? 1 : 0, &nblocks)); break;
                default: BH_REQUIRE(false, "unsupported input dtype code %d", in_dtype);
            }
        }
        if (do_fill) {
            // the deskew kernel already produced the zero mask and the block sums (and skipped storing zeros)
            BH_TRY(bh::fill_overhang_impl(ctx, out, os[0], os[1], os[2], fill_mode, fill_value, 3, mean_out, nblocks, 26, nullptr));
        } else if (mean_out) {
            *mean_out = 0.0f;
        }
    } else {
        // T_FILL times the conditional pass: a dozen launches that return at once unless the flag is up
        bh::ScopedTimer t(ctx, bh::T_FILL);
        switch (in_dtype) {
            case BH_DT_F32: BH_TRY(bh::launch_deskew(ctx, (const float*)in, out, g, 1, &nblocks)); break;
            case BH_DT_U16: BH_TRY(bh::launch_deskew(ctx, (const uint16_t*)in, out, g, 1, &nblocks)); break;
            case BH_DT_U8: BH_TRY(bh::launch_deskew(ctx, (const uint8_t*)in, out, g, 1, &nblocks)); break;
            default: BH_TRY(bh::launch_deskew(ctx, (const int16_t*)in, out, g, 1, &nblocks)); break;
        }
        BH_TRY(bh::fill_overhang_impl(ctx, out, os[0], os[1], os[2], fill_mode, fill_value, 3, mean_out, nblocks, 26, g.enable));
    }
    return BH_OK;
}

int bh_deskew_fill_path(bh_ctx* ctx, int* path) {
    BH_REQUIRE(ctx != nullptr && path != nullptr, "NULL argument");
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    auto it = ctx->scratch.find("fill_stats");
    *path = 0;
    if (ctx->deskew_path == 0 || it == ctx->scratch.end() || it->second.ptr == nullptr) return BH_OK;
    bh::FillStats h;
    BH_CHECK_HIP(hipMemcpyAsync(&h, it->second.ptr, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    BH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    *path = h.fallback ? 2 : 1;
    return BH_OK;
}

}  // extern "C"
