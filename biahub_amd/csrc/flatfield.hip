// Flat-field correction on gfx950: exact per-pixel median along Z, then divide the static pattern out.
//
//   flat_field_zyx(zyx) = zyx / median(zyx, axis=0) * mean(median(zyx, axis=0))      (biahub/flat_field.py:101-120)
//
// The reference's cost is the median: np.median partitions along an axis whose stride is a whole plane, 5-130 CPU-seconds
// per position depending on the host (flat_field.py:56-99 tiles it to stay in cache).  Here a workgroup stages all Z
// samples of 64 neighbouring pixels in LDS (one coalesced read of the volume) and every lane finds its own column's order
// statistic by a most-significant-bit-first search on order-preserving integer keys: 16 counting sweeps over LDS for
// uint16 camera data, 32 for float32 — no sorting, no data-dependent branches, and the result is the exact element
// np.median picks (for an even count the mean of the two middle elements, computed in the reference's type).
// The apply pass re-reads the volume once: out = float32(double(v) / pattern * mean) for integer input (the reference's
// float64 expression, cast by _flat_field_czyx :154), float32 arithmetic for float32 input.
#include "common.hpp"

#include <algorithm>
#include <cmath>

namespace bh {


template <typename T>
struct FfKey;
template <>
struct FfKey<uint8_t> {
    using key_t = uint16_t;
    static constexpr int bits = 8;
    static __device__ key_t to(uint8_t v) { return v; }
    static __device__ double from(unsigned k) { return (double)k; }
};
template <>
struct FfKey<uint16_t> {
    using key_t = uint16_t;
    static constexpr int bits = 16;
    static __device__ key_t to(uint16_t v) { return v; }
    static __device__ double from(unsigned k) { return (double)k; }
};
template <>
struct FfKey<int16_t> {
    using key_t = uint16_t;
    static constexpr int bits = 16;
    static __device__ key_t to(int16_t v) { return (uint16_t)((uint16_t)v ^ 0x8000u); }
    static __device__ double from(unsigned k) { return (double)(int16_t)(uint16_t)(k ^ 0x8000u); }
};
template <>
struct FfKey<float> {
    using key_t = uint32_t;
    static constexpr int bits = 32;
    static __device__ key_t to(float v) {
        const uint32_t b = __float_as_uint(v);
        return (b & 0x80000000u) ? ~b : (b | 0x80000000u);  // total order of finite floats (-0 < +0)
    }
    static __device__ double from(unsigned k) {
        const uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
        return (double)__uint_as_float(b);
    }
};

// One workgroup = `cols` neighbouring pixels of one image row, all Z samples in LDS as keys[z][cols]; thread t owns column
// t % cols and the z-subset {g, g + groups, ...}, g = t / cols; per search step the groups' counts meet in LDS.
template <typename TIN, int FF_NT>
__global__ __launch_bounds__(FF_NT) void median_z_kernel(const TIN* __restrict__ in, int Z, int Y, int X, int cols,
                                                         int tiles_x, double* __restrict__ pattern) {
    using K = FfKey<TIN>;
    using key_t = typename K::key_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    key_t* keys = reinterpret_cast<key_t*>(smem);
    unsigned* part = reinterpret_cast<unsigned*>(smem + (((size_t)Z * cols * sizeof(key_t) + 15) & ~(size_t)15));  // [2][FF_NT]
    const int groups = FF_NT / cols, col = threadIdx.x % cols, grp = threadIdx.x / cols;
    const size_t plane = (size_t)Y * X;
    for (int tile = blockIdx.x; tile < tiles_x * Y; tile += gridDim.x) {
        const int y = tile / tiles_x, x0 = (tile - y * tiles_x) * cols;
        __syncthreads();  // the previous tile's readers are done
        const TIN* src = in + (size_t)y * X + min(x0 + col, X - 1);
        const int nz = (Z - grp + groups - 1) / groups;  // this thread's samples: z = grp + j * groups
        const int zs = groups * cols;                    // their distance in the LDS image
#pragma unroll 8
        for (int j = 0; j < nz; ++j) keys[(grp + j * groups) * cols + col] = K::to(src[(size_t)(grp + j * groups) * plane]);
        __syncthreads();
        const key_t* mine = keys + grp * cols + col;
        const int k = (Z - 1) >> 1;  // lower middle element (0-based)
        unsigned res = 0;
        int buf = 0;
        auto total = [&](unsigned mine) -> unsigned {  // sum over the groups of this column
            unsigned* p = part + buf * FF_NT;
            p[threadIdx.x] = mine;
            __syncthreads();
            unsigned s = 0;
            for (int g = 0; g < groups; ++g) s += p[g * cols + col];
            buf ^= 1;  // the other buffer is only rewritten after the next barrier
            return s;
        };
        for (int bit = K::bits - 1; bit >= 0; --bit) {
            const unsigned cand = res | (1u << bit);
            unsigned c = 0;
#pragma unroll 8
            for (int j = 0; j < nz; ++j) c += (unsigned)(mine[j * zs] < cand);
            if (total(c) <= (unsigned)k) res = cand;  // largest r with #(key < r) <= k is the k-th smallest key
        }
        unsigned hi = res;
        if ((Z & 1) == 0) {  // even count: the upper middle element is res again, or the smallest key above it
            unsigned cle = 0, mn = 0xffffffffu;
#pragma unroll 8
            for (int j = 0; j < nz; ++j) {
                const unsigned v = mine[j * zs];
                cle += (unsigned)(v <= res);
                mn = v > res ? min(mn, v) : mn;
            }
            const unsigned tle = total(cle);
            unsigned* p = part + buf * FF_NT;
            p[threadIdx.x] = mn;
            __syncthreads();
            for (int g = 0; g < groups; ++g) mn = min(mn, p[g * cols + col]);
            buf ^= 1;
            hi = tle > (unsigned)(k + 1) ? res : mn;
        }
        if (grp == 0 && x0 + col < X) {
            double m;
            if (sizeof(TIN) == 4)
                m = (double)(((float)K::from(res) + (float)K::from(hi)) * 0.5f);  // np.median of float32 stays float32
            else
                m = (K::from(res) + K::from(hi)) * 0.5;
            pattern[(size_t)y * X + x0 + col] = (Z & 1) ? K::from(res) : m;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 8 / 16-bit input (camera data): range-adaptive radix selection with per-pixel histograms instead of 16 bit sweeps.
// A workgroup owns 128 neighbouring pixels of an image row; lane l of every wave owns pixels 2l and 2l + 1 and counts
// them in the two 16-bit halves of hist[bin][l] (256 x 64 words = 64 KiB; Z <= 65535 so a half never overflows).  The
// bank of a counter is its lane, so the LDS atomics are conflict-free whatever the data.
//   pass 0 (no atomics): per pixel the smallest and largest key over z -> base mn and shift s = the fewest low bits to drop
//           so that (key - mn) >> s fits 256 bins.  Camera counts of one pixel along z span a few hundred levels: s = 0.
//   pass 1: histogram of (key - mn) >> s; per pixel the bins holding the two middle ranks.  With s = 0 a bin IS a value: done —
//           ONE atomic per sample instead of the two of a fixed high-byte / low-byte split (the pass is bound by its LDS
//           atomics: ~32 cycles per wave instruction; round 3: 3.8 ms at config-2 size).
//   pass 2 (only when some pixel of the tile has s > 0; workgroup-uniform): histogram of the dropped low bits of the samples
//           in the first bin (and the smallest low part of the second bin, for the pixel whose middle samples straddle bins).
// The re-reads of the 128 columns are L2 hits.
constexpr int FH_NT = 512, FH_COLS = 128, FH_NW = FH_NT / 64, FH_U = 16;  // 2 workgroups x 8 waves x 16 row loads in flight per CU

template <typename TIN>
__global__ __launch_bounds__(FH_NT) void median_hist_kernel(const TIN* __restrict__ in, int Z, int Y, int X, int tiles_x,
                                                            double* __restrict__ pattern) {
    using K = FfKey<TIN>;
    __shared__ __attribute__((aligned(16))) unsigned hist[256 * 64];
    __shared__ unsigned short sb1[FH_COLS], sb2[FH_COLS], sr1[FH_COLS];
    __shared__ unsigned minlo[FH_COLS];
    __shared__ unsigned wmn[FH_NW][FH_COLS], wmx[FH_NW][FH_COLS];
    __shared__ int any_shift;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t plane = (size_t)Y * X;
    const int k = (Z - 1) >> 1, k2 = Z >> 1;  // the two middle ranks (equal for odd Z)
    const bool paired = (X & 1) == 0;          // pixel pairs are naturally aligned: one load per pair
    int streak = 0, skip = 0;  // consecutive probed tiles that were wide; tiles to run without a probe
    for (int tile = blockIdx.x; tile < tiles_x * Y; tile += gridDim.x) {
        const int y = tile / tiles_x, x0 = (tile - y * tiles_x) * FH_COLS;
        const int ca = min(x0 + 2 * lane, paired ? X - 2 : X - 1), cb = min(ca + 1, X - 1);
        const TIN* src = in + (size_t)y * X;
        auto load2 = [&](int z, unsigned& ka, unsigned& kb) {
            const TIN* row = src + (size_t)z * plane;
            if (paired) {
                struct alignas(2 * sizeof(TIN)) Pair { TIN a, b; };
                const Pair v = *reinterpret_cast<const Pair*>(row + ca);
                ka = K::to(v.a), kb = K::to(v.b);
            } else {
                ka = K::to(row[ca]), kb = K::to(row[cb]);
            }
        };
        for (int i = threadIdx.x; i < 256 * 16; i += FH_NT) reinterpret_cast<uint4*>(hist)[i] = make_uint4(0, 0, 0, 0);
        if (threadIdx.x < FH_COLS) minlo[threadIdx.x] = 0xffffffffu;
        if (threadIdx.x == 0) any_shift = 0;
        // pass 0: range of every pixel.  Skipped (fixed split: base 0, the high byte first) for 8-bit input, whose keys fit the
        // bins as they are, and while the data proves wide: after a probed tile that needed the refinement the next 1, then 3,
        // then 7 tiles of this workgroup (neighbours along the row) take the fixed split without asking — a range pass that
        // finds "wide" again is a read of the columns for nothing (12-bit-wide pixels: 4.8 -> 3.55 ms; the fixed split alone:
        // 3.9), while an isolated wide tile (a bright bead in a noise-like image) costs its neighbour only.
        const bool probe = K::bits > 8 && skip == 0;
        if (skip > 0) --skip;
        unsigned mna = probe ? 0xffffffffu : 0u, mxa = probe ? 0u : (K::bits > 8 ? 0xffffu : 0xffu);
        unsigned mnb = mna, mxb = mxa;
        for (int z0 = wave; probe && z0 < Z; z0 += FH_NW * FH_U) {
            unsigned ka[FH_U], kb[FH_U];
#pragma unroll
            for (int u = 0; u < FH_U; ++u) load2(min(z0 + FH_NW * u, Z - 1), ka[u], kb[u]);  // clamped rows repeat a sample: harmless here
#pragma unroll
            for (int u = 0; u < FH_U; ++u) {
                mna = min(mna, ka[u]), mxa = max(mxa, ka[u]);
                mnb = min(mnb, kb[u]), mxb = max(mxb, kb[u]);
            }
        }
        if (probe) {  // (workgroup-uniform)
            wmn[wave][2 * lane] = mna, wmx[wave][2 * lane] = mxa;
            wmn[wave][2 * lane + 1] = mnb, wmx[wave][2 * lane + 1] = mxb;
        }
        __syncthreads();
        if (probe) {
#pragma unroll
            for (int w = 0; w < FH_NW; ++w) {
                mna = min(mna, wmn[w][2 * lane]), mxa = max(mxa, wmx[w][2 * lane]);
                mnb = min(mnb, wmn[w][2 * lane + 1]), mxb = max(mxb, wmx[w][2 * lane + 1]);
            }
        }
        // bits to drop so that the range fits 256 bins (0 for a range below 256)
        const int sa = max(0, 24 - (int)__builtin_clz((mxa - mna) | 1u)), sb = max(0, 24 - (int)__builtin_clz((mxb - mnb) | 1u));
        if ((sa | sb) != 0) any_shift = 1;  // benign race: every writer stores 1
        __syncthreads();
        const bool refine = any_shift != 0;
        if (probe) {
            streak = refine ? min(streak + 1, 3) : 0;
            skip = (1 << streak) - 1;
        }
        // pass 1: bins of (key - mn) >> s
        for (int z0 = wave; z0 < Z; z0 += FH_NW * FH_U) {
            unsigned ka[FH_U], kb[FH_U];
#pragma unroll
            for (int u = 0; u < FH_U; ++u) load2(min(z0 + FH_NW * u, Z - 1), ka[u], kb[u]);
#pragma unroll
            for (int u = 0; u < FH_U; ++u)
                if (z0 + FH_NW * u < Z) {
                    atomicAdd(&hist[((ka[u] - mna) >> sa) * 64 + lane], 1u);
                    atomicAdd(&hist[((kb[u] - mnb) >> sb) * 64 + lane], 0x10000u);
                }
        }
        __syncthreads();
        if (threadIdx.x < FH_COLS) {  // one thread per pixel: the bins of ranks k and k2
            const int c = threadIdx.x, l = c >> 1, sh = (c & 1) * 16;
            unsigned cum = 0, b1 = 0, b2 = 0, r1 = 0;
            bool f1 = false, f2 = false;
            for (int b0 = 0; b0 < 256; b0 += 8) {  // eight independent LDS reads in flight per step
                unsigned n[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) n[u] = (hist[(b0 + u) * 64 + l] >> sh) & 0xffffu;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (!f1 && cum + n[u] > (unsigned)k) f1 = true, b1 = b0 + u, r1 = (unsigned)k - cum;
                    if (!f2 && cum + n[u] > (unsigned)k2) f2 = true, b2 = b0 + u;
                    cum += n[u];
                }
                if (f2) break;  // both ranks placed (k2 >= k)
            }
            sb1[c] = (unsigned short)b1, sb2[c] = (unsigned short)b2, sr1[c] = (unsigned short)r1;
        }
        __syncthreads();
        const unsigned a1 = sb1[2 * lane], a2 = sb2[2 * lane], c1 = sb1[2 * lane + 1], c2 = sb2[2 * lane + 1];
        if (refine) {  // (workgroup-uniform)
            for (int i = threadIdx.x; i < 256 * 16; i += FH_NT) reinterpret_cast<uint4*>(hist)[i] = make_uint4(0, 0, 0, 0);
            __syncthreads();
            // pass 2: dropped low bits of the samples in bin b1; smallest low part of bin b2 when it is another bin
            const unsigned ma = (1u << sa) - 1u, mb = (1u << sb) - 1u;
            for (int z0 = wave; z0 < Z; z0 += FH_NW * FH_U) {
                unsigned ka[FH_U], kb[FH_U];
#pragma unroll
                for (int u = 0; u < FH_U; ++u) load2(min(z0 + FH_NW * u, Z - 1), ka[u], kb[u]);
#pragma unroll
                for (int u = 0; u < FH_U; ++u)
                    if (z0 + FH_NW * u < Z) {
                        const unsigned da = ka[u] - mna, db = kb[u] - mnb;
                        if ((da >> sa) == a1) atomicAdd(&hist[(da & ma) * 64 + lane], 1u);
                        else if ((da >> sa) == a2) atomicMin(&minlo[2 * lane], da & ma);
                        if ((db >> sb) == c1) atomicAdd(&hist[(db & mb) * 64 + lane], 0x10000u);
                        else if ((db >> sb) == c2) atomicMin(&minlo[2 * lane + 1], db & mb);
                    }
            }
            __syncthreads();
        }
        // this lane's pixels' base and shift travel to the pixel-owning threads through LDS (wmn / wmx are free again)
        if (wave == 0) {
            wmn[0][2 * lane] = mna, wmn[0][2 * lane + 1] = mnb;
            wmx[0][2 * lane] = (unsigned)sa, wmx[0][2 * lane + 1] = (unsigned)sb;
        }
        __syncthreads();
        if (threadIdx.x < FH_COLS && x0 + (int)threadIdx.x < X) {
            const int c = threadIdx.x, l = c >> 1, sh = (c & 1) * 16;
            const unsigned b1 = sb1[c], b2 = sb2[c], r1 = sr1[c], mn = wmn[0][c], s = wmx[0][c];
            unsigned lo1 = 0, lo2 = 0;
            if (refine) {
                unsigned cum = 0;
                bool f1 = false, f2 = false;
                for (int b0 = 0; b0 < 256; b0 += 8) {
                    unsigned n[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) n[u] = (hist[(b0 + u) * 64 + l] >> sh) & 0xffffu;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (!f1 && cum + n[u] > r1) f1 = true, lo1 = b0 + u;
                        if (!f2 && cum + n[u] > r1 + 1) f2 = true, lo2 = b0 + u;  // the next rank, if in the same bin
                        cum += n[u];
                    }
                }
            }
            const unsigned v1 = mn + (b1 << s) + lo1;
            const unsigned v2 = (Z & 1) ? v1 : (b2 == b1 ? mn + (b1 << s) + lo2 : mn + (b2 << s) + (refine ? minlo[c] : 0u));
            pattern[(size_t)y * X + x0 + c] = (Z & 1) ? K::from(v1) : (K::from(v1) + K::from(v2)) * 0.5;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void ff_sum_partial_kernel(const double* __restrict__ v, long long n,
                                                             double* __restrict__ part) {
    double a = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) a += v[i];
    __shared__ double red[256];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void ff_sum_final_kernel(const double* __restrict__ part, int n, double inv_count,
                                                           double* __restrict__ out) {
    __shared__ double red[256];
    double a = 0;
    for (int i = threadIdx.x; i < n; i += 256) a += part[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0] * inv_count;
}

// one thread per pixel, walking Z: the pattern value is read once, every access is coalesced along x
template <typename TIN>
__global__ __launch_bounds__(256) void flat_apply_kernel(const TIN* __restrict__ in, const double* __restrict__ pattern,
                                                         const double* __restrict__ mean, long long plane, int Z, int zchunk,
                                                         float* __restrict__ out) {
    const double m = mean[0];
    const float mf = (float)m;
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= plane) return;
    const double pat = pattern[p];
    const float patf = (float)pat;
    const int z0 = blockIdx.y * zchunk, z1 = min(Z, z0 + zchunk);
    const TIN* src = in + (size_t)z0 * plane + p;
    float* dst = out + (size_t)z0 * plane + p;
#pragma unroll 4
    for (int z = z0; z < z1; ++z, src += plane, dst += plane) {
        if (sizeof(TIN) == 4)
            *dst = ((float)*src / patf) * mf;          // float32 / float32 * float32, as numpy evaluates it
        else
            *dst = (float)((double)*src / pat * m);    // integer / float64 * float64, cast by the CZYX adapter
    }
}

template <typename TIN, int FF_NT>
static int run_median(bh_ctx* ctx, const TIN* in, int64_t Z, int64_t Y, int64_t X, double* pattern) {
    using key_t = typename FfKey<TIN>::key_t;
    const size_t budget = 144 * 1024;  // of the CU's 160 KiB
    int cols = 64;
    while (cols > 4 && (size_t)Z * cols * sizeof(key_t) > budget) cols >>= 1;
    if ((size_t)Z * cols * sizeof(key_t) > budget) {
        set_error("flat-field median: Z = %lld does not fit the LDS staging (max %zu samples per pixel)", (long long)Z,
                  budget / (4 * sizeof(key_t)));
        return BH_ERR_UNSUPPORTED;
    }
    const size_t lds = (((size_t)Z * cols * sizeof(key_t) + 15) & ~(size_t)15) + 2 * FF_NT * sizeof(unsigned);
    const int tiles_x = (int)ceil_div(X, cols);
    const int64_t tiles = (int64_t)tiles_x * Y;
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / (lds + 256)));
    const int grid = (int)std::min<int64_t>(tiles, (int64_t)ctx->num_cus * per_cu);
    auto kern = median_z_kernel<TIN, FF_NT>;
    BH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(FF_NT), lds, ctx->stream, in, (int)Z, (int)Y, (int)X, cols, tiles_x, pattern);
    BH_CHECK_HIP(hipGetLastError());
    return BH_OK;
}

template <int NT>
static int dispatch_median_nt(bh_ctx* ctx, const void* in, int dtype, int64_t Z, int64_t Y, int64_t X, double* pattern) {
    switch (dtype) {
        case BH_DT_U8: return run_median<uint8_t, NT>(ctx, (const uint8_t*)in, Z, Y, X, pattern);
        case BH_DT_U16: return run_median<uint16_t, NT>(ctx, (const uint16_t*)in, Z, Y, X, pattern);
        case BH_DT_I16: return run_median<int16_t, NT>(ctx, (const int16_t*)in, Z, Y, X, pattern);
        case BH_DT_F32: return run_median<float, NT>(ctx, (const float*)in, Z, Y, X, pattern);
        default: BH_REQUIRE(false, "unsupported input dtype code %d", dtype);
    }
    return BH_OK;
}

template <typename TIN>
static int run_median_hist(bh_ctx* ctx, const TIN* in, int64_t Z, int64_t Y, int64_t X, double* pattern) {
    const int tiles_x = (int)ceil_div(X, FH_COLS);
    const int grid = (int)std::min<int64_t>((int64_t)tiles_x * Y, (int64_t)ctx->num_cus * 2);
    hipLaunchKernelGGL(median_hist_kernel<TIN>, dim3(grid), dim3(FH_NT), 0, ctx->stream, in, (int)Z, (int)Y, (int)X, tiles_x,
                       pattern);
    BH_CHECK_HIP(hipGetLastError());
    return BH_OK;
}

static int dispatch_median(bh_ctx* ctx, const void* in, int dtype, int64_t Z, int64_t Y, int64_t X, double* pattern) {
    const bool bitsearch = getenv("BH_FF_BITSEARCH") != nullptr;  // force the bit-search kernel (tests, comparison)
    if (!bitsearch && dtype != BH_DT_F32 && Z <= 65535 && X >= 2) {
        switch (dtype) {
            case BH_DT_U8: return run_median_hist(ctx, (const uint8_t*)in, Z, Y, X, pattern);
            case BH_DT_U16: return run_median_hist(ctx, (const uint16_t*)in, Z, Y, X, pattern);
            case BH_DT_I16: return run_median_hist(ctx, (const int16_t*)in, Z, Y, X, pattern);
            default: BH_REQUIRE(false, "unsupported input dtype code %d", dtype);
        }
    }
    // measured (tools/time_ops.py flatfield): 32-bit keys want 16 waves per workgroup, 16-bit keys 8
    static const int forced = getenv("BH_FF_NT") ? atoi(getenv("BH_FF_NT")) : 0;
    const int nt = forced ? forced : (dtype == BH_DT_F32 ? 1024 : 512);
    if (nt == 1024) return dispatch_median_nt<1024>(ctx, in, dtype, Z, Y, X, pattern);
    if (nt == 512) return dispatch_median_nt<512>(ctx, in, dtype, Z, Y, X, pattern);
    return dispatch_median_nt<256>(ctx, in, dtype, Z, Y, X, pattern);
}

}  // namespace bh

using namespace bh;

extern "C" int bh_median_z(bh_ctx* ctx, const void* in, int in_dtype, int64_t Z, int64_t Y, int64_t X, double* pattern) {
    BH_REQUIRE(ctx && in && pattern, "null argument");
    BH_REQUIRE(Z > 0 && Y > 0 && X > 0 && Z < (1ll << 24) && Y < (1ll << 31) && X < (1ll << 31), "bad shape");
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    return dispatch_median(ctx, in, in_dtype, Z, Y, X, pattern);
}

extern "C" int bh_flat_field(bh_ctx* ctx, const void* in, int in_dtype, int64_t Z, int64_t Y, int64_t X, float* out,
                             double* pattern, double* mean_out) {
    BH_REQUIRE(ctx && in && out, "null argument");
    BH_REQUIRE(Z > 0 && Y > 0 && X > 0 && Z < (1ll << 24) && Y < (1ll << 31) && X < (1ll << 31), "bad shape");
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    const int64_t plane = Y * X;
    double* pat = pattern;
    if (!pat) BH_TRY(get_scratch(ctx, "ff_pattern", sizeof(double) * (size_t)plane, (void**)&pat));
    double* red;
    const int nblk = (int)std::min<int64_t>(ceil_div(plane, 256), 1024);
    BH_TRY(get_scratch(ctx, "ff_reduce", sizeof(double) * (size_t)(nblk + 2), (void**)&red));
    ScopedTimer timer(ctx, T_FLATFIELD);
    BH_TRY(dispatch_median(ctx, in, in_dtype, Z, Y, X, pat));
    hipLaunchKernelGGL(ff_sum_partial_kernel, dim3(nblk), dim3(256), 0, ctx->stream, pat, (long long)plane, red);
    hipLaunchKernelGGL(ff_sum_final_kernel, dim3(1), dim3(256), 0, ctx->stream, red, nblk, 1.0 / (double)plane, red + nblk);
    const int zchunk = 32;
    const dim3 grid((unsigned)ceil_div(plane, 256), (unsigned)ceil_div(Z, zchunk));
#define BH_FF_APPLY(T) \
    hipLaunchKernelGGL(flat_apply_kernel<T>, grid, dim3(256), 0, ctx->stream, (const T*)in, pat, red + nblk, (long long)plane, (int)Z, zchunk, out)
    switch (in_dtype) {
        case BH_DT_U8: BH_FF_APPLY(uint8_t); break;
        case BH_DT_U16: BH_FF_APPLY(uint16_t); break;
        case BH_DT_I16: BH_FF_APPLY(int16_t); break;
        case BH_DT_F32: BH_FF_APPLY(float); break;
        default: BH_REQUIRE(false, "unsupported input dtype code %d", in_dtype);
    }
#undef BH_FF_APPLY
    BH_CHECK_HIP(hipGetLastError());
    if (mean_out) {
        BH_CHECK_HIP(hipMemcpyAsync(mean_out, red + nblk, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        BH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    }
    return BH_OK;
}
