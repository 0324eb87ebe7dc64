// Overhang fill for deskewed volumes on gfx950 — restates biahub/deskew.py:339-368.
//
// Reference: mask = (data == 0); 3x max_pool3d(k=3,s=1,p=1) (26-connected dilation, i.e. a
// Chebyshev ball of radius `iterations`); fill = mean(data[~mask]) or a constant;
// out = where(mask, fill, data).  Three full float passes + a float mask there.
//
// Here the mask is one BIT per voxel (rows padded to 64 bits), dilation is separable
// (x by shifts inside 32-bit words, then y, then z by OR-ing rows/planes), the mean uses
// sum(valid) = sum(all) - sum(dilated shell) so the float volume is read once, and the fill
// pass writes only masked voxels.
#include "common.hpp"

namespace bh {

// `enable` (may be null): a device flag; 0 = this launch has nothing to do.  The one-pass deskew (deskew.hip) queues the mask
// pipeline behind itself unconditionally and raises the flag only when it met a zero that geometry does not explain.
#define BH_FILL_ENABLED(enable) \
    if ((enable) != nullptr && *(enable) == 0) return

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// pass 0: zero mask bits + per-block partial sum.  One wave handles 64 consecutive x of one row
// per step; W32 = words per row (even).
__global__ __launch_bounds__(256) void mask0_kernel(const float* __restrict__ data, uint32_t* __restrict__ m0,
                                                    double* __restrict__ partial, int64_t rows, int X, int W32) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int W64 = W32 / 2;
    const int64_t nseg = rows * W64;
    double s = 0.0;
    for (int64_t seg = (int64_t)blockIdx.x * 4 + wave; seg < nseg; seg += (int64_t)gridDim.x * 4) {
        const int64_t row = seg / W64;
        const int x = (int)(seg % W64) * 64 + lane;
        float v = 1.0f;
        bool ok = x < X;
        if (ok) v = data[row * X + x];
        const unsigned long long b = __ballot(ok && v == 0.0f);
        if (ok) s += (double)v;
        if (lane == 0) {
            m0[row * W32 + (seg % W64) * 2] = (uint32_t)b;
            m0[row * W32 + (seg % W64) * 2 + 1] = (uint32_t)(b >> 32);
        }
    }
    __shared__ double sh[4];
    s = wave_sum(s);
    if (lane == 0) sh[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// dilate along x inside each row of W32 words by radius r (< 32)
__global__ void dilate_x_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int64_t nwords, int W32,
                                int r, const int* __restrict__ enable) {
    BH_FILL_ENABLED(enable);
    const unsigned int nw = (unsigned int)nwords;  // host: nwords < 2^32 (64-bit divisions are software loops on this part)
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += gridDim.x * blockDim.x) {
        const int wi = (int)(i % (unsigned int)W32);
        const uint32_t w = src[i];
        const uint32_t l = wi > 0 ? src[i - 1] : 0u;
        const uint32_t h = wi + 1 < W32 ? src[i + 1] : 0u;
        uint32_t o = w;
        for (int s = 1; s <= r; ++s) o |= (w << s) | (l >> (32 - s)) | (w >> s) | (h << (32 - s));
        dst[i] = o;
    }
}

// dilate along an outer axis: element (o, m, i) with stride `stride` words between neighbours.  A thread owns two words
// (rows hold an even number of them, so a pair never straddles a row and `stride` is even): 8-byte accesses, and one 32-bit
// division per pair instead of two 64-bit ones per word (software loops on this part) — 0.79 -> 0.4 ms per pass on the
// deskewed config-2 mask (134 M words).
__global__ void dilate_outer_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int64_t nwords,
                                    int64_t stride, int len, int r, const int* __restrict__ enable) {
    BH_FILL_ENABLED(enable);
    const uint2* __restrict__ s2 = reinterpret_cast<const uint2*>(src);
    uint2* __restrict__ d2 = reinterpret_cast<uint2*>(dst);
    const unsigned int np = (unsigned int)(nwords >> 1), st2 = (unsigned int)(stride >> 1);  // host: nwords < 2^32
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < np; i += gridDim.x * blockDim.x) {
        const int pos = (int)((i / st2) % (unsigned int)len);
        uint2 o = s2[i];
        for (int d = 1; d <= r; ++d) {
            if (pos - d >= 0) {
                const uint2 a = s2[i - d * st2];
                o.x |= a.x;
                o.y |= a.y;
            }
            if (pos + d < len) {
                const uint2 a = s2[i + d * st2];
                o.x |= a.x;
                o.y |= a.y;
            }
        }
        d2[i] = o;
    }
}

// `r` iterations of the 6-connected (cross) structuring element = the L1 ball of radius r: a voxel is set when a set voxel
// lies within |dz| + |dy| + |dx| <= r (SciPy binary_dilation's default structure and border_value 0, which the legacy
// _fill_overhang_with_mean uses, biahub/deskew.py:277-336).  Per (dz, dy) the row (z + dz, y + dy) is dilated along x by
// the remaining radius inside its words.  Not separable, ~2 r^2 row reads per word: the legacy path is not a hot one.
__global__ void dilate_cross_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int64_t nwords, int W32,
                                    int Y, int Z, int r) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int wi = (int)(i % W32);
        const int64_t row = i / W32;
        const int y = (int)(row % Y), z = (int)(row / Y);
        uint32_t o = 0u;
        for (int dz = -r; dz <= r; ++dz) {
            if (z + dz < 0 || z + dz >= Z) continue;
            const int ry = r - (dz < 0 ? -dz : dz);
            for (int dy = -ry; dy <= ry; ++dy) {
                if (y + dy < 0 || y + dy >= Y) continue;
                const int rx = ry - (dy < 0 ? -dy : dy);
                const uint32_t* p = src + ((int64_t)(z + dz) * Y + (y + dy)) * W32 + wi;
                const uint32_t w = p[0];
                o |= w;
                if (rx > 0) {
                    const uint32_t l = wi > 0 ? p[-1] : 0u;
                    const uint32_t h = wi + 1 < W32 ? p[1] : 0u;
                    for (int s = 1; s <= rx; ++s) o |= (w << s) | (l >> (32 - s)) | (w >> s) | (h << (32 - s));
                }
            }
        }
        dst[i] = o;
    }
}

// shell sum: voxels in the dilated mask that are not exact zeros; also counts masked voxels.  A thread owns a pair of mask
// words (rows hold an even number of them): 8-byte loads and one 32-bit division per pair — the first version divided a
// 64-bit word index per word, which is a software loop on this part and made the pass 1.2 ms for 1 GB of masks.
__global__ __launch_bounds__(256) void shell_kernel(const float* __restrict__ data, const uint32_t* __restrict__ m0,
                                                    const uint32_t* __restrict__ md, double* __restrict__ psum,
                                                    unsigned long long* __restrict__ pcnt, int64_t nwords, int X,
                                                    int W32, const int* __restrict__ enable) {
    BH_FILL_ENABLED(enable);
    double s = 0.0;
    unsigned long long c = 0;
    const uint2* __restrict__ m02 = reinterpret_cast<const uint2*>(m0);
    const uint2* __restrict__ md2 = reinterpret_cast<const uint2*>(md);
    const unsigned int np = (unsigned int)(nwords >> 1), W2 = (unsigned int)(W32 >> 1);  // host: nwords < 2^32
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < np; i += gridDim.x * blockDim.x) {
        const unsigned int row = i / W2;
        const int x0 = (int)(i - row * W2) * 64;
        if (x0 >= X) continue;
        const uint2 dv = md2[i], zv = m02[i];
        unsigned long long d = ((unsigned long long)dv.y << 32) | dv.x;
        const unsigned long long z = ((unsigned long long)zv.y << 32) | zv.x;
        if (X - x0 < 64) d &= (1ull << (X - x0)) - 1ull;
        c += __popcll(d);
        unsigned long long shell = d & ~z;
        const float* rowp = data + (size_t)row * X + x0;
        while (shell) {
            const int b = __ffsll((long long)shell) - 1;
            shell &= shell - 1;
            s += (double)rowp[b];
        }
    }
    __shared__ double shs[4];
    __shared__ unsigned long long shc[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    s = wave_sum(s);
    c = wave_sum_u64(c);
    if (lane == 0) {
        shs[wave] = s;
        shc[wave] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        psum[blockIdx.x] = shs[0] + shs[1] + shs[2] + shs[3];
        pcnt[blockIdx.x] = shc[0] + shc[1] + shc[2] + shc[3];
    }
}

// deterministic final reduction (single block) + fill value
constexpr int FIN_NT = 1024;  // the fused deskew hands over one partial per workgroup (262 144 at config 2): 256 threads took 0.41 ms
__global__ __launch_bounds__(FIN_NT) void finalize_kernel(const double* __restrict__ p_all, int n_all,
                                                       const double* __restrict__ p_shell,
                                                       const unsigned long long* __restrict__ p_cnt, int n_shell,
                                                       FillStats* st, unsigned long long total, int fill_mode,
                                                       float fill_value, const int* __restrict__ enable) {
    BH_FILL_ENABLED(enable);
    __shared__ double sa[FIN_NT], ss[FIN_NT];
    __shared__ unsigned long long sc[FIN_NT];
    double a = 0, s = 0;
    unsigned long long c = 0;
    {   // eight independent loads in flight per thread (fixed order: the result is reproducible)
        double a8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int i = threadIdx.x;
        for (; i + 7 * FIN_NT < n_all; i += 8 * FIN_NT) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a8[u] += p_all[i + u * FIN_NT];
        }
        for (; i < n_all; i += FIN_NT) a8[0] += p_all[i];
        a = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    }
    for (int i = threadIdx.x; i < n_shell; i += FIN_NT) {
        s += p_shell[i];
        c += p_cnt[i];
    }
    sa[threadIdx.x] = a;
    ss[threadIdx.x] = s;
    sc[threadIdx.x] = c;
    __syncthreads();
    for (int o = FIN_NT / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            sa[threadIdx.x] += sa[threadIdx.x + o];
            ss[threadIdx.x] += ss[threadIdx.x + o];
            sc[threadIdx.x] += sc[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        st->sum_all = sa[0];
        st->sum_shell = ss[0];
        st->n_masked = sc[0];
        if (fill_mode == BH_FILL_MEAN) {
            const double nvalid = (double)(total - sc[0]);
            st->fill = (float)((sa[0] - ss[0]) / nvalid);  // 0/0 -> NaN like torch's empty mean
        } else {
            st->fill = fill_value;
        }
    }
}

// Fill pass: a wavefront owns 2048 voxels of one row at a time (64 mask words, one per lane) and writes the fill value in
// 16-byte stores: lane l of step s takes the four voxels x = a0 + 4 (64 s + l) .. + 3, a0 in 0..3 chosen per row so that
// the group is 16-B aligned in memory (rows of an (.., X) volume start at any multiple of 4 B), and fetches its four mask bits
// from the word(s) that hold them with one or two cross-lane reads.  The first version stored 4 B per lane and segment —
// ~20 instructions per 64 voxels, which is what bounded it (2.6 ms for 9.9 GB at config 2, a plain fill_ writes 6.9 TB/s).
// Vector loads and stores retire through ONE in-order counter (vmcnt), so the mask words of the NEXT unit are requested
// before this unit's stores are issued: waiting for them then never waits for a store.
__global__ __launch_bounds__(256) void apply_fill_kernel(float* __restrict__ data, const uint32_t* __restrict__ md,
                                                         const FillStats* __restrict__ st, int64_t rows, int X,
                                                         int W32, const int* __restrict__ enable) {
    BH_FILL_ENABLED(enable);
    const float fill = st->fill;
    const float4 fill4 = make_float4(fill, fill, fill, fill);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nchunk = (X + 2047) / 2048;
    const long nunits = rows * nchunk, step = (long)gridDim.x * 4;
    auto load_words = [&](long u, uint32_t& m, uint32_t& mnext) {
        const long row = u / nchunk;
        const int c = (int)(u - row * nchunk);
        const uint32_t* w = md + row * W32 + c * 64;
        m = (c * 64 + lane < W32) ? w[lane] : 0u;
        mnext = (c * 64 + 64 < W32) ? w[64] : 0u;
    };
#ifndef BH_FILL_CONTIG
#define BH_FILL_CONTIG 1  // a wavefront walks a CONTIGUOUS range of units (2.1 MB of voxels) instead of every (4 x grid)-th one: 2.69 -> 2.25 ms
                          // at config 2 (same box, tools/time_deskew.py: fill passes 4.85 -> 4.4 ms); 0 is the A/B switch
#endif
    const long per = (nunits + step - 1) / step;
    long u = BH_FILL_CONTIG ? ((long)blockIdx.x * 4 + wave) * per : (long)blockIdx.x * 4 + wave;
    const long ustep = BH_FILL_CONTIG ? 1 : step;
    const long uend = BH_FILL_CONTIG ? min(nunits, u + per) : nunits;
    uint32_t m = 0u, mnext = 0u;
    if (u < uend) load_words(u, m, mnext);
    for (; u < uend; u += ustep) {
        const uint32_t cur = m, curn = mnext;
        if (u + ustep < uend) load_words(u + ustep, m, mnext);
        if (__ballot(cur != 0u) == 0ull && curn == 0u) continue;
        const long row = u / nchunk;
        const int c = (int)(u - row * nchunk);
        const int x0 = c * 2048;
        float* rowp = data + row * X;
        const int a0 = (int)((4 - ((row * X) & 3)) & 3);
        const uint32_t w0 = (uint32_t)__shfl((int)cur, 0, 64);
        if (c == 0 && lane < a0 && lane < X && ((w0 >> lane) & 1u)) rowp[lane] = fill;  // the row's unaligned head
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int xr = a0 + 256 * s + 4 * lane;  // relative to x0: 0 .. 2050
            const int idx = xr >> 5, sh = xr & 31;
            const uint32_t lo = (uint32_t)__shfl((int)cur, idx & 63, 64);  // every lane takes part in every cross-lane read
            uint32_t bits = (idx < 64 ? lo : curn) >> sh;
            if (a0 != 0) {  // (wave-uniform) a group may straddle two words
                const uint32_t hs = (uint32_t)__shfl((int)cur, (idx + 1) & 63, 64);
                const uint32_t hi = idx + 1 < 64 ? hs : curn;
                if (sh > 28) bits |= hi << (32 - sh);
            }
            bits &= 15u;
            const int x = x0 + xr;
            if (bits == 15u && x + 3 < X) {
                *reinterpret_cast<float4*>(rowp + x) = fill4;
            } else if (bits != 0u) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (((bits >> e) & 1u) && x + e < X) rowp[x + e] = fill;
            }
        }
    }
}

// mask buffer shared with the fused deskew prologue
int fill_mask_buffers(bh_ctx* ctx, int64_t rows, int64_t X, uint32_t** m0, int* W32) {
    *W32 = (int)(ceil_div(X, 64) * 2);
    return get_scratch(ctx, "fill_m0", (size_t)rows * *W32 * 4, (void**)m0);
}

// fused_partials > 0: the zero mask ("fill_m0") and that many block sums ("fill_pall") were already written by
// the deskew kernel, which also skipped storing exact zeros; otherwise both come from mask0_kernel here.
int fill_overhang_impl(bh_ctx* ctx, float* data, int64_t Z, int64_t Y, int64_t X, int fill_mode, float fill_value,
                       int iterations, float* mean_out, int fused_partials, int connectivity, const int* enable) {
    BH_REQUIRE(iterations >= 0 && iterations < 32, "dilation_iterations must be in [0,31], got %d", iterations);
    BH_REQUIRE(X < (1ll << 31) && Y < (1ll << 31) && Z < (1ll << 31), "volume too large");
    ScopedTimer timer(ctx, T_FILL);
    const int64_t rows = Z * Y;
    const int W32 = (int)(ceil_div(X, 64) * 2);
    const int64_t nwords = rows * W32;
    BH_REQUIRE(nwords < (1ll << 32), "volume too large for the 32-bit mask index (%lld mask words)", (long long)nwords);
    const int nblk = ctx->num_cus * 8;
    uint32_t *mA, *mB, *m0;
    double *p_all, *p_shell;
    unsigned long long* p_cnt;
    FillStats* st;
    BH_TRY(get_scratch(ctx, "fill_m0", nwords * 4, (void**)&m0));
    BH_TRY(get_scratch(ctx, "fill_mA", nwords * 4, (void**)&mA));
    BH_TRY(get_scratch(ctx, "fill_mB", nwords * 4, (void**)&mB));
    const int n_all = fused_partials > 0 ? fused_partials : nblk;
    BH_TRY(get_scratch(ctx, "fill_pall", (size_t)n_all * sizeof(double), (void**)&p_all));
    BH_TRY(get_scratch(ctx, "fill_pshell", nblk * sizeof(double), (void**)&p_shell));
    BH_TRY(get_scratch(ctx, "fill_pcnt", nblk * sizeof(unsigned long long), (void**)&p_cnt));
    BH_TRY(get_scratch(ctx, "fill_stats", sizeof(FillStats), (void**)&st));
    hipStream_t s = ctx->stream;
    if (fused_partials <= 0)
        hipLaunchKernelGGL(mask0_kernel, dim3(nblk), dim3(256), 0, s, data, m0, p_all, rows, (int)X, W32);
    const int tb = 256;
    const int gb = (int)std::min<int64_t>(ceil_div(nwords, tb), (int64_t)ctx->num_cus * 16);
    const uint32_t* md = m0;
    if (iterations > 0 && connectivity == 6) {
        BH_REQUIRE(enable == nullptr, "the conditional mask pipeline dilates with the 26-connected element only");
        hipLaunchKernelGGL(dilate_cross_kernel, dim3(gb), dim3(tb), 0, s, m0, mA, nwords, W32, (int)Y, (int)Z, iterations);
        md = mA;
    } else if (iterations > 0) {
        hipLaunchKernelGGL(dilate_x_kernel, dim3(gb), dim3(tb), 0, s, m0, mA, nwords, W32, iterations, enable);
        hipLaunchKernelGGL(dilate_outer_kernel, dim3(gb), dim3(tb), 0, s, mA, mB, nwords, (int64_t)W32, (int)Y,
                           iterations, enable);
        hipLaunchKernelGGL(dilate_outer_kernel, dim3(gb), dim3(tb), 0, s, mB, mA, nwords, (int64_t)W32 * Y, (int)Z,
                           iterations, enable);
        md = mA;
    }
    hipLaunchKernelGGL(shell_kernel, dim3(nblk), dim3(256), 0, s, data, m0, md, p_shell, p_cnt, nwords, (int)X, W32, enable);
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(FIN_NT), 0, s, p_all, n_all, p_shell, p_cnt, nblk, st,
                       (unsigned long long)(rows * X), fill_mode, fill_value, enable);
    hipLaunchKernelGGL(apply_fill_kernel, dim3(nblk), dim3(256), 0, s, data, md, st, rows, (int)X, W32, enable);
    BH_CHECK_HIP(hipGetLastError());
    if (mean_out) {
        FillStats h;
        BH_CHECK_HIP(hipMemcpyAsync(&h, st, sizeof(h), hipMemcpyDeviceToHost, s));
        BH_CHECK_HIP(hipStreamSynchronize(s));
        *mean_out = h.fill;
    }
    return BH_OK;
}

}  // namespace bh

extern "C" int bh_overhang_fill(bh_ctx* ctx, float* data, int64_t Z, int64_t Y, int64_t X, int fill_mode,
                                float fill_value, int dilation_iterations, float* mean_out) {
    BH_REQUIRE(ctx != nullptr && data != nullptr, "NULL argument");
    BH_REQUIRE(Z > 0 && Y > 0 && X > 0, "invalid shape");
    BH_REQUIRE(fill_mode == BH_FILL_CONSTANT || fill_mode == BH_FILL_MEAN, "fill_mode must be CONSTANT or MEAN");
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    return bh::fill_overhang_impl(ctx, data, Z, Y, X, fill_mode, fill_value, dilation_iterations, mean_out, 0, 26, nullptr);
}

extern "C" int bh_overhang_fill_connectivity(bh_ctx* ctx, float* data, int64_t Z, int64_t Y, int64_t X, int fill_mode,
                                             float fill_value, int dilation_iterations, int connectivity, float* mean_out) {
    BH_REQUIRE(ctx != nullptr && data != nullptr, "NULL argument");
    BH_REQUIRE(Z > 0 && Y > 0 && X > 0, "invalid shape");
    BH_REQUIRE(fill_mode == BH_FILL_CONSTANT || fill_mode == BH_FILL_MEAN, "fill_mode must be CONSTANT or MEAN");
    BH_REQUIRE(connectivity == 6 || connectivity == 26, "connectivity must be 6 or 26, got %d", connectivity);
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    return bh::fill_overhang_impl(ctx, data, Z, Y, X, fill_mode, fill_value, dilation_iterations, mean_out, 0, connectivity, nullptr);
}
