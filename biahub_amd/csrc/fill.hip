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

struct FillStats {       // device-resident
    double sum_all;      // sum of every voxel (zeros contribute nothing)
    double sum_shell;    // sum over dilated & ~zero
    unsigned long long n_masked;  // voxels in the dilated mask
    float fill;          // value written
    float pad;
};

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
                                int r) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int wi = (int)(i % W32);
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
                                    int64_t stride, int len, int r) {
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

// shell sum: voxels in the dilated mask that are not exact zeros; also counts masked voxels
__global__ __launch_bounds__(256) void shell_kernel(const float* __restrict__ data, const uint32_t* __restrict__ m0,
                                                    const uint32_t* __restrict__ md, double* __restrict__ psum,
                                                    unsigned long long* __restrict__ pcnt, int64_t nwords, int X,
                                                    int W32) {
    double s = 0.0;
    unsigned long long c = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / W32;
        const int wi = (int)(i % W32);
        const int x0 = wi * 32;
        if (x0 >= X) continue;
        uint32_t valid = (X - x0 >= 32) ? 0xFFFFFFFFu : ((1u << (X - x0)) - 1u);
        const uint32_t d = md[i] & valid;
        c += __popc(d);
        uint32_t shell = d & ~m0[i];
        while (shell) {
            const int b = __ffs(shell) - 1;
            shell &= shell - 1;
            s += (double)data[row * X + x0 + b];
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
__global__ __launch_bounds__(256) void finalize_kernel(const double* __restrict__ p_all, int n_all,
                                                       const double* __restrict__ p_shell,
                                                       const unsigned long long* __restrict__ p_cnt, int n_shell,
                                                       FillStats* st, unsigned long long total, int fill_mode,
                                                       float fill_value) {
    __shared__ double sa[256], ss[256];
    __shared__ unsigned long long sc[256];
    double a = 0, s = 0;
    unsigned long long c = 0;
    for (int i = threadIdx.x; i < n_all; i += 256) a += p_all[i];
    for (int i = threadIdx.x; i < n_shell; i += 256) {
        s += p_shell[i];
        c += p_cnt[i];
    }
    sa[threadIdx.x] = a;
    ss[threadIdx.x] = s;
    sc[threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
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

// the 64-bit value of lane k (k wave-uniform) in scalar registers: v_readlane, no trip through the LDS crossbar
__device__ __forceinline__ unsigned long long bcast_lane(unsigned long long v, int k) {
    const unsigned int lo = __builtin_amdgcn_readlane((unsigned int)v, k);
    const unsigned int hi = __builtin_amdgcn_readlane((unsigned int)(v >> 32), k);
    return ((unsigned long long)hi << 32) | lo;
}

constexpr int FILL_K = 16;  // mask words per lane and region: a wavefront owns FILL_K * 64 segments (256 KiB of voxels) at a time

__global__ __launch_bounds__(256) void apply_fill_kernel(float* __restrict__ data, const uint32_t* __restrict__ md,
                                                         const FillStats* __restrict__ st, int64_t rows, int X,
                                                         int W32) {
    // A wavefront walks 64-voxel segments and stores the fill value where the segment's mask word says so (one 256-B store
    // per segment).  Vector loads and stores retire through ONE in-order counter (vmcnt): waiting for a mask load issued
    // after stores means waiting for every one of those stores to be acknowledged.  The first version loaded 64 mask words
    // per 16 KiB of voxels and so drained its stores every 16 KiB: 22 us per chunk per wavefront, 3 TB/s.  Here a wavefront
    // loads the mask words of 256 KiB of voxels first (FILL_K coalesced loads, 32 registers) and then only stores: one drain
    // per 256 KiB.
    const float fill = st->fill;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W64 = W32 / 2;
    const long nseg = rows * W64;
    const unsigned long long* __restrict__ md64 = reinterpret_cast<const unsigned long long*>(md);
    const long region = (long)FILL_K * 64;
    const long step = (long)gridDim.x * 4 * region;
    for (long s0 = ((long)blockIdx.x * 4 + wave) * region; s0 < nseg; s0 += step) {
        unsigned long long w[FILL_K];
#pragma unroll
        for (int j = 0; j < FILL_K; ++j) {
            const long mine = s0 + 64 * j + lane;
            w[j] = mine < nseg ? md64[mine] : 0ull;
        }
#pragma unroll
        for (int j = 0; j < FILL_K; ++j) {
            if (__ballot(w[j] != 0ull) == 0ull) continue;
            const long sj = s0 + 64 * j;
            long row = sj / W64;
            int wq = (int)(sj - row * W64);
            const int cnt = (int)min((long)64, nseg - sj);
            for (int k = 0; k < cnt; ++k) {
                const unsigned long long m = bcast_lane(w[j], k);
                if (m != 0ull) {
                    const int x = wq * 64 + lane;
                    if (x < X && ((m >> lane) & 1ull)) data[row * X + x] = fill;
                }
                if (++wq == W64) {
                    wq = 0;
                    ++row;
                }
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
                       int iterations, float* mean_out, int fused_partials, int connectivity) {
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
        hipLaunchKernelGGL(dilate_cross_kernel, dim3(gb), dim3(tb), 0, s, m0, mA, nwords, W32, (int)Y, (int)Z, iterations);
        md = mA;
    } else if (iterations > 0) {
        hipLaunchKernelGGL(dilate_x_kernel, dim3(gb), dim3(tb), 0, s, m0, mA, nwords, W32, iterations);
        hipLaunchKernelGGL(dilate_outer_kernel, dim3(gb), dim3(tb), 0, s, mA, mB, nwords, (int64_t)W32, (int)Y,
                           iterations);
        hipLaunchKernelGGL(dilate_outer_kernel, dim3(gb), dim3(tb), 0, s, mB, mA, nwords, (int64_t)W32 * Y, (int)Z,
                           iterations);
        md = mA;
    }
    hipLaunchKernelGGL(shell_kernel, dim3(nblk), dim3(256), 0, s, data, m0, md, p_shell, p_cnt, nwords, (int)X, W32);
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, s, p_all, n_all, p_shell, p_cnt, nblk, st,
                       (unsigned long long)(rows * X), fill_mode, fill_value);
    hipLaunchKernelGGL(apply_fill_kernel, dim3(nblk), dim3(256), 0, s, data, md, st, rows, (int)X, W32);
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
    return bh::fill_overhang_impl(ctx, data, Z, Y, X, fill_mode, fill_value, dilation_iterations, mean_out, 0, 26);
}

extern "C" int bh_overhang_fill_connectivity(bh_ctx* ctx, float* data, int64_t Z, int64_t Y, int64_t X, int fill_mode,
                                             float fill_value, int dilation_iterations, int connectivity, float* mean_out) {
    BH_REQUIRE(ctx != nullptr && data != nullptr, "NULL argument");
    BH_REQUIRE(Z > 0 && Y > 0 && X > 0, "invalid shape");
    BH_REQUIRE(fill_mode == BH_FILL_CONSTANT || fill_mode == BH_FILL_MEAN, "fill_mode must be CONSTANT or MEAN");
    BH_REQUIRE(connectivity == 6 || connectivity == 26, "connectivity must be 6 or 26, got %d", connectivity);
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    return bh::fill_overhang_impl(ctx, data, Z, Y, X, fill_mode, fill_value, dilation_iterations, mean_out, 0, connectivity);
}
