// Intensity-registration building blocks on gfx950.
//
// The reference's `estimate` (biahub/registration/ants.py:55-122) hands two volumes to ants.registration with
// type_of_transform="Similarity", shrink factors (6,3,1), smoothing sigmas (2,1,0) voxels and (2100,1200,50)
// iterations (:93-98); ANTs/ITK then run a multi-resolution gradient descent on the Mattes mutual-information metric
// (32 bins, regular sampling at rate 0.2).  ANTs is a third-party binary (antspyx 0.6.1, absent here), so what this
// file provides are the three data-parallel pieces of that loop, each one flat C-ABI call; the optimiser itself is a
// few dozen host-side 4x4 operations per iteration (biahub_amd/registration/ants.py).  Parity is unpinned: only the
// recovered matrix can be compared (tests do that against a known ground truth); the CPU restatement of exactly these
// three definitions lives in oracle/oracle_np.py.
//
//   bh_image_stats    min / max / sum / first moments   (intensity range of the Parzen bins, centre-of-mass init)
//   bh_smooth_shrink  separable Gaussian (edge-clamped, radius ceil(4 sigma)) fused with integer subsampling per axis
//   bh_mattes_mi      Mattes MI value and its derivative w.r.t. the 3x4 pull matrix (fixed index -> moving index)
//
// Mattes MI (Mattes et al. 2003, as in ITK's MattesMutualInformationImageToImageMetricv4): joint histogram with a box
// window on the fixed intensity and a cubic B-spline window on the moving intensity, two padding bins at each end.
//   dMI/dP_ij = (1/N) sum_s w_s * dM/dc_i(c_s) * xh_s[j],   w_s = -(1/binsize_m) sum_k B3'(b_k - t_s) log(p(f_s,b_k)/p_M(b_k))
// with c_s = P xh_s the sample's moving-image index and t_s its Parzen coordinate.  Two passes over the samples
// (histogram, then gradient); both are gathers out of L2, the histogram accumulates in 2^-20 fixed point (integer
// atomics: exact and order independent) and the gradient through per-workgroup partial sums reduced in a fixed order,
// so a metric evaluation is bit-reproducible.
#include "common.hpp"

#include <algorithm>
#include <cmath>

namespace bh {

// ------------------------------------------------------------------------------------------------ image statistics
constexpr int ST_NT = 256;

__global__ __launch_bounds__(ST_NT) void stats_partial_kernel(const float* __restrict__ in, int Z, int Y, int X,
                                                              double* __restrict__ part) {
    // one row (z, y) per wave iteration: lanes along x; row sums are weighted by z and y once per row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = ST_NT / 64;
    const long long rows = (long long)Z * Y;
    float mn = INFINITY, mx = -INFINITY;
    double s = 0, sz = 0, sy = 0, sx = 0;
    for (long long r = (long long)blockIdx.x * nw + wave; r < rows; r += (long long)gridDim.x * nw) {
        const int z = (int)(r / Y), y = (int)(r - (long long)z * Y);
        const float* row = in + r * X;
        double rs = 0, rx = 0;
        for (int x = lane; x < X; x += 64) {
            const float v = row[x];
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
            rs += (double)v;
            rx += (double)v * (double)x;
        }
        s += rs;
        sx += rx;
        sz += rs * (double)z;
        sy += rs * (double)y;
    }
    __shared__ double red[ST_NT / 64][6];
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_down(mn, o));
        mx = fmaxf(mx, __shfl_down(mx, o));
        s += __shfl_down(s, o);
        sz += __shfl_down(sz, o);
        sy += __shfl_down(sy, o);
        sx += __shfl_down(sx, o);
    }
    if (lane == 0) {
        red[wave][0] = mn, red[wave][1] = mx, red[wave][2] = s, red[wave][3] = sz, red[wave][4] = sy, red[wave][5] = sx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double o[6] = {red[0][0], red[0][1], red[0][2], red[0][3], red[0][4], red[0][5]};
        for (int w = 1; w < nw; ++w) {
            o[0] = fmin(o[0], red[w][0]);
            o[1] = fmax(o[1], red[w][1]);
            for (int k = 2; k < 6; ++k) o[k] += red[w][k];
        }
        for (int k = 0; k < 6; ++k) part[(size_t)blockIdx.x * 6 + k] = o[k];
    }
}

// fixed-order tree over the per-workgroup partials: column k of `part` (n rows, `cols` columns) -> out[k]
// mode 0: sum; columns listed in minmax (0 -> min, 1 -> max) when with_minmax
__global__ __launch_bounds__(256) void reduce_partials_kernel(const double* __restrict__ part, int n, int cols,
                                                              int with_minmax, double* __restrict__ out) {
    __shared__ double red[256];
    for (int k = 0; k < cols; ++k) {
        const int op = with_minmax ? (k == 0 ? 1 : (k == 1 ? 2 : 0)) : 0;
        double a = op == 1 ? INFINITY : (op == 2 ? -INFINITY : 0.0);
        for (int i = threadIdx.x; i < n; i += 256) {
            const double v = part[(size_t)i * cols + k];
            a = op == 1 ? fmin(a, v) : (op == 2 ? fmax(a, v) : a + v);
        }
        red[threadIdx.x] = a;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) {
                const double b = red[threadIdx.x + o];
                red[threadIdx.x] = op == 1 ? fmin(red[threadIdx.x], b) : (op == 2 ? fmax(red[threadIdx.x], b) : red[threadIdx.x] + b);
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) out[k] = red[0];
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ smooth + shrink
constexpr int GS_MAXR = 32;

struct GaussParams {
    int Z, Y, X;     // input dims
    int axis;        // 0 z, 1 y, 2 x
    int No, f, o;    // output length along axis, shrink factor, first input index
    int R;           // kernel radius
    float w[2 * GS_MAXR + 1];
};

// out[.., i, ..] = sum_k w[k] * in[.., clamp(f i + o + k - R), ..]; lanes along x (coalesced for the y and z passes; the x
// pass reads overlapping windows out of L1/L2)
__global__ __launch_bounds__(256) void gauss_shrink_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           GaussParams p) {
    const int Zo = p.axis == 0 ? p.No : p.Z, Yo = p.axis == 1 ? p.No : p.Y, Xo = p.axis == 2 ? p.No : p.X;
    const long long total = (long long)Zo * Yo * Xo;
    const int N = p.axis == 0 ? p.Z : (p.axis == 1 ? p.Y : p.X);
    const long long stride = p.axis == 0 ? (long long)p.Y * p.X : (p.axis == 1 ? p.X : 1);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % Xo);
        const long long t = i / Xo;
        const int y = (int)(t % Yo), z = (int)(t / Yo);
        const int ia = p.axis == 0 ? z : (p.axis == 1 ? y : x);
        const int c = p.f * ia + p.o;
        const long long base = ((long long)(p.axis == 0 ? 0 : z) * p.Y + (p.axis == 1 ? 0 : y)) * p.X + (p.axis == 2 ? 0 : x);
        float acc = 0.0f;
        for (int k = -p.R; k <= p.R; ++k) {
            const int j = max(0, min(c + k, N - 1));
            acc += p.w[k + p.R] * in[base + (long long)j * stride];
        }
        out[i] = acc;
    }
}

// ------------------------------------------------------------------------------------------------ Mattes MI
constexpr int MI_NT = 256, MI_CHUNK = 4096, MI_PAD = 2, MI_MAXBINS = 64;
constexpr float MI_FIX = 1048576.0f;  // 2^20

struct MiParams {
    double P[12];
    int Zf, Yf, Xf, Zm, Ym, Xm;
    long long nsamples, stride, offset;
    float fscale, fnmin, mscale, mnmin;  // Parzen coordinate = v * scale - nmin
    int bins;
};

__device__ __forceinline__ float bspline3(float u) {
    const float a = fabsf(u);
    if (a < 1.0f) return (4.0f - 6.0f * a * a + 3.0f * a * a * a) * (1.0f / 6.0f);
    if (a < 2.0f) {
        const float b = 2.0f - a;
        return b * b * b * (1.0f / 6.0f);
    }
    return 0.0f;
}
__device__ __forceinline__ float bspline3_deriv(float u) {
    const float a = fabsf(u);
    if (a < 1.0f) return -2.0f * u + 1.5f * u * a;
    if (a < 2.0f) {
        const float b = 2.0f - a;
        return (u < 0.0f ? 0.5f : -0.5f) * b * b;
    }
    return 0.0f;
}

// One sample: fixed voxel `s` -> bins, moving Parzen coordinate, moving-image gradient (index space) and its own
// (z, y, x).  Returns false when the mapped point is outside the interpolable range [0, N-1]^3.
__device__ __forceinline__ bool mi_sample(const float* __restrict__ F, const float* __restrict__ M, const MiParams& p,
                                          long long s, int& fi, int& mi, float& mterm, float g[3], double xh[3]) {
    const long long idx = p.offset + s * p.stride;
    int z, y, x;
    if ((long long)p.Zf * p.Yf * p.Xf < (1ll << 31)) {
        const unsigned u = (unsigned)idx, yx = (unsigned)(p.Yf * p.Xf);
        z = (int)(u / yx);
        const unsigned r = u - (unsigned)z * yx;
        y = (int)(r / (unsigned)p.Xf);
        x = (int)(r - (unsigned)y * (unsigned)p.Xf);
    } else {
        const long long yx = (long long)p.Yf * p.Xf;
        z = (int)(idx / yx);
        const long long r = idx - (long long)z * yx;
        y = (int)(r / p.Xf);
        x = (int)(r - (long long)y * p.Xf);
    }
    xh[0] = z, xh[1] = y, xh[2] = x;
    double c[3];
    const int dims[3] = {p.Zm, p.Ym, p.Xm};
    bool ok = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        c[a] = p.P[4 * a] * xh[0] + p.P[4 * a + 1] * xh[1] + p.P[4 * a + 2] * xh[2] + p.P[4 * a + 3];
        ok = ok && c[a] >= 0.0 && c[a] <= (double)(dims[a] - 1);
    }
    if (!ok) return false;
    int i0[3], i1[3];
    float fr[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        i0[a] = min((int)c[a], max(dims[a] - 2, 0));
        i1[a] = min(i0[a] + 1, dims[a] - 1);
        fr[a] = (float)(c[a] - (double)i0[a]);
    }
    const size_t sY = (size_t)p.Xm, sZ = (size_t)p.Ym * p.Xm;
    const float* b00 = M + (size_t)i0[0] * sZ + (size_t)i0[1] * sY;
    const float* b01 = M + (size_t)i0[0] * sZ + (size_t)i1[1] * sY;
    const float* b10 = M + (size_t)i1[0] * sZ + (size_t)i0[1] * sY;
    const float* b11 = M + (size_t)i1[0] * sZ + (size_t)i1[1] * sY;
    const float v000 = b00[i0[2]], v001 = b00[i1[2]], v010 = b01[i0[2]], v011 = b01[i1[2]];
    const float v100 = b10[i0[2]], v101 = b10[i1[2]], v110 = b11[i0[2]], v111 = b11[i1[2]];
    const float fz = fr[0], fy = fr[1], fx = fr[2];
    // x lerps, then y, then z; the gradient is the analytic derivative of this trilinear interpolant
    const float a00 = v000 + fx * (v001 - v000), a01 = v010 + fx * (v011 - v010);
    const float a10 = v100 + fx * (v101 - v100), a11 = v110 + fx * (v111 - v110);
    const float b0 = a00 + fy * (a01 - a00), b1 = a10 + fy * (a11 - a10);
    const float m = b0 + fz * (b1 - b0);
    g[0] = b1 - b0;
    g[1] = (a01 - a00) + fz * ((a11 - a10) - (a01 - a00));
    const float d00 = v001 - v000, d01 = v011 - v010, d10 = v101 - v100, d11 = v111 - v110;
    const float e0 = d00 + fy * (d01 - d00), e1 = d10 + fy * (d11 - d10);
    g[2] = e0 + fz * (e1 - e0);
    const float f = F[idx];
    const float fterm = f * p.fscale - p.fnmin;
    fi = max(MI_PAD, min((int)floorf(fterm), p.bins - MI_PAD - 1));
    mterm = m * p.mscale - p.mnmin;
    mi = max(MI_PAD, min((int)floorf(mterm), p.bins - MI_PAD - 1));
    return true;
}

__global__ __launch_bounds__(MI_NT) void mi_hist_kernel(const float* __restrict__ F, const float* __restrict__ M,
                                                        MiParams p, unsigned long long* __restrict__ hist,
                                                        unsigned long long* __restrict__ nvalid) {
    extern __shared__ unsigned lh[];  // bins * bins
    const int nb2 = p.bins * p.bins;
    const long long nchunks = (p.nsamples + MI_CHUNK - 1) / MI_CHUNK;
    for (long long ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        for (int i = threadIdx.x; i < nb2; i += MI_NT) lh[i] = 0;
        __syncthreads();
        unsigned cnt = 0;
        for (int j = 0; j < MI_CHUNK / MI_NT; ++j) {
            const long long s = ch * MI_CHUNK + (long long)j * MI_NT + threadIdx.x;
            if (s < p.nsamples) {
                int fi, mi;
                float mterm, g[3];
                double xh[3];
                if (mi_sample(F, M, p, s, fi, mi, mterm, g, xh)) {
                    ++cnt;
#pragma unroll
                    for (int k = -1; k <= 2; ++k) {
                        const int b = mi + k;
                        const unsigned q = (unsigned)(bspline3((float)b - mterm) * MI_FIX + 0.5f);
                        if (q) atomicAdd(&lh[fi * p.bins + b], q);
                    }
                }
            }
        }
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
        if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(nvalid, (unsigned long long)cnt);
        __syncthreads();
        for (int i = threadIdx.x; i < nb2; i += MI_NT)
            if (lh[i]) atomicAdd(&hist[i], (unsigned long long)lh[i]);
        __syncthreads();
    }
}

// one workgroup: joint / marginal pdfs, MI value, and the table log(p(f,m) / p_M(m)) the gradient pass reads
__global__ __launch_bounds__(1024) void mi_table_kernel(const unsigned long long* __restrict__ hist, int bins,
                                                        float* __restrict__ table, double* __restrict__ out) {
    __shared__ double pM[MI_MAXBINS], pF[MI_MAXBINS], red[1024];
    const int nb2 = bins * bins;
    double t = 0;
    for (int i = threadIdx.x; i < nb2; i += 1024) t += (double)hist[i];
    red[threadIdx.x] = t;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const double total = red[0];
    __syncthreads();
    if ((int)threadIdx.x < bins) {
        double a = 0, b = 0;
        for (int k = 0; k < bins; ++k) {
            a += (double)hist[k * bins + threadIdx.x];  // over fixed bins -> moving marginal
            b += (double)hist[threadIdx.x * bins + k];  // over moving bins -> fixed marginal
        }
        pM[threadIdx.x] = total > 0 ? a / total : 0;
        pF[threadIdx.x] = total > 0 ? b / total : 0;
    }
    __syncthreads();
    double mi = 0;
    for (int i = threadIdx.x; i < nb2; i += 1024) {
        const int f = i / bins, m = i - f * bins;
        const double pj = total > 0 ? (double)hist[i] / total : 0;
        float l = 0.0f;
        if (pj > 1e-16 && pM[m] > 1e-16) {
            l = (float)log(pj / pM[m]);
            if (pF[f] * pM[m] > 1e-16) mi += pj * log(pj / (pF[f] * pM[m]));
        }
        table[i] = l;
    }
    red[threadIdx.x] = mi;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = red[0];
        out[1] = total / (double)MI_FIX;
    }
}

__global__ __launch_bounds__(MI_NT) void mi_grad_kernel(const float* __restrict__ F, const float* __restrict__ M,
                                                        MiParams p, const float* __restrict__ table,
                                                        double* __restrict__ part) {
    extern __shared__ float lt[];  // bins * bins log-ratio table, then reused for the reduction
    const int nb2 = p.bins * p.bins;
    for (int i = threadIdx.x; i < nb2; i += MI_NT) lt[i] = table[i];
    __syncthreads();
    double acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0;
    const long long nchunks = (p.nsamples + MI_CHUNK - 1) / MI_CHUNK;
    for (long long ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        for (int j = 0; j < MI_CHUNK / MI_NT; ++j) {
            const long long s = ch * MI_CHUNK + (long long)j * MI_NT + threadIdx.x;
            if (s >= p.nsamples) continue;
            int fi, mi;
            float mterm, g[3];
            double xh[3];
            if (!mi_sample(F, M, p, s, fi, mi, mterm, g, xh)) continue;
            float w = 0.0f;
#pragma unroll
            for (int k = -1; k <= 2; ++k) {
                const int b = mi + k;
                w += bspline3_deriv((float)b - mterm) * lt[fi * p.bins + b];
            }
            w *= -p.mscale;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double ga = (double)(w * g[a]);
                acc[4 * a] += ga * xh[0];
                acc[4 * a + 1] += ga * xh[1];
                acc[4 * a + 2] += ga * xh[2];
                acc[4 * a + 3] += ga;
            }
        }
    }
    __shared__ double red[MI_NT / 64][12];
#pragma unroll
    for (int k = 0; k < 12; ++k)
        for (int o = 32; o > 0; o >>= 1) acc[k] += __shfl_down(acc[k], o);
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 12; ++k) red[threadIdx.x >> 6][k] = acc[k];
    __syncthreads();
    if (threadIdx.x < 12) {
        double a = 0;
        for (int w = 0; w < MI_NT / 64; ++w) a += red[w][threadIdx.x];
        part[(size_t)blockIdx.x * 12 + threadIdx.x] = a;
    }
}

// ------------------------------------------------------------------------------------------------ Sobel magnitude
// skimage.filters.sobel on a 3-D volume (biahub/registration/ants.py:272-275): per axis the derivative [1, 0, -1] along
// it and [1, 2, 1] / 4 along the other two, mode "reflect" (== edge clamp for a radius-1 stencil); magnitude
// sqrt((gz^2 + gy^2 + gx^2) / 3).
__global__ __launch_bounds__(256) void sobel_kernel(const float* __restrict__ in, float* __restrict__ out, int Z, int Y,
                                                    int X) {
    const long long total = (long long)Z * Y * X;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % X);
        const long long t = i / X;
        const int y = (int)(t % Y), z = (int)(t / Y);
        float v[3][3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int zz = max(0, min(z + a - 1, Z - 1)), yy = max(0, min(y + b - 1, Y - 1)),
                              xx = max(0, min(x + c - 1, X - 1));
                    v[a][b][c] = in[((long long)zz * Y + yy) * X + xx];
                }
        const float sm[3] = {0.25f, 0.5f, 0.25f};
        float gz = 0, gy = 0, gx = 0;
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                gz += sm[b] * sm[c] * (v[0][b][c] - v[2][b][c]);
                gy += sm[b] * sm[c] * (v[b][0][c] - v[b][2][c]);
                gx += sm[b] * sm[c] * (v[b][c][0] - v[b][c][2]);
            }
        out[i] = sqrtf((gz * gz + gy * gy + gx * gx) * (1.0f / 3.0f));
    }
}

}  // namespace bh

using namespace bh;

extern "C" int bh_sobel(bh_ctx* ctx, const float* in, int64_t Z, int64_t Y, int64_t X, float* out) {
    BH_REQUIRE(ctx && in && out && in != out, "null or aliased argument");
    BH_REQUIRE(Z > 0 && Y > 0 && X > 0 && Z < (1ll << 31) && Y < (1ll << 31) && X < (1ll << 31), "bad shape");
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    const int grid = (int)std::min<int64_t>(ceil_div(Z * Y * X, 256), (int64_t)ctx->num_cus * 16);
    hipLaunchKernelGGL(sobel_kernel, dim3(grid), dim3(256), 0, ctx->stream, in, out, (int)Z, (int)Y, (int)X);
    BH_CHECK_HIP(hipGetLastError());
    return BH_OK;
}

extern "C" int bh_image_stats(bh_ctx* ctx, const float* in, int64_t Z, int64_t Y, int64_t X, double out[6]) {
    BH_REQUIRE(ctx && in && out, "null argument");
    BH_REQUIRE(Z > 0 && Y > 0 && X > 0 && Z < (1ll << 31) && Y < (1ll << 31) && X < (1ll << 31), "bad shape");
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    const int grid = (int)std::min<int64_t>(ceil_div(Z * Y, ST_NT / 64), (int64_t)ctx->num_cus * 8);
    double* part;
    BH_TRY(get_scratch(ctx, "reg_part", sizeof(double) * (size_t)(ctx->num_cus * 8 * 12 + 16), (void**)&part));
    double* res = part + (size_t)ctx->num_cus * 8 * 12;
    hipLaunchKernelGGL(stats_partial_kernel, dim3(grid), dim3(ST_NT), 0, ctx->stream, in, (int)Z, (int)Y, (int)X, part);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, ctx->stream, part, grid, 6, 1, res);
    BH_CHECK_HIP(hipGetLastError());
    BH_CHECK_HIP(hipMemcpyAsync(out, res, 6 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    BH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return BH_OK;
}

extern "C" int bh_smooth_shrink(bh_ctx* ctx, const float* in, int64_t Z, int64_t Y, int64_t X, const double sigma[3],
                                const int factor[3], float* out, int64_t out_shape[3], int64_t offset[3]) {
    BH_REQUIRE(sigma && factor && out_shape && offset, "null argument");
    BH_REQUIRE(Z > 0 && Y > 0 && X > 0 && Z < (1ll << 31) && Y < (1ll << 31) && X < (1ll << 31), "bad shape");
    const int64_t N[3] = {Z, Y, X};
    int R[3];
    for (int a = 0; a < 3; ++a) {
        BH_REQUIRE(factor[a] >= 1, "shrink factor must be >= 1");
        BH_REQUIRE(sigma[a] >= 0.0 && std::ceil(4.0 * sigma[a]) <= GS_MAXR, "sigma must be in [0, %d]", GS_MAXR / 4);
        R[a] = (int)std::ceil(4.0 * sigma[a]);
        out_shape[a] = std::max<int64_t>(1, N[a] / factor[a]);
        offset[a] = ((N[a] - 1) - (int64_t)factor[a] * (out_shape[a] - 1)) / 2;  // centres the kept samples
    }
    if (!out) return BH_OK;  // geometry query only
    BH_REQUIRE(ctx && in, "null argument");
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    // x, then y, then z: every pass already drops the samples the later ones do not need
    const size_t n1 = (size_t)Z * Y * out_shape[2], n2 = (size_t)Z * out_shape[1] * out_shape[2];
    float *t1, *t2;
    BH_TRY(get_scratch(ctx, "reg_gs1", n1 * sizeof(float), (void**)&t1));
    BH_TRY(get_scratch(ctx, "reg_gs2", n2 * sizeof(float), (void**)&t2));
    int64_t d[3] = {Z, Y, X};
    const float* src = in;
    for (int pass = 0; pass < 3; ++pass) {
        const int a = 2 - pass;
        GaussParams p;
        p.Z = (int)d[0], p.Y = (int)d[1], p.X = (int)d[2];
        p.axis = a, p.No = (int)out_shape[a], p.f = factor[a], p.o = (int)offset[a], p.R = R[a];
        double wsum = 0, w[2 * GS_MAXR + 1];
        for (int k = -R[a]; k <= R[a]; ++k) wsum += (w[k + R[a]] = R[a] ? std::exp(-0.5 * k * k / (sigma[a] * sigma[a])) : 1.0);
        for (int k = 0; k <= 2 * R[a]; ++k) p.w[k] = (float)(w[k] / wsum);
        float* dst = pass == 0 ? t1 : (pass == 1 ? t2 : out);
        d[a] = out_shape[a];
        const int64_t total = d[0] * d[1] * d[2];
        const int grid = (int)std::min<int64_t>(ceil_div(total, 256), (int64_t)ctx->num_cus * 16);
        hipLaunchKernelGGL(gauss_shrink_kernel, dim3(grid), dim3(256), 0, ctx->stream, src, dst, p);
        BH_CHECK_HIP(hipGetLastError());
        src = dst;
    }
    return BH_OK;
}

extern "C" int bh_mattes_mi(bh_ctx* ctx, const float* fixed, int64_t Zf, int64_t Yf, int64_t Xf, const float* moving,
                            int64_t Zm, int64_t Ym, int64_t Xm, const double P[12], const double range[4], int bins,
                            int64_t stride, int64_t offset, double* value, double grad[12], double* nvalid) {
    BH_REQUIRE(ctx && fixed && moving && P && range && value && grad && nvalid, "null argument");
    BH_REQUIRE(Zf > 0 && Yf > 0 && Xf > 0 && Zm > 0 && Ym > 0 && Xm > 0, "bad shape");
    BH_REQUIRE(Zm < (1ll << 31) && Ym < (1ll << 31) && Xm < (1ll << 31) && Zf < (1ll << 31) && Yf < (1ll << 31) &&
                   Xf < (1ll << 31),
               "bad shape");
    BH_REQUIRE(bins >= 2 * MI_PAD + 2 && bins <= MI_MAXBINS, "bins must be in [%d, %d]", 2 * MI_PAD + 2, MI_MAXBINS);
    BH_REQUIRE(stride >= 1 && offset >= 0 && offset < Zf * Yf * Xf, "bad sampling stride / offset");
    BH_REQUIRE(range[1] > range[0] && range[3] > range[2], "intensity range is empty (constant image)");
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    MiParams p;
    for (int i = 0; i < 12; ++i) p.P[i] = P[i];
    p.Zf = (int)Zf, p.Yf = (int)Yf, p.Xf = (int)Xf, p.Zm = (int)Zm, p.Ym = (int)Ym, p.Xm = (int)Xm;
    p.stride = stride, p.offset = offset;
    p.nsamples = (Zf * Yf * Xf - offset + stride - 1) / stride;
    const double fbin = (range[1] - range[0]) / (double)(bins - 2 * MI_PAD);
    const double mbin = (range[3] - range[2]) / (double)(bins - 2 * MI_PAD);
    p.fscale = (float)(1.0 / fbin), p.fnmin = (float)(range[0] / fbin - MI_PAD);
    p.mscale = (float)(1.0 / mbin), p.mnmin = (float)(range[2] / mbin - MI_PAD);
    p.bins = bins;
    const int nb2 = bins * bins;
    const int64_t nchunks = ceil_div(p.nsamples, (int64_t)MI_CHUNK);
    const int grid = (int)std::min<int64_t>(nchunks, (int64_t)ctx->num_cus * 8);
    // scratch: [hist u64 nb2][nvalid u64][pad][out double 16][table float nb2][partials double grid*12]
    char* base;
    const size_t o_hist = 0, o_nv = sizeof(unsigned long long) * nb2, o_out = o_nv + 16, o_tab = o_out + 16 * sizeof(double),
                 o_part = o_tab + sizeof(float) * ((nb2 + 3) & ~3), o_end = o_part + sizeof(double) * (size_t)ctx->num_cus * 8 * 12;
    BH_TRY(get_scratch(ctx, "reg_mi", o_end, (void**)&base));
    auto* hist = (unsigned long long*)(base + o_hist);
    auto* nv = (unsigned long long*)(base + o_nv);
    auto* outd = (double*)(base + o_out);
    auto* table = (float*)(base + o_tab);
    auto* part = (double*)(base + o_part);
    BH_CHECK_HIP(hipMemsetAsync(base, 0, o_out, ctx->stream));
    hipLaunchKernelGGL(mi_hist_kernel, dim3(grid), dim3(MI_NT), nb2 * sizeof(unsigned), ctx->stream, fixed, moving, p, hist, nv);
    hipLaunchKernelGGL(mi_table_kernel, dim3(1), dim3(1024), 0, ctx->stream, hist, bins, table, outd);
    hipLaunchKernelGGL(mi_grad_kernel, dim3(grid), dim3(MI_NT), nb2 * sizeof(float), ctx->stream, fixed, moving, p, table, part);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, ctx->stream, part, grid, 12, 0, outd + 2);
    BH_CHECK_HIP(hipGetLastError());
    double h[14];
    unsigned long long hn;
    BH_CHECK_HIP(hipMemcpyAsync(h, outd, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    BH_CHECK_HIP(hipMemcpyAsync(&hn, nv, sizeof(hn), hipMemcpyDeviceToHost, ctx->stream));
    BH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    *value = h[0];
    *nvalid = (double)hn;
    for (int i = 0; i < 12; ++i) grad[i] = h[1] > 0 ? h[2 + i] / h[1] : 0.0;
    return BH_OK;
}
