// Cubic B-spline resampling for gfx950 — SciPy's `affine_transform(order=3, mode="constant")` as the reference calls it:
//   biahub/core/transform.py:374-396  Transform._apply_scipy (any order; 3 here, 0 / 1 in affine.hip)
//   biahub/register.py:271-272        apply_affine_transform(method="scipy") -> scipy.ndimage.affine_transform(zyx, matrix, ...)
//
// SciPy's algorithm (ni_splines.c, ni_interpolation.c), restated in oracle/oracle_np.py:spline_*:
//   1. prefilter: per axis, c = 6 s; causal c[i] += z c[i-1]; anticausal c[i] = z (c[i+1] - c[i]); pole z = sqrt(3) - 2, mirror
//      boundary (whole-sample symmetric) — the exact inverse of sampling the cubic B-spline at the integers;
//   2. out(p) = sum of the 4 x 4 x 4 taps floor(c) - 1 .. floor(c) + 2 with B-spline weights, taps mirrored at the edges; cval
//      where a coordinate leaves [0, n - 1].
//
// The recursion is a two-sided exponential filter h[k] ~ z^|k| on the mirror-extended samples: |z| = 0.268, so z^20 = 3.6e-12 —
// below float32 resolution.  Every thread therefore filters a BLOCK of B outputs along the axis from B + 2 R inputs (R = 20
// run-in samples each side, mirrored indices past the ends) held in registers: no serial dependency across the volume, no
// second sweep, every load independent of the recursion.  Along y and z the lanes run along x (coalesced loads straight into
// registers); along x a workgroup stages its rows in LDS (skewed by one word per 16 so that 64 lanes reading at a stride of 16
// words hit 64 different banks).  Three out-of-place passes (in -> A -> B -> A), float32 coefficients.
#include "common.hpp"

#include <cmath>

namespace bh {

namespace sp {

constexpr int R = 20;                             // run-in samples (truncation z^R = 3.6e-12)
constexpr float Zp = -0.26794919243112270647f;    // sqrt(3) - 2
constexpr float GAIN = 6.0f;                      // (1 - z)(1 - 1/z)
constexpr float INIT_C = 1.0f / (1.0f - Zp);      // steady state of the causal recursion on a constant signal
constexpr float INIT_A = -Zp / (1.0f - Zp);       // ... of the anticausal one

__host__ __device__ __forceinline__ int mirror(int i, int n) {  // whole-sample symmetric extension, any i
    if (n <= 1) return 0;
    const int s2 = 2 * n - 2;
    i %= s2;
    if (i < 0) i += s2;
    return i >= n ? s2 - i : i;
}

template <typename T>
__device__ __forceinline__ float clean(T v) {
    return (float)v;
}
template <>
__device__ __forceinline__ float clean<float>(float v) {  // np.nan_to_num(nan=0) (register.py:254): NaN -> 0, +-inf -> +-FLT_MAX
    if (v != v) return 0.0f;
    return fminf(fmaxf(v, -3.4028234663852886e38f), 3.4028234663852886e38f);
}

// s[0 .. B + 2 R): samples at positions p0 - R .. p0 + B + R - 1 (mirror-extended).  On return s[R .. R + B) hold the
// coefficients of positions p0 .. p0 + B - 1.
template <int B>
__device__ __forceinline__ void filter_block(float (&s)[B + 2 * R]) {
    constexpr int L = B + 2 * R;
    s[0] = GAIN * s[0] * INIT_C;
#pragma unroll
    for (int i = 1; i < L; ++i) s[i] = __builtin_fmaf(Zp, s[i - 1], GAIN * s[i]);
    s[L - 1] = INIT_A * s[L - 1];
#pragma unroll
    for (int i = L - 2; i >= R; --i) s[i] = Zp * (s[i + 1] - s[i]);
}

// ---- pass along y or z: lanes along x, a thread filters B outputs of one column --------------------------------------------
// element (o, i, x) at src[o * ostride + i * astride + x]; n = axis length, nblk = ceil(n / B) blocks per column
template <int B>
__global__ __launch_bounds__(256) void axis_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, long astride,
                                                   long ostride, int nouter, int X, int nblk) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int blk = blockIdx.y % nblk, o = blockIdx.y / nblk;
    if (x >= X || o >= nouter) return;
    const float* col = src + (long)o * ostride + x;
    const int p0 = blk * B;
    float s[B + 2 * R];
    if (p0 - R >= 0 && p0 + B + R <= n) {  // interior block: no index arithmetic
        const float* q = col + (long)(p0 - R) * astride;
#pragma unroll
        for (int i = 0; i < B + 2 * R; ++i) s[i] = q[(long)i * astride];
    } else {
#pragma unroll
        for (int i = 0; i < B + 2 * R; ++i) s[i] = col[(long)mirror(p0 - R + i, n) * astride];
    }
    filter_block<B>(s);
    float* out = dst + (long)o * ostride + x;
#pragma unroll
    for (int i = 0; i < B; ++i)
        if (p0 + i < n) out[(long)(p0 + i) * astride] = s[R + i];
}

// ---- pass along x: rows staged in LDS, a thread filters BX consecutive outputs of a row ------------------------------------
constexpr int BX = 16;
constexpr int XCH = 4096;  // outputs of one row chunk (one workgroup step)
__host__ __device__ __forceinline__ int skew(int i) { return i + (i >> 4); }  // one pad word per 16: lanes 16 words apart -> 17

template <typename TIN>
__global__ __launch_bounds__(256) void x_kernel(const TIN* __restrict__ src, float* __restrict__ dst, long rows, int X, int nchunk,
                                                int rpw, int tpr) {
    extern __shared__ float lds[];
    const int clen = min(X, XCH);                // outputs per chunk
    const int span = clen + 2 * R;               // staged samples per row
    const int pitch = skew(span) + 1;
    const long unit0 = (long)blockIdx.x * rpw;   // first (row, chunk) unit of this workgroup
    const long nunits = rows * nchunk;
    // stage: rpw units, lanes along x
    for (int u = 0; u < rpw; ++u) {
        const long unit = unit0 + u;
        if (unit >= nunits) break;
        const long row = unit / nchunk;
        const int p0 = (int)(unit % nchunk) * XCH;
        const TIN* rp = src + row * X;
        float* lr = lds + u * pitch;
        for (int i = threadIdx.x; i < span; i += 256) lr[skew(i)] = clean<TIN>(rp[mirror(p0 - R + i, X)]);
    }
    __syncthreads();
    const int u = threadIdx.x / tpr, t = threadIdx.x % tpr;
    const long unit = unit0 + u;
    if (u >= rpw || unit >= nunits) return;
    const long row = unit / nchunk;
    const int p0 = (int)(unit % nchunk) * XCH;
    const float* lr = lds + u * pitch;
    for (int b = t; b * BX < clen; b += tpr) {
        float s[BX + 2 * R];
#pragma unroll
        for (int i = 0; i < BX + 2 * R; ++i) s[i] = lr[skew(min(b * BX + i, span - 1))];
        filter_block<BX>(s);
        float* out = dst + row * X + p0 + b * BX;
#pragma unroll
        for (int i = 0; i < BX; ++i)
            if (b * BX + i < clen && p0 + b * BX + i < X) out[i] = s[R + i];
    }
}

// conversion only (an axis of length 1 is not filtered — SciPy leaves such lines alone)
template <typename TIN>
__global__ void convert_kernel(const TIN* __restrict__ src, float* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = clean<TIN>(src[i]);
}

// ---- interpolation ------------------------------------------------------------------------------------------------------
struct GatherParams {
    double m[12];
    int Zi, Yi, Xi, Zo, Yo, Xo, cz, cy, cx;
    float cval;
};

__device__ __forceinline__ void weights3(double c, int& base, float (&w)[4]) {
    const double fl = floor(c);
    const double x = c - fl, z = 1.0 - x;
    const double w1 = (x * x * (x - 2.0) * 3.0 + 4.0) / 6.0;
    const double w2 = (z * z * (z - 2.0) * 3.0 + 4.0) / 6.0;
    const double w0 = z * z * z / 6.0;
    base = (int)fl - 1;
    w[0] = (float)w0;
    w[1] = (float)w1;
    w[2] = (float)w2;
    w[3] = (float)(1.0 - w0 - w1 - w2);
}

// One output voxel per thread, 8 x 8 x 4 voxel blocks per workgroup (x fastest): the 64 taps come through the vector cache,
// which neighbouring voxels of a block share.
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ coef, float* __restrict__ out, GatherParams p) {
#pragma clang fp contract(off)
    const int ox = blockIdx.x * 8 + (threadIdx.x & 7);
    const int oy = blockIdx.y * 8 + ((threadIdx.x >> 3) & 7);
    const int oz = blockIdx.z * 4 + (threadIdx.x >> 6);
    if (ox >= p.Xo || oy >= p.Yo || oz >= p.Zo) return;
    const double gz = (double)(oz + p.cz), gy = (double)(oy + p.cy), gx = (double)(ox + p.cx);
    const double c0 = p.m[0] * gz + p.m[1] * gy + p.m[2] * gx + p.m[3];
    const double c1 = p.m[4] * gz + p.m[5] * gy + p.m[6] * gx + p.m[7];
    const double c2 = p.m[8] * gz + p.m[9] * gy + p.m[10] * gx + p.m[11];
    float r = p.cval;
    // SciPy "constant": a coordinate outside [0, n - 1] gives cval (no interpolation past the edge samples)
    if (c0 >= 0.0 && c0 <= (double)(p.Zi - 1) && c1 >= 0.0 && c1 <= (double)(p.Yi - 1) && c2 >= 0.0 && c2 <= (double)(p.Xi - 1)) {
        int bz, by, bx;
        float wz[4], wy[4], wx[4];
        weights3(c0, bz, wz);
        weights3(c1, by, wy);
        weights3(c2, bx, wx);
        // tap indices: inside the volume as they are, mirrored about the edge samples otherwise (one test per axis)
        int ix[4], iy[4], iz[4];
        const bool xin = bx >= 0 && bx + 3 < p.Xi, yin = by >= 0 && by + 3 < p.Yi, zin = bz >= 0 && bz + 3 < p.Zi;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ix[k] = xin ? bx + k : mirror(bx + k, p.Xi);
            iy[k] = yin ? by + k : mirror(by + k, p.Yi);
            iz[k] = zin ? bz + k : mirror(bz + k, p.Zi);
        }
        float acc = 0.0f;
#pragma unroll
        for (int dz = 0; dz < 4; ++dz) {
            float az = 0.0f;
#pragma unroll
            for (int dy = 0; dy < 4; ++dy) {
                const float* rowp = coef + ((long)iz[dz] * p.Yi + iy[dy]) * p.Xi;
                float ax = wx[0] * rowp[ix[0]];
                ax = __builtin_fmaf(wx[1], rowp[ix[1]], ax);
                ax = __builtin_fmaf(wx[2], rowp[ix[2]], ax);
                ax = __builtin_fmaf(wx[3], rowp[ix[3]], ax);
                az = __builtin_fmaf(wy[dy], ax, az);
            }
            acc = __builtin_fmaf(wz[dz], az, acc);
        }
        r = acc;
    }
    out[((long)oz * p.Yo + oy) * p.Xo + ox] = r;
}

template <typename TIN>
static int prefilter_typed(bh_ctx* ctx, const TIN* in, int64_t Z, int64_t Y, int64_t X, float* coef, float* tmp) {
    hipStream_t s = ctx->stream;
    const long rows = Z * Y, n = rows * X;
    constexpr int B = 64;
    // x: in -> coef
    if (X > 1) {
        const int clen = (int)std::min<int64_t>(X, XCH), nchunk = (int)ceil_div(X, XCH);
        const int tpr = std::min(256, (int)ceil_div(clen, BX));
        const int span = clen + 2 * R, pitch = skew(span) + 1;
        int rpw = std::max(1, 256 / tpr);
        rpw = std::min<int>(rpw, std::max(1, (int)(160 * 1024 / 4 / pitch) - 0));
        const size_t lds = (size_t)rpw * pitch * sizeof(float);
        BH_REQUIRE(lds <= 160 * 1024, "spline prefilter row chunk does not fit LDS");
        const long nunits = rows * nchunk;
        const long grid = ceil_div(nunits, rpw);
        BH_REQUIRE(grid < (1ll << 31), "volume too large for the spline prefilter");
        auto kern = x_kernel<TIN>;
        if (lds > 64 * 1024)
            BH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, in, coef, rows, (int)X, nchunk, rpw, tpr);
    } else {
        hipLaunchKernelGGL(convert_kernel<TIN>, dim3((unsigned)std::min<long>(ceil_div(n, 256), 65535)), dim3(256), 0, s, in, coef, n);
    }
    float *a = coef, *b = tmp;
    auto axis = [&](int len, long astride, long ostride, int nouter) -> int {
        if (len <= 1) return BH_OK;
        const int nblk = (int)ceil_div(len, B);
        const long gy = (long)nouter * nblk;
        BH_REQUIRE(gy <= 65535ll * 32768ll, "volume too large for the spline prefilter");
        // grid.y is limited to 65535: fold the rest into z-less launches per slab of outer indices
        const int max_o = std::max(1, 65535 / nblk);
        for (int o0 = 0; o0 < nouter; o0 += max_o) {
            const int no = std::min(max_o, nouter - o0);
            hipLaunchKernelGGL(axis_kernel<B>, dim3((unsigned)ceil_div(X, 256), (unsigned)(no * nblk)), dim3(256), 0, s,
                               a + (long)o0 * ostride, b + (long)o0 * ostride, len, astride, ostride, no, (int)X, nblk);
        }
        std::swap(a, b);
        return BH_OK;
    };
    BH_TRY(axis((int)Y, X, (long)Y * X, (int)Z));  // y: columns (z, x)
    BH_TRY(axis((int)Z, (long)Y * X, X, (int)Y));  // z: columns (y, x)
    if (a != coef) BH_CHECK_HIP(hipMemcpyAsync(coef, a, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, s));
    BH_CHECK_HIP(hipGetLastError());
    return BH_OK;
}

int prefilter(bh_ctx* ctx, const void* in, int in_dtype, int64_t Z, int64_t Y, int64_t X, float* coef) {
    float* tmp = nullptr;
    if (Y > 1 || Z > 1) BH_TRY(get_scratch(ctx, "spline_tmp", (size_t)Z * Y * X * sizeof(float), (void**)&tmp));
    switch (in_dtype) {
        case BH_DT_F32: return prefilter_typed(ctx, (const float*)in, Z, Y, X, coef, tmp);
        case BH_DT_U16: return prefilter_typed(ctx, (const uint16_t*)in, Z, Y, X, coef, tmp);
        case BH_DT_U8: return prefilter_typed(ctx, (const uint8_t*)in, Z, Y, X, coef, tmp);
        case BH_DT_I16: return prefilter_typed(ctx, (const int16_t*)in, Z, Y, X, coef, tmp);
        default: BH_REQUIRE(false, "unsupported input dtype code %d", in_dtype);
    }
    return BH_OK;
}

}  // namespace sp

// the cubic branch of bh_affine (affine.hip): prefilter into the context's scratch, then gather
int affine_cubic(bh_ctx* ctx, const void* in, int in_dtype, int64_t Zi, int64_t Yi, int64_t Xi, const double matrix[12], float cval,
                 float* out, int64_t Zo, int64_t Yo, int64_t Xo, const int64_t crop_lo[3]) {
    float* coef = nullptr;
    BH_TRY(get_scratch(ctx, "spline_coef", (size_t)Zi * Yi * Xi * sizeof(float), (void**)&coef));
    BH_TRY(sp::prefilter(ctx, in, in_dtype, Zi, Yi, Xi, coef));
    sp::GatherParams p;
    for (int i = 0; i < 12; ++i) p.m[i] = matrix[i];
    p.Zi = (int)Zi;
    p.Yi = (int)Yi;
    p.Xi = (int)Xi;
    p.Zo = (int)Zo;
    p.Yo = (int)Yo;
    p.Xo = (int)Xo;
    p.cz = crop_lo ? (int)crop_lo[0] : 0;
    p.cy = crop_lo ? (int)crop_lo[1] : 0;
    p.cx = crop_lo ? (int)crop_lo[2] : 0;
    p.cval = cval;
    const dim3 grid((unsigned)ceil_div(Xo, 8), (unsigned)ceil_div(Yo, 8), (unsigned)ceil_div(Zo, 4));
    BH_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "affine output too large");
    hipLaunchKernelGGL(sp::gather_kernel, grid, dim3(256), 0, ctx->stream, coef, out, p);
    BH_CHECK_HIP(hipGetLastError());
    return BH_OK;
}

}  // namespace bh

extern "C" int bh_spline_prefilter(bh_ctx* ctx, const void* in, int in_dtype, int64_t Z, int64_t Y, int64_t X, float* coef) {
    BH_REQUIRE(ctx && in && coef, "NULL argument");
    BH_REQUIRE(Z > 0 && Y > 0 && X > 0 && Z < (1ll << 30) && Y < (1ll << 30) && X < (1ll << 30), "invalid shape");
    BH_REQUIRE((const void*)coef != in, "the prefilter runs out of place");
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    return bh::sp::prefilter(ctx, in, in_dtype, Z, Y, X, coef);
}
