// Binning (biahub/process_data.py:29-105 binning_czyx): sum or mean over (bz, by, bx) windows, then the reference's range
// normalisation and cast back to the input dtype.  Two flat calls so that the host can apply the reference's rules, which
// differ per mode (sum: per-channel min/max stretch to the dtype range; mean: integer dtypes scaled by the maximum over
// ALL channels):
//   bh_bin_reduce   binned float32 volume (sum, or sum / count) + its min and max
//   bh_bin_finish   out = cast((v - sub) * mul / div) in float32, in the reference's operation order (no contraction),
//                   truncating like ndarray.astype
// Integer inputs with windows of <= 256 samples sum exactly in float32, so the result is bit-identical to numpy's.
#include "common.hpp"

#include <algorithm>

namespace bh {

template <typename TIN>
__global__ __launch_bounds__(256) void bin_reduce_kernel(const TIN* __restrict__ in, int Y, int X, int Zb, int Yb, int Xb,
                                                         int fz, int fy, int fx, int mean, float* __restrict__ out,
                                                         float* __restrict__ part) {
    const long long total = (long long)Zb * Yb * Xb;
    float mn = INFINITY, mx = -INFINITY;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % Xb), y = (int)((i / Xb) % Yb), z = (int)(i / ((long long)Xb * Yb));
        float s = 0.0f;
        for (int a = 0; a < fz; ++a)
            for (int b = 0; b < fy; ++b) {
                const TIN* row = in + ((size_t)(z * fz + a) * Y + (size_t)(y * fy + b)) * X + (size_t)x * fx;
                for (int c = 0; c < fx; ++c) s += (float)row[c];
            }
        if (mean) s = s / (float)(fz * fy * fx);  // float32 division by the count, like numpy's mean (not a reciprocal)
        out[i] = s;
        mn = fminf(mn, s);
        mx = fmaxf(mx, s);
    }
    __shared__ float smn[256], smx[256];
    smn[threadIdx.x] = mn, smx[threadIdx.x] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            smn[threadIdx.x] = fminf(smn[threadIdx.x], smn[threadIdx.x + o]);
            smx[threadIdx.x] = fmaxf(smx[threadIdx.x], smx[threadIdx.x + o]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) part[2 * blockIdx.x] = smn[0], part[2 * blockIdx.x + 1] = smx[0];
}

__global__ __launch_bounds__(256) void bin_minmax_final_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
    __shared__ float smn[256], smx[256];
    float mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 256) mn = fminf(mn, part[2 * i]), mx = fmaxf(mx, part[2 * i + 1]);
    smn[threadIdx.x] = mn, smx[threadIdx.x] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            smn[threadIdx.x] = fminf(smn[threadIdx.x], smn[threadIdx.x + o]);
            smx[threadIdx.x] = fmaxf(smx[threadIdx.x], smx[threadIdx.x + o]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = smn[0], out[1] = smx[0];
}

template <typename TOUT>
__global__ __launch_bounds__(256) void bin_finish_kernel(const float* __restrict__ v, long long n, int apply, float sub,
                                                         float mul, float div, TOUT* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float r = v[i];
        if (apply) {
#pragma clang fp contract(off)
            r = r - sub;
            r = r * mul;
            r = r / div;
        }
        out[i] = (TOUT)r;  // truncation toward zero, like ndarray.astype
    }
}

}  // namespace bh

using namespace bh;

extern "C" int bh_bin_reduce(bh_ctx* ctx, const void* in, int in_dtype, int64_t Z, int64_t Y, int64_t X, const int factor[3],
                             int mean, float* out, float minmax[2]) {
    BH_REQUIRE(ctx && in && factor && out && minmax, "null argument");
    BH_REQUIRE(Z > 0 && Y > 0 && X > 0 && Z < (1ll << 31) && Y < (1ll << 31) && X < (1ll << 31), "bad shape");
    BH_REQUIRE(factor[0] >= 1 && factor[1] >= 1 && factor[2] >= 1, "binning factors must be >= 1");
    const int64_t Zb = Z / factor[0], Yb = Y / factor[1], Xb = X / factor[2];
    BH_REQUIRE(Zb > 0 && Yb > 0 && Xb > 0, "binning factor larger than the volume");
    // the reference reshapes to (Zb, fz, Yb, fy, Xb, fx): numpy raises when the shape is not divisible
    BH_REQUIRE(Zb * factor[0] == Z && Yb * factor[1] == Y && Xb * factor[2] == X,
               "cannot reshape array of size %lld into shape (%lld,%d,%lld,%d,%lld,%d)", (long long)(Z * Y * X),
               (long long)Zb, factor[0], (long long)Yb, factor[1], (long long)Xb, factor[2]);
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    const int64_t total = Zb * Yb * Xb;
    const int grid = (int)std::min<int64_t>(ceil_div(total, 256), (int64_t)ctx->num_cus * 16);
    float* part;
    BH_TRY(get_scratch(ctx, "bin_part", sizeof(float) * (size_t)(2 * grid + 2), (void**)&part));
#define BH_BIN(T)                                                                                                   \
    hipLaunchKernelGGL(bin_reduce_kernel<T>, dim3(grid), dim3(256), 0, ctx->stream, (const T*)in, (int)Y, (int)X,   \
                       (int)Zb, (int)Yb, (int)Xb, factor[0], factor[1], factor[2], mean, out, part)
    switch (in_dtype) {
        case BH_DT_U8: BH_BIN(uint8_t); break;
        case BH_DT_U16: BH_BIN(uint16_t); break;
        case BH_DT_I16: BH_BIN(int16_t); break;
        case BH_DT_F32: BH_BIN(float); break;
        default: BH_REQUIRE(false, "unsupported input dtype code %d", in_dtype);
    }
#undef BH_BIN
    hipLaunchKernelGGL(bin_minmax_final_kernel, dim3(1), dim3(256), 0, ctx->stream, (const float*)part, grid, part + 2 * grid);
    BH_CHECK_HIP(hipGetLastError());
    BH_CHECK_HIP(hipMemcpyAsync(minmax, part + 2 * grid, 2 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    BH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return BH_OK;
}

extern "C" int bh_bin_finish(bh_ctx* ctx, const float* v, int64_t n, int apply, float sub, float mul, float div,
                             int out_dtype, void* out) {
    BH_REQUIRE(ctx && v && out && n > 0, "null argument");
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    const int grid = (int)std::min<int64_t>(ceil_div(n, 256), (int64_t)ctx->num_cus * 16);
#define BH_FIN(T) \
    hipLaunchKernelGGL(bin_finish_kernel<T>, dim3(grid), dim3(256), 0, ctx->stream, v, (long long)n, apply, sub, mul, div, (T*)out)
    switch (out_dtype) {
        case BH_DT_U8: BH_FIN(uint8_t); break;
        case BH_DT_U16: BH_FIN(uint16_t); break;
        case BH_DT_I16: BH_FIN(int16_t); break;
        case BH_DT_F32: BH_FIN(float); break;
        default: BH_REQUIRE(false, "unsupported output dtype code %d", out_dtype);
    }
#undef BH_FIN
    BH_CHECK_HIP(hipGetLastError());
    return BH_OK;
}
