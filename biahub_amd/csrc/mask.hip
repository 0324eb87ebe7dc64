// Validity masks for crop estimation on gfx950 (biahub/estimate_crop.py:28-143).
//   :58-62   mask = (data != 0) & ~isnan(data)   per (t, first channel) volume
//   :82-94   per-volume count of valid voxels (the reference sums the boolean mask), AND of the selected masks
// One bit per voxel (64 voxels per wavefront ballot), so a position's T x 2 masks stay on the device (V / 8 bytes each)
// and the AND over the selected volumes is a word-wise pass.  Pure predicate + popcount: bit-exact against NumPy.
#include "common.hpp"

namespace bh {

template <typename T>
__device__ __forceinline__ bool voxel_valid(T v) {
    return v != (T)0;
}
template <>
__device__ __forceinline__ bool voxel_valid<float>(float v) {
    return v != 0.0f && v == v;  // NaN != 0 is true, so the NaN test is not redundant
}

// bits: ceil(n / 64) * 2 words; voxel i -> bit i % 32 of word i / 32 (padding bits 0).  partial[block] = valid voxels seen.
template <typename T>
__global__ __launch_bounds__(256) void valid_mask_kernel(const T* __restrict__ v, int64_t n, uint32_t* __restrict__ bits,
                                                         unsigned long long* __restrict__ partial) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t nseg = (n + 63) / 64;
    unsigned long long cnt = 0;
    for (int64_t seg = (int64_t)blockIdx.x * 4 + wave; seg < nseg; seg += (int64_t)gridDim.x * 4) {
        const int64_t i = seg * 64 + lane;
        const bool ok = i < n && voxel_valid<T>(v[i]);
        const unsigned long long b = __ballot(ok);
        if (lane == 0) {
            bits[2 * seg] = (uint32_t)b;
            bits[2 * seg + 1] = (uint32_t)(b >> 32);
            cnt += __popcll(b);
        }
    }
    __shared__ unsigned long long sh[4];
    if (lane == 0) sh[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ void sum_partials_kernel(const unsigned long long* __restrict__ partial, int n, unsigned long long* __restrict__ out) {
    __shared__ unsigned long long sh[256];
    unsigned long long s = 0;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = sh[0];
}

__global__ void bits_and_kernel(uint32_t* __restrict__ acc, const uint32_t* __restrict__ src, int64_t nwords) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (int64_t)gridDim.x * blockDim.x)
        acc[i] &= src[i];
}

__global__ void bits_unpack_kernel(const uint32_t* __restrict__ bits, int64_t n, uint8_t* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (bits[i >> 5] >> (i & 31)) & 1u;
}

}  // namespace bh

using namespace bh;

static dim3 mask_grid(bh_ctx* ctx, int64_t n) {
    return dim3((unsigned)std::min<int64_t>(ceil_div(n, 256), (int64_t)ctx->num_cus * 16));
}

extern "C" {

int bh_valid_mask(bh_ctx* ctx, const void* vol, int dtype, int64_t n, uint32_t* bits, uint64_t* count) {
    BH_REQUIRE(ctx && vol && bits && count, "NULL argument");
    BH_REQUIRE(n > 0, "empty volume");
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    const int nblk = ctx->num_cus * 8;
    unsigned long long *partial, *total;
    BH_TRY(get_scratch(ctx, "mask_partial", (size_t)(nblk + 1) * sizeof(unsigned long long), (void**)&partial));
    total = partial + nblk;
    hipStream_t s = ctx->stream;
#define BH_MASK(T) hipLaunchKernelGGL(valid_mask_kernel<T>, dim3(nblk), dim3(256), 0, s, (const T*)vol, n, bits, partial)
    switch (dtype) {
        case BH_DT_F32: BH_MASK(float); break;
        case BH_DT_U16: BH_MASK(uint16_t); break;
        case BH_DT_I16: BH_MASK(int16_t); break;
        case BH_DT_U8: BH_MASK(uint8_t); break;
        default: set_error("bh_valid_mask: unsupported dtype code %d", dtype); return BH_ERR_UNSUPPORTED;
    }
#undef BH_MASK
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, partial, nblk, total);
    BH_CHECK_HIP(hipGetLastError());
    unsigned long long h = 0;
    BH_CHECK_HIP(hipMemcpyAsync(&h, total, sizeof(h), hipMemcpyDeviceToHost, s));
    BH_CHECK_HIP(hipStreamSynchronize(s));
    *count = h;
    return BH_OK;
}

int bh_bits_and(bh_ctx* ctx, uint32_t* acc, const uint32_t* src, int64_t nwords) {
    BH_REQUIRE(ctx && acc && src, "NULL argument");
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    if (nwords <= 0) return BH_OK;
    hipLaunchKernelGGL(bits_and_kernel, mask_grid(ctx, nwords), dim3(256), 0, ctx->stream, acc, src, nwords);
    BH_CHECK_HIP(hipGetLastError());
    return BH_OK;
}

int bh_bits_unpack(bh_ctx* ctx, const uint32_t* bits, int64_t n, uint8_t* out) {
    BH_REQUIRE(ctx && bits && out, "NULL argument");
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    if (n <= 0) return BH_OK;
    hipLaunchKernelGGL(bits_unpack_kernel, mask_grid(ctx, n), dim3(256), 0, ctx->stream, bits, n, out);
    BH_CHECK_HIP(hipGetLastError());
    return BH_OK;
}

}  // extern "C"
