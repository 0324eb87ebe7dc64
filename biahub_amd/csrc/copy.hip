// Bit-exact crop / flip (pure indexing) on gfx950.
//   biahub/utils/array_ops.py:9-59  copy_n_paste / copy_n_paste_czyx  (crop, ZYX variant NaN->0)
//   biahub/flip.py:22-32            [:, :, ::-1] and/or [:, ::-1, :]
// One thread per output element, X fastest so reads and writes are row-contiguous (a flipped
// row is read back-to-front inside the same cache lines).  No arithmetic touches the payload
// (except the optional NaN->0), so results equal numpy slicing bit for bit.
#include "common.hpp"

namespace bh {

struct CopyParams {
    int64_t C, Zi, Yi, Xi, Zo, Yo, Xo;
    int64_t lz, ly, lx;
    int flip_y, flip_x, nan_to_zero;
};

template <typename T>
__device__ __forceinline__ T nan_clean(T v, int) {
    return v;
}
template <>
__device__ __forceinline__ uint32_t nan_clean<uint32_t>(uint32_t v, int on) {
    // np.nan_to_num(nan=0): NaN -> 0, +-inf -> +-FLT_MAX, everything else untouched
    if (!on || (v & 0x7F800000u) != 0x7F800000u) return v;
    return (v & 0x007FFFFFu) ? 0u : ((v & 0x80000000u) | 0x7F7FFFFFu);
}
template <>
__device__ __forceinline__ uint64_t nan_clean<uint64_t>(uint64_t v, int on) {
    if (!on || (v & 0x7FF0000000000000ull) != 0x7FF0000000000000ull) return v;
    return (v & 0x000FFFFFFFFFFFFFull) ? 0ull : ((v & 0x8000000000000000ull) | 0x7FEFFFFFFFFFFFFFull);
}

template <typename T>
__global__ __launch_bounds__(256) void crop_flip_kernel(const T* __restrict__ in, T* __restrict__ out, CopyParams p) {
    const int64_t rows = p.C * p.Zo * p.Yo;
    for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
        const int64_t y = row % p.Yo, z = (row / p.Yo) % p.Zo, c = row / (p.Yo * p.Zo);
        const int64_t sy = p.ly + (p.flip_y ? p.Yo - 1 - y : y);
        const T* src = in + ((c * p.Zi + (p.lz + z)) * p.Yi + sy) * p.Xi + p.lx;
        T* dst = out + row * p.Xo;
        for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < p.Xo; x += (int64_t)gridDim.x * blockDim.x) {
            const int64_t sx = p.flip_x ? p.Xo - 1 - x : x;
            dst[x] = nan_clean<T>(src[sx], p.nan_to_zero);
        }
    }
}

template <typename T>
static int launch_copy(bh_ctx* ctx, const void* in, void* out, const CopyParams& p) {
    const int64_t rows = p.C * p.Zo * p.Yo;
    dim3 grid((unsigned)std::min<int64_t>(ceil_div(p.Xo, 256), 64), (unsigned)std::min<int64_t>(rows, 65535));
    hipLaunchKernelGGL(crop_flip_kernel<T>, grid, dim3(256), 0, ctx->stream, (const T*)in, (T*)out, p);
    BH_CHECK_HIP(hipGetLastError());
    return BH_OK;
}

}  // namespace bh

extern "C" int bh_crop_flip(bh_ctx* ctx, const void* in, int itemsize, int64_t C, int64_t Zi, int64_t Yi, int64_t Xi,
                            const int64_t lo[3], int64_t Zo, int64_t Yo, int64_t Xo, int flip_y, int flip_x,
                            int nan_to_zero, void* out) {
    using namespace bh;
    BH_REQUIRE(ctx && in && out, "NULL argument");
    BH_REQUIRE(C > 0 && Zi > 0 && Yi > 0 && Xi > 0, "invalid input shape");
    CopyParams p;
    p.C = C;
    p.Zi = Zi;
    p.Yi = Yi;
    p.Xi = Xi;
    p.Zo = Zo;
    p.Yo = Yo;
    p.Xo = Xo;
    p.lz = lo ? lo[0] : 0;
    p.ly = lo ? lo[1] : 0;
    p.lx = lo ? lo[2] : 0;
    p.flip_y = flip_y;
    p.flip_x = flip_x;
    p.nan_to_zero = nan_to_zero;
    BH_REQUIRE(Zo >= 0 && Yo >= 0 && Xo >= 0, "negative output shape");
    BH_REQUIRE(p.lz >= 0 && p.ly >= 0 && p.lx >= 0 && p.lz + Zo <= Zi && p.ly + Yo <= Yi && p.lx + Xo <= Xi,
               "crop box [%lld:%lld, %lld:%lld, %lld:%lld] outside input (%lld,%lld,%lld)", (long long)p.lz,
               (long long)(p.lz + Zo), (long long)p.ly, (long long)(p.ly + Yo), (long long)p.lx,
               (long long)(p.lx + Xo), (long long)Zi, (long long)Yi, (long long)Xi);
    BH_REQUIRE(!nan_to_zero || itemsize == 4 || itemsize == 8, "nan_to_zero needs a 4- or 8-byte float type");
    if (Zo == 0 || Yo == 0 || Xo == 0) return BH_OK;  // empty crop: nothing to move
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    ScopedTimer timer(ctx, T_CROPFLIP);
    switch (itemsize) {
        case 1: return launch_copy<uint8_t>(ctx, in, out, p);
        case 2: return launch_copy<uint16_t>(ctx, in, out, p);
        case 4: return launch_copy<uint32_t>(ctx, in, out, p);
        case 8: return launch_copy<uint64_t>(ctx, in, out, p);
        default: BH_REQUIRE(false, "unsupported itemsize %d (1, 2, 4 or 8)", itemsize);
    }
    return BH_OK;
}
