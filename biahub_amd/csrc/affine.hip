// 3-D affine pull-resample on gfx950 with LDS-tiled trilinear sampling.
//
//   out(p) = in(M p),  p = (z, y, x, 1) in index space, M a 3x4 pull matrix.
//
// Replaces ANTsTransform.apply_to_image (biahub/register.py:261-269, stabilize.py:82-88: ITK
// ResampleImageFilter, origin 0 / spacing 1 / identity direction because both images come from
// ants.from_numpy) and scipy.ndimage.affine_transform (core/transform.py:384-396).  NaN -> 0
// (register.py:254) is folded into the load.
//
// A workgroup owns a 8 x 4 x 64 (z, y, x) output tile.  It maps the tile's corners through M,
// stages the source bounding box (clipped to the volume) into LDS with row-contiguous reads and
// samples from LDS; when the box does not fit (strong scale/rotation) it gathers from global
// memory through L2 instead.  Coordinates are float64 (ITK and SciPy both use doubles), weights
// and accumulation float32.
#include "common.hpp"

namespace bh {

constexpr int ATX = 64, ATY = 4, ATZ = 8;
constexpr int A_LDS_FLOATS = 12288;  // 48 KiB -> three workgroups per CU

struct AffineParams {
    double m[12];
    int Zi, Yi, Xi;
    int Zo, Yo, Xo;
    int cz, cy, cx;
    int interp, boundary;
    float cval;
};

template <typename T>
__device__ __forceinline__ float load_clean(const T* p) {
    return (float)*p;
}
template <>
__device__ __forceinline__ float load_clean<float>(const float* p) {
    const float v = *p;  // np.nan_to_num(nan=0): NaN -> 0, +-inf -> +-FLT_MAX
    if (v != v) return 0.0f;
    return fminf(fmaxf(v, -3.4028234663852886e38f), 3.4028234663852886e38f);
}

__device__ __forceinline__ void map_point(const double* m, double z, double y, double x, double c[3]) {
#pragma clang fp contract(off)
    c[0] = m[0] * z + m[1] * y + m[2] * x + m[3];
    c[1] = m[4] * z + m[5] * y + m[6] * x + m[7];
    c[2] = m[8] * z + m[9] * y + m[10] * x + m[11];
}

template <typename TIN>
__global__ __launch_bounds__(256) void affine_kernel(const TIN* __restrict__ in, float* __restrict__ out,
                                                     AffineParams p) {
    __shared__ float tile[A_LDS_FLOATS];
    __shared__ int box[6];
    const int tx = threadIdx.x & 63;
    const int ty = threadIdx.x >> 6;
    const int ox0 = blockIdx.x * ATX, oy0 = blockIdx.y * ATY, oz0 = blockIdx.z * ATZ;

    // source bounding box of this tile (uniform across the block)
    if (threadIdx.x == 0) {
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        const int z1 = min(oz0 + ATZ, p.Zo) - 1, y1 = min(oy0 + ATY, p.Yo) - 1, x1 = min(ox0 + ATX, p.Xo) - 1;
        for (int c = 0; c < 8; ++c) {
            double q[3];
            map_point(p.m, (double)((c & 4 ? z1 : oz0) + p.cz), (double)((c & 2 ? y1 : oy0) + p.cy),
                      (double)((c & 1 ? x1 : ox0) + p.cx), q);
            for (int a = 0; a < 3; ++a) {
                lo[a] = fmin(lo[a], q[a]);
                hi[a] = fmax(hi[a], q[a]);
            }
        }
        const int dims[3] = {p.Zi, p.Yi, p.Xi};
        for (int a = 0; a < 3; ++a) {
            // nearest needs floor(c+0.5); linear needs floor(c) and floor(c)+1: [floor(lo), floor(hi)+1] covers both
            double l = floor(lo[a]), h = floor(hi[a]) + 1.0;
            l = fmax(l, 0.0);
            h = fmin(h, (double)(dims[a] - 1));
            box[a] = (int)l;
            box[3 + a] = (h >= l) ? (int)(h - l) + 1 : 0;
        }
    }
    __syncthreads();
    const int bz = box[0], by = box[1], bx = box[2];
    const int dz = box[3], dy = box[4], dx = box[5];
    const int64_t nbox = (int64_t)dz * dy * dx;
    const bool staged = nbox > 0 && nbox <= A_LDS_FLOATS;
    const size_t sY = (size_t)p.Xi, sZ = (size_t)p.Yi * p.Xi;
    if (staged) {
        const int n = (int)nbox;
        for (int i = threadIdx.x; i < n; i += 256) {
            const int x = i % dx, y = (i / dx) % dy, z = i / (dx * dy);
            tile[i] = load_clean(in + (size_t)(bz + z) * sZ + (size_t)(by + y) * sY + (bx + x));
        }
    }
    __syncthreads();

    auto fetch = [&](int iz, int iy, int ix) -> float {  // indices guaranteed inside the volume
        if (staged) return tile[((iz - bz) * dy + (iy - by)) * dx + (ix - bx)];
        return load_clean(in + (size_t)iz * sZ + (size_t)iy * sY + ix);
    };

    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy >= p.Yo || ox >= p.Xo) return;
    for (int k = 0; k < ATZ; ++k) {
        const int oz = oz0 + k;
        if (oz >= p.Zo) break;
        double c[3];
        map_point(p.m, (double)(oz + p.cz), (double)(oy + p.cy), (double)(ox + p.cx), c);
        const int dims[3] = {p.Zi, p.Yi, p.Xi};
        bool inside = true;
        if (p.boundary == BH_BOUNDARY_ITK) {
            for (int a = 0; a < 3; ++a) inside = inside && (c[a] >= -0.5) && (c[a] < (double)dims[a] - 0.5);
        } else if (p.boundary == BH_BOUNDARY_SCIPY_CONSTANT) {
            for (int a = 0; a < 3; ++a) inside = inside && (c[a] >= 0.0) && (c[a] <= (double)(dims[a] - 1));
        } else {
            // guard the int conversions below; anything this far out has no in-range neighbour
            for (int a = 0; a < 3; ++a) inside = inside && (c[a] > -2.0) && (c[a] < (double)dims[a] + 1.0);
        }
        float r = p.cval;
        if (inside) {
            if (p.interp == BH_INTERP_NEAREST) {
                int i[3];
                bool ok = true;
                for (int a = 0; a < 3; ++a) {
                    i[a] = (int)floor(c[a] + 0.5);
                    if (p.boundary == BH_BOUNDARY_ITK) i[a] = max(0, min(i[a], dims[a] - 1));
                    ok = ok && i[a] >= 0 && i[a] < dims[a];
                }
                r = ok ? fetch(i[0], i[1], i[2]) : p.cval;
            } else {
                int b[3];
                float f[3];
                for (int a = 0; a < 3; ++a) {
                    const double fl = floor(c[a]);
                    b[a] = (int)fl;
                    f[a] = (float)(c[a] - fl);
                }
                float acc = 0.0f;
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    const int qz = n >> 2, qy = (n >> 1) & 1, qx = n & 1;
                    int iz = b[0] + qz, iy = b[1] + qy, ix = b[2] + qx;
                    const float w = (qz ? f[0] : 1.0f - f[0]) * (qy ? f[1] : 1.0f - f[1]) * (qx ? f[2] : 1.0f - f[2]);
                    const bool ok = iz >= 0 && iz < p.Zi && iy >= 0 && iy < p.Yi && ix >= 0 && ix < p.Xi;
                    float v;
                    if (p.boundary == BH_BOUNDARY_ZEROS) {
                        v = ok ? fetch(iz, iy, ix) : p.cval;
                    } else {  // clamp: weight of an out-of-range neighbour is zero or it repeats the edge
                        iz = max(0, min(iz, p.Zi - 1));
                        iy = max(0, min(iy, p.Yi - 1));
                        ix = max(0, min(ix, p.Xi - 1));
                        v = fetch(iz, iy, ix);
                    }
                    acc += w * v;
                }
                r = acc;
            }
        }
        out[((size_t)oz * p.Yo + oy) * p.Xo + ox] = r;
    }
}

template <typename TIN>
static int launch_affine(bh_ctx* ctx, const TIN* in, float* out, const AffineParams& p) {
    dim3 grid((unsigned)ceil_div(p.Xo, ATX), (unsigned)ceil_div(p.Yo, ATY), (unsigned)ceil_div(p.Zo, ATZ));
    BH_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "affine grid too large");
    hipLaunchKernelGGL(affine_kernel<TIN>, grid, dim3(256), 0, ctx->stream, in, out, p);
    BH_CHECK_HIP(hipGetLastError());
    return BH_OK;
}

}  // namespace bh

extern "C" int bh_affine(bh_ctx* ctx, const void* in, int in_dtype, int64_t Zi, int64_t Yi, int64_t Xi,
                         const double matrix[12], int interpolation, int boundary, float cval, float* out, int64_t Zo,
                         int64_t Yo, int64_t Xo, const int64_t crop_lo[3]) {
    using namespace bh;
    BH_REQUIRE(ctx && in && out && matrix, "NULL argument");
    BH_REQUIRE(Zi > 0 && Yi > 0 && Xi > 0 && Zo > 0 && Yo > 0 && Xo > 0, "invalid shape");
    BH_REQUIRE(Zi < (1ll << 30) && Yi < (1ll << 30) && Xi < (1ll << 30) && Zo < (1ll << 30) && Yo < (1ll << 30) &&
                   Xo < (1ll << 30),
               "volume too large");
    BH_REQUIRE(interpolation == BH_INTERP_NEAREST || interpolation == BH_INTERP_LINEAR, "unknown interpolation %d",
               interpolation);
    BH_REQUIRE(boundary >= BH_BOUNDARY_ITK && boundary <= BH_BOUNDARY_ZEROS, "unknown boundary %d", boundary);
    for (int i = 0; i < 12; ++i) BH_REQUIRE(matrix[i] == matrix[i], "matrix contains NaN");
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    AffineParams p;
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
    p.interp = interpolation;
    p.boundary = boundary;
    p.cval = cval;
    ScopedTimer timer(ctx, T_AFFINE);
    switch (in_dtype) {
        case BH_DT_F32: return launch_affine(ctx, (const float*)in, out, p);
        case BH_DT_U16: return launch_affine(ctx, (const uint16_t*)in, out, p);
        case BH_DT_U8: return launch_affine(ctx, (const uint8_t*)in, out, p);
        case BH_DT_I16: return launch_affine(ctx, (const int16_t*)in, out, p);
        default: BH_REQUIRE(false, "unsupported input dtype code %d", in_dtype);
    }
    return BH_OK;
}
