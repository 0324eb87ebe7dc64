// 3-D affine pull-resample on gfx950 with LDS-tiled trilinear sampling.
//
//   out(p) = in(M p),  p = (z, y, x, 1) in index space, M a 3x4 pull matrix.
//
// Replaces ANTsTransform.apply_to_image (biahub/register.py:261-269, stabilize.py:82-88: ITK
// ResampleImageFilter, origin 0 / spacing 1 / identity direction because both images come from
// ants.from_numpy) and scipy.ndimage.affine_transform (core/transform.py:384-396).  NaN -> 0
// (register.py:254) is folded into the load.
//
// A workgroup owns a 8 x 8 x 64 (z, y, x) output tile.  It maps the tile's corners through M,
// stages the source bounding box (clipped to the volume) into LDS with row-contiguous reads and
// samples from LDS; when the box does not fit (strong scale/rotation) it gathers from global
// memory through L2 instead.  Coordinates are float64 (ITK and SciPy both use doubles), weights
// and accumulation float32.
#include "common.hpp"

#include <cmath>

namespace bh {

constexpr int ATX = 64, ATY = 8, ATZ = 8;
constexpr int A_LDS_FLOATS = 9216;   // 36 KiB -> four workgroups per CU

struct AffineParams {
    double m[12];
    long long mq[12];  // the same matrix in Q32.32 fixed point (linear interior fast path)
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

// per-axis interpolation plan: clamped neighbour indices and their weights
template <int BOUNDARY>
__device__ __forceinline__ void axis_plan(double c, int n, int& j0, int& j1, float& w0, float& w1) {
    const double fl = floor(c);
    const int i0 = (int)fl;
    const float f = (float)(c - fl);
    w0 = 1.0f - f;
    w1 = f;
    if (BOUNDARY == BH_BOUNDARY_ZEROS) {  // out-of-range neighbours contribute cval, not data
        if (i0 < 0 || i0 >= n) w0 = 0.0f;
        if (i0 + 1 < 0 || i0 + 1 >= n) w1 = 0.0f;
    }
    j0 = max(0, min(i0, n - 1));
    j1 = max(0, min(i0 + 1, n - 1));
}

// INTERP / BOUNDARY are compile-time so the sampling loop carries no mode branches.
template <typename TIN, int INTERP, int BOUNDARY>
__global__ __launch_bounds__(256) void affine_kernel(const TIN* __restrict__ in, float* __restrict__ out,
                                                     AffineParams p) {
    __shared__ float tile[A_LDS_FLOATS];
    __shared__ int box[9];  // origin[3], extent[3], interior flag per axis[3]
    const int tx = threadIdx.x & 63;
    const int ty = threadIdx.x >> 6;
    const int ox0 = blockIdx.x * ATX, oy0 = blockIdx.y * ATY, oz0 = blockIdx.z * ATZ;

    // source bounding box of this tile (uniform across the block): an affine map of a box is bounded per axis by
    // base + sum of the negative / positive edge extents; threads 0..2 take one source axis each.  A relative
    // 1e-9 slack absorbs the rounding difference to the per-voxel evaluation below.
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        const int z1 = min(oz0 + ATZ, p.Zo) - 1, y1 = min(oy0 + ATY, p.Yo) - 1, x1 = min(ox0 + ATX, p.Xo) - 1;
        const double base = p.m[4 * a] * (double)(oz0 + p.cz) + p.m[4 * a + 1] * (double)(oy0 + p.cy) +
                            p.m[4 * a + 2] * (double)(ox0 + p.cx) + p.m[4 * a + 3];
        const double ez = p.m[4 * a] * (double)(z1 - oz0), ey = p.m[4 * a + 1] * (double)(y1 - oy0),
                     ex = p.m[4 * a + 2] * (double)(x1 - ox0);
        double lo = base + fmin(ez, 0.0) + fmin(ey, 0.0) + fmin(ex, 0.0);
        double hi = base + fmax(ez, 0.0) + fmax(ey, 0.0) + fmax(ex, 0.0);
        const double slack = 1e-9 * (fabs(lo) + fabs(hi) + 1.0);
        lo -= slack;
        hi += slack;
        const int n = a == 0 ? p.Zi : (a == 1 ? p.Yi : p.Xi);
        // nearest needs floor(c+0.5); linear needs floor(c) and floor(c)+1: [floor(lo), floor(hi)+1] covers both
        double l = fmax(floor(lo), 0.0), h = fmin(floor(hi) + 1.0, (double)(n - 1));
        box[a] = (int)l;
        box[3 + a] = (h >= l) ? (int)(h - l) + 1 : 0;
        // interior: every floor(c) and floor(c)+1 of this tile is a valid index on this axis (no clamp, no test)
        box[6 + a] = (floor(lo) >= 0.0 && floor(hi) + 1.0 <= (double)(n - 1)) ? 1 : 0;
    }
    __syncthreads();
    const int bz = box[0], by = box[1], bx = box[2];
    const int dz = box[3], dy = box[4], dx = box[5];
    const int64_t nbox = (int64_t)dz * dy * dx;
    const int ox = ox0 + tx;
    if (nbox == 0) {  // no source voxel can contribute to this tile
        for (int yy = ty; yy < ATY; yy += 4) {
            const int oy = oy0 + yy;
            if (oy < p.Yo && ox < p.Xo)
                for (int k = 0; k < ATZ && oz0 + k < p.Zo; ++k)
                    out[((size_t)(oz0 + k) * p.Yo + oy) * p.Xo + ox] = p.cval;
        }
        return;
    }
    const bool staged = nbox <= A_LDS_FLOATS;
    const size_t sY = (size_t)p.Xi, sZ = (size_t)p.Yi * p.Xi;
    if (staged) {
        // rows of the box are contiguous in x: one wave per row, lanes along x (coalesced)
        const int nrows = dz * dy;
        if (sizeof(TIN) == 4) {
            // float32: LDS-DMA, every row of this wave in flight at once, no VGPR staging
            const int wv = __builtin_amdgcn_readfirstlane(ty);
            int z = wv / dy, y = wv - z * dy;  // row r = z * dy + y, advanced by 4 rows per step
            const int qz = 4 / dy, qy = 4 - qz * dy;
            for (int r = wv; r < nrows; r += 4) {
                const TIN* rowp = in + (size_t)(bz + z) * sZ + (size_t)(by + y) * sY + bx;
                for (int x0 = 0; x0 < dx; x0 += 64) {
                    if (x0 + tx < dx) {
                        const TIN* src = rowp + x0 + tx;
                        const unsigned lds_dst = (unsigned)(size_t)(tile + r * dx + x0);
                        unsigned keep;
                        asm volatile(
                            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                            : "=&s"(keep)
                            : "v"(src), "s"(lds_dst)
                            : "memory");
                    }
                }
                z += qz;
                y += qy;
                if (y >= dy) {
                    y -= dy;
                    ++z;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // np.nan_to_num on the staged copy
            for (int i = threadIdx.x; i < (int)nbox; i += 256) tile[i] = load_clean(tile + i);
        } else {
            constexpr int U = 8;
            for (int rb = ty; rb < nrows; rb += 4 * U) {
                for (int x0 = 0; x0 < dx; x0 += 64) {
                    const int x = min(x0 + tx, dx - 1);
                    float v[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int r = min(rb + 4 * u, nrows - 1);
                        const int z = r / dy, y = r - z * dy;
                        v[u] = load_clean(in + (size_t)(bz + z) * sZ + (size_t)(by + y) * sY + bx + x);
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int r = rb + 4 * u;
                        if (r < nrows && x0 + tx < dx) tile[r * dx + x0 + tx] = v[u];
                    }
                }
            }
        }
    }
    __syncthreads();
    if (ox >= p.Xo) return;

    // strides / origin of whichever copy of the source we sample from
    const int fsz = staged ? dy * dx : 0, fsy = staged ? dx : 0;
    auto fetch = [&](int iz, int iy, int ix) -> float {  // indices inside the volume (and inside the box when staged)
        if (staged) return tile[(iz - bz) * fsz + (iy - by) * fsy + (ix - bx)];
        return load_clean(in + (size_t)iz * sZ + (size_t)iy * sY + ix);
    };
    const int dims[3] = {p.Zi, p.Yi, p.Xi};
    if (INTERP == BH_INTERP_LINEAR && BOUNDARY != BH_BOUNDARY_ZEROS) {
        // Linear with edge clamp (ITK / SciPy rules).  Coordinates in Q32.32 fixed point: the integer part is
        // floor(c) for free and the next z is one 64-bit add per axis; the eight taps are combined as three nested
        // lerps.  Tiles whose source box is strictly inside the volume (the bulk of a registration warp) skip all
        // bounds handling and read the taps as paired LDS loads; the arithmetic is identical in both branches, so a
        // voxel's value does not depend on which tile (or crop) computed it.
        const bool interior = staged && box[6] && box[7] && box[8];
        const int sxy = dy * dx;
        for (int yy = ty; yy < ATY; yy += 4) {
            const int oy = oy0 + yy;
            if (oy >= p.Yo) break;
            long long c0[3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
                c0[a] = p.mq[4 * a] * (long long)(oz0 + p.cz) + p.mq[4 * a + 1] * (long long)(oy + p.cy) +
                        p.mq[4 * a + 2] * (long long)(ox + p.cx) + p.mq[4 * a + 3];
            float* o = out + ((size_t)oz0 * p.Yo + oy) * p.Xo + ox;
            const size_t ostep = (size_t)p.Yo * p.Xo;
            for (int k = 0; k < ATZ && oz0 + k < p.Zo; ++k) {
                const int iz = (int)(c0[0] >> 32), iy = (int)(c0[1] >> 32), ix = (int)(c0[2] >> 32);
                const float fz = (float)(unsigned)(c0[0] & 0xffffffffll) * 2.3283064365386963e-10f;
                const float fy = (float)(unsigned)(c0[1] & 0xffffffffll) * 2.3283064365386963e-10f;
                const float fx = (float)(unsigned)(c0[2] & 0xffffffffll) * 2.3283064365386963e-10f;
                float v000, v001, v010, v011, v100, v101, v110, v111;
                bool inside = true;
                if (interior) {
                    const float* t0 = tile + ((iz - bz) * sxy + (iy - by) * dx + (ix - bx));
                    const float* t1 = t0 + sxy;
                    v000 = t0[0], v001 = t0[1], v010 = t0[dx], v011 = t0[dx + 1];
                    v100 = t1[0], v101 = t1[1], v110 = t1[dx], v111 = t1[dx + 1];
                } else {
                    // The inside/outside decision at the volume faces uses the float64 coordinate in numpy / ITK
                    // association (ties such as c == -0.5 exactly must fall like the reference's); only boundary
                    // tiles pay for it.
                    {
#pragma clang fp contract(off)
                        const double zd = (double)(oz0 + k + p.cz), yd = (double)(oy + p.cy), xd = (double)(ox + p.cx);
#pragma unroll
                        for (int a = 0; a < 3; ++a) {
                            const double ca = p.m[4 * a] * zd + p.m[4 * a + 1] * yd + p.m[4 * a + 2] * xd + p.m[4 * a + 3];
                            if (BOUNDARY == BH_BOUNDARY_ITK)
                                inside = inside && ca >= -0.5 && ca < (double)dims[a] - 0.5;
                            else
                                inside = inside && ca >= 0.0 && ca <= (double)(dims[a] - 1);
                        }
                    }
                    if (inside) {
                        const int z0 = max(0, min(iz, p.Zi - 1)), z1 = max(0, min(iz + 1, p.Zi - 1));
                        const int y0 = max(0, min(iy, p.Yi - 1)), y1 = max(0, min(iy + 1, p.Yi - 1));
                        const int x0 = max(0, min(ix, p.Xi - 1)), x1 = max(0, min(ix + 1, p.Xi - 1));
                        v000 = fetch(z0, y0, x0), v001 = fetch(z0, y0, x1), v010 = fetch(z0, y1, x0), v011 = fetch(z0, y1, x1);
                        v100 = fetch(z1, y0, x0), v101 = fetch(z1, y0, x1), v110 = fetch(z1, y1, x0), v111 = fetch(z1, y1, x1);
                    } else {
                        v000 = v001 = v010 = v011 = v100 = v101 = v110 = v111 = 0.0f;
                    }
                }
                const float a00 = v000 + fx * (v001 - v000);
                const float a01 = v010 + fx * (v011 - v010);
                const float a10 = v100 + fx * (v101 - v100);
                const float a11 = v110 + fx * (v111 - v110);
                const float b0 = a00 + fy * (a01 - a00);
                const float b1 = a10 + fy * (a11 - a10);
                o[k * ostep] = inside ? b0 + fz * (b1 - b0) : p.cval;
#pragma unroll
                for (int a = 0; a < 3; ++a) c0[a] += p.mq[4 * a];
            }
        }
        return;
    }
    for (int yy = ty; yy < ATY; yy += 4) {
        const int oy = oy0 + yy;
        if (oy >= p.Yo) break;
        // numpy / ITK association: ((m0*z + m1*y) + m2*x) + m3 — the y and x products are per-row constants
        double py[3], px[3];
        {
#pragma clang fp contract(off)
            const double yd = (double)(oy + p.cy), xd = (double)(ox + p.cx);
            for (int a = 0; a < 3; ++a) {
                py[a] = p.m[4 * a + 1] * yd;
                px[a] = p.m[4 * a + 2] * xd;
            }
        }
        for (int k = 0; k < ATZ; ++k) {
            const int oz = oz0 + k;
            if (oz >= p.Zo) break;
            double c[3];
            {
#pragma clang fp contract(off)
                const double zd = (double)(oz + p.cz);
                for (int a = 0; a < 3; ++a) c[a] = ((p.m[4 * a] * zd + py[a]) + px[a]) + p.m[4 * a + 3];
            }
            bool inside = true;
            if (BOUNDARY == BH_BOUNDARY_ITK) {
                for (int a = 0; a < 3; ++a) inside = inside && (c[a] >= -0.5) && (c[a] < (double)dims[a] - 0.5);
            } else if (BOUNDARY == BH_BOUNDARY_SCIPY_CONSTANT) {
                for (int a = 0; a < 3; ++a) inside = inside && (c[a] >= 0.0) && (c[a] <= (double)(dims[a] - 1));
            } else {  // guard the int conversions; anything this far out has no in-range neighbour
                for (int a = 0; a < 3; ++a) inside = inside && (c[a] > -2.0) && (c[a] < (double)dims[a] + 1.0);
            }
            float r = p.cval;
            if (inside) {
                if (INTERP == BH_INTERP_NEAREST) {
                    int i[3];
                    bool ok = true;
                    for (int a = 0; a < 3; ++a) {
                        i[a] = (int)floor(c[a] + 0.5);
                        if (BOUNDARY == BH_BOUNDARY_ITK) i[a] = max(0, min(i[a], dims[a] - 1));
                        ok = ok && i[a] >= 0 && i[a] < dims[a];
                    }
                    if (ok) r = fetch(i[0], i[1], i[2]);
                } else {
                    int z0, z1, y0, y1, x0, x1;
                    float wz0, wz1, wy0, wy1, wx0, wx1;
                    axis_plan<BOUNDARY>(c[0], p.Zi, z0, z1, wz0, wz1);
                    axis_plan<BOUNDARY>(c[1], p.Yi, y0, y1, wy0, wy1);
                    axis_plan<BOUNDARY>(c[2], p.Xi, x0, x1, wx0, wx1);
                    // same accumulation order as the oracle: dz outer, dy, dx inner
                    float acc = 0.0f;
                    acc += (wz0 * wy0 * wx0) * fetch(z0, y0, x0);
                    acc += (wz0 * wy0 * wx1) * fetch(z0, y0, x1);
                    acc += (wz0 * wy1 * wx0) * fetch(z0, y1, x0);
                    acc += (wz0 * wy1 * wx1) * fetch(z0, y1, x1);
                    acc += (wz1 * wy0 * wx0) * fetch(z1, y0, x0);
                    acc += (wz1 * wy0 * wx1) * fetch(z1, y0, x1);
                    acc += (wz1 * wy1 * wx0) * fetch(z1, y1, x0);
                    acc += (wz1 * wy1 * wx1) * fetch(z1, y1, x1);
                    if (BOUNDARY == BH_BOUNDARY_ZEROS) {
                        const float cover = (wz0 + wz1) * (wy0 + wy1) * (wx0 + wx1);
                        acc += (1.0f - cover) * p.cval;
                    }
                    r = acc;
                }
            }
            out[((size_t)oz * p.Yo + oy) * p.Xo + ox] = r;
        }
    }
}

template <typename TIN>
static int launch_affine(bh_ctx* ctx, const TIN* in, float* out, const AffineParams& p) {
    dim3 grid((unsigned)ceil_div(p.Xo, ATX), (unsigned)ceil_div(p.Yo, ATY), (unsigned)ceil_div(p.Zo, ATZ));
    BH_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "affine grid too large");
    auto run = [&](auto kern) -> int {
        hipLaunchKernelGGL(kern, grid, dim3(256), 0, ctx->stream, in, out, p);
        BH_CHECK_HIP(hipGetLastError());
        return BH_OK;
    };
#define BH_AFF(I, B) return run(affine_kernel<TIN, I, B>)
    if (p.interp == BH_INTERP_NEAREST) {
        if (p.boundary == BH_BOUNDARY_ITK) BH_AFF(BH_INTERP_NEAREST, BH_BOUNDARY_ITK);
        if (p.boundary == BH_BOUNDARY_SCIPY_CONSTANT) BH_AFF(BH_INTERP_NEAREST, BH_BOUNDARY_SCIPY_CONSTANT);
        BH_AFF(BH_INTERP_NEAREST, BH_BOUNDARY_ZEROS);
    }
    if (p.boundary == BH_BOUNDARY_ITK) BH_AFF(BH_INTERP_LINEAR, BH_BOUNDARY_ITK);
    if (p.boundary == BH_BOUNDARY_SCIPY_CONSTANT) BH_AFF(BH_INTERP_LINEAR, BH_BOUNDARY_SCIPY_CONSTANT);
    BH_AFF(BH_INTERP_LINEAR, BH_BOUNDARY_ZEROS);
#undef BH_AFF
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
    for (int i = 0; i < 12; ++i) {
        p.m[i] = matrix[i];
        BH_REQUIRE(std::fabs(matrix[i]) < 1073741824.0, "matrix entry %d out of range", i);
        p.mq[i] = std::llround(matrix[i] * 4294967296.0);
    }
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
