// 3-D affine pull-resample on gfx950 with LDS-tiled trilinear sampling.
//
//   out(p) = in(M p),  p = (z, y, x, 1) in index space, M a 3x4 pull matrix.
//
// Replaces ANTsTransform.apply_to_image (biahub/register.py:261-269, stabilize.py:82-88: ITK
// ResampleImageFilter, origin 0 / spacing 1 / identity direction because both images come from
// ants.from_numpy) and scipy.ndimage.affine_transform (core/transform.py:384-396).  NaN -> 0
// (register.py:254) is folded into the load.
//
// A workgroup owns a 8 x 8 x 64 (z, y, x) output tile.  It bounds the tile's source box analytically, stages the
// box (clipped to the volume) into LDS by LDS-DMA — 16 B per lane over the box as a flat list of quads when the rows
// are 16-B aligned, 1 KiB per instruction — and samples from LDS; when the box does not fit (strong scale / rotation)
// it gathers from global memory through L2 instead.  The launch asks for exactly the LDS the matrix can need, so gentle
// warps run more workgroups per CU.  Linear interpolation with edge clamp runs in Q32.32 fixed-point coordinates with
// packed-fp32 lerps (31 VALU instructions per voxel in the interior loop) and handles np.nan_to_num lazily: a
// non-finite result means a non-finite tap, and only then are the taps cleaned and the voxel redone.  The general
// path (nearest, ZEROS boundary, volume faces) keeps float64 coordinates like ITK and SciPy.
// MI355X, 512 x 2048 x 2048 f32, 2 deg / 1.02 similarity: 4.6 ms (8 B/voxel algorithmic = 3.7 TB/s).
#include "common.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace bh {

constexpr int ATX = 64, ATY = 8, ATZ = 8;
constexpr int A_LDS_FLOATS = 9984;      // 39 KiB cap -> at least four workgroups per CU; gentler warps get less (host bound)
constexpr int A_LDS_FLOATS_X8 = 13056;  // 16-bit input staged in groups of 8: wider boxes, 51 KiB cap -> three per CU

struct AffineParams {
    double m[12];
    long long mq[12];  // the same matrix in Q32.32 fixed point (linear interior fast path)
    int Zi, Yi, Xi;
    int Zo, Yo, Xo;
    int cz, cy, cx;
    int interp, boundary;
    float cval;
    int lds_floats;  // staging capacity of this launch (dynamic LDS), <= A_LDS_FLOATS[_X8]
    int x4;          // float32 rows are 16-B aligned: stage with 16-B LDS-DMA, box x range rounded out to 4 floats
    int x8;          // 16-bit rows are 16-B aligned: stage 8 samples per lane (one 16-B load), x range rounded out to 8
    int zslot;       // z walk: bytes of one plane slot of a wave's LDS ring (0: planes travel through registers)
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

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 clean2(f2 v) {
    const float a = v.x, b = v.y;
    return f2{load_clean(&a), load_clean(&b)};
}

// Trilinear blend of the taps P0 = (z0,y0,x0|x1), P1 = (z0,y1,..), P2 = (z1,y0,..), P3 = (z1,y1,..) with the Q0.32
// fractions qz, qy, qx: y and z lerps on x-pairs (packed fp32 FMAs), x last.  Explicit fma so that every call site
// rounds identically.
__device__ __forceinline__ float lerp8f(f2 P0, f2 P1, f2 P2, f2 P3, float fzf, float fyf, float fx) {
    const f2 fy = {fyf, fyf}, fz = {fzf, fzf};
    const f2 A0 = __builtin_elementwise_fma(fy, P1 - P0, P0);
    const f2 A1 = __builtin_elementwise_fma(fy, P3 - P2, P2);
    const f2 B = __builtin_elementwise_fma(fz, A1 - A0, A0);
    return __builtin_fmaf(fx, B.y - B.x, B.x);
}
__device__ __forceinline__ float lerp8(f2 P0, f2 P1, f2 P2, f2 P3, unsigned qz, unsigned qy, unsigned qx) {
    const f2 fzy = f2{(float)qz, (float)qy} * 2.3283064365386963e-10f;
    return lerp8f(P0, P1, P2, P3, fzy.x, fzy.y, (float)qx * 2.3283064365386963e-10f);
}

constexpr int A_NW = 4, A_NT = 64 * A_NW;  // waves / threads per workgroup

struct TileBox {  // per-tile source box, written by threads 0..2, read by everyone after a barrier
    int org[3], ext[3], interior[3];
    unsigned rcp_l, rcp_dy;  // ceil(2^32 / (ext_x / 4)), ceil(2^32 / ext_y): exact small divisions in stage_box
    int ox0, oy0, oz0;
};

// Source bounding box of an output tile: an affine map of a box is bounded per axis by base + the sum of the
// negative / positive edge extents; threads 0..2 take one source axis each.  A relative 1e-9 slack absorbs the
// rounding difference to the per-voxel evaluation.
__device__ __forceinline__ void compute_box(const AffineParams& p, int tile, int ntx, int nty, TileBox* b) {
    const int tzi = tile / (ntx * nty), rem = tile - tzi * (ntx * nty), tyi = rem / ntx, txi = rem - tyi * ntx;
    const int ox0 = txi * ATX, oy0 = tyi * ATY, oz0 = tzi * ATZ;
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
        const double l = fmax(floor(lo), 0.0), h = fmin(floor(hi) + 1.0, (double)(n - 1));
        int org = (int)l, ext = (h >= l) ? (int)(h - l) + 1 : 0;
        if (a == 2 && p.x4 && ext > 0) {  // whole 16-B quads; Xi % 4 == 0, so this stays inside the row
            const int end = (org + ext + 3) & ~3;
            org &= ~3;
            ext = end - org;
            b->rcp_l = (unsigned)(0xffffffffu / (unsigned)(ext >> 2)) + 1u;  // wraps to 0 for 1: handled by the user
        }
        if (a == 2 && p.x8 && ext > 0) {  // 16-bit input: whole 16-B groups of 8 samples; Xi % 8 == 0
            const int end = (org + ext + 7) & ~7;
            org &= ~7;
            ext = end - org;
            b->rcp_l = (unsigned)(0xffffffffu / (unsigned)(ext >> 3)) + 1u;
        }
        if (a == 1 && ext > 0) b->rcp_dy = (unsigned)(0xffffffffu / (unsigned)ext) + 1u;
        b->org[a] = org;
        b->ext[a] = ext;
        // interior: every floor(c) and floor(c)+1 of this tile is a valid index on this axis (no clamp, no test)
        b->interior[a] = (floor(lo) >= 0.0 && floor(hi) + 1.0 <= (double)(n - 1)) ? 1 : 0;
    }
    if (threadIdx.x == 3) {
        b->ox0 = ox0;
        b->oy0 = oy0;
        b->oz0 = oz0;
    }
}

// Start staging the tile's source box into `tile` (rows contiguous in x: one wave per row, lanes along x).
// float32: LDS-DMA, every row of a wave in flight at once, nothing staged in VGPRs, returns immediately — the
// caller waits (vmcnt(0) + barrier) before the first read.  Other dtypes: batched register loads (synchronous).
template <typename TIN>
__device__ __forceinline__ void stage_box(const TIN* __restrict__ in, const AffineParams& p, const TileBox& b,
                                          float* tile) {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int bz = b.org[0], by = b.org[1], bx = b.org[2];
    const int dz = b.ext[0], dy = b.ext[1], dx = b.ext[2];
    const int64_t nbox = (int64_t)dz * dy * dx;
    if (nbox == 0 || nbox > p.lds_floats) return;
    const size_t sY = (size_t)p.Xi, sZ = (size_t)p.Yi * p.Xi;
    const int nrows = dz * dy;
    if (sizeof(TIN) == 4 && p.x4) {
        // 16-B LDS-DMA over the box as a flat list of quads (row pitch = dx exactly, so the lane-linear LDS image IS
        // the box): 64 quads = 1 KiB per instruction whatever the row length, each lane fetching from its own row.
        const unsigned L = (unsigned)dx >> 2, S = (unsigned)nrows * L;
        const unsigned wv = __builtin_amdgcn_readfirstlane(ty);
        const TIN* base = in + (size_t)bz * sZ + (size_t)by * sY + bx;
        for (unsigned i = wv * 64; i < S; i += A_NT) {
            const unsigned q = i + tx;
            if (q < S) {
                // q / L and r / dy by multiply-high with ceil(2^32 / d): exact while q * d < 2^32 (q < 10^4 here)
                const unsigned r = L == 1 ? q : __umulhi(q, b.rcp_l), xq = q - r * L;
                const unsigned z = dy == 1 ? r : __umulhi(r, b.rcp_dy), y = r - z * (unsigned)dy;
                const TIN* src = base + ((size_t)z * sZ + (size_t)y * sY + 4 * xq);
                const unsigned lds_dst = (unsigned)(size_t)(tile + 4 * i);
                unsigned keep;
                asm volatile(
                    "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                    : "=&s"(keep)
                    : "v"(src), "s"(lds_dst)
                    : "memory");
            }
        }
    } else if (sizeof(TIN) == 4) {
        const int wv = __builtin_amdgcn_readfirstlane(ty);
        int z = wv / dy, y = wv - z * dy;  // row r = z * dy + y, advanced by A_NW rows per step
        const int qz = A_NW / dy, qy = A_NW - qz * dy;
        for (int r = wv; r < nrows; r += A_NW) {
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
    } else if (sizeof(TIN) == 2 && p.x8) {
        // 16-bit input: the box as a flat list of 8-sample groups, one 16-B load per lane, widened in registers
        const unsigned L = (unsigned)dx >> 3, S = (unsigned)nrows * L;
        const TIN* base = in + (size_t)bz * sZ + (size_t)by * sY + bx;
        for (unsigned i0 = 0; i0 < S; i0 += A_NT * 2) {
            uint4 raw[2];
            unsigned qq[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const unsigned q = min(i0 + u * A_NT + threadIdx.x, S - 1);
                const unsigned r = L == 1 ? q : __umulhi(q, b.rcp_l), xq = q - r * L;
                const unsigned z = dy == 1 ? r : __umulhi(r, b.rcp_dy), y = r - z * (unsigned)dy;
                raw[u] = *reinterpret_cast<const uint4*>(base + ((size_t)z * sZ + (size_t)y * sY + 8 * xq));
                qq[u] = q;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (i0 + u * A_NT + threadIdx.x < S) {
                    const unsigned w[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
                    float f[8];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        f[2 * k] = (float)(TIN)(w[k] & 0xffffu);
                        f[2 * k + 1] = (float)(TIN)(w[k] >> 16);
                    }
                    float4* dst = reinterpret_cast<float4*>(tile + 8 * qq[u]);
                    dst[0] = make_float4(f[0], f[1], f[2], f[3]);
                    dst[1] = make_float4(f[4], f[5], f[6], f[7]);
                }
            }
        }
    } else {
        constexpr int U = 8;
        for (int rb = ty; rb < nrows; rb += A_NW * U) {
            for (int x0 = 0; x0 < dx; x0 += 64) {
                const int x = min(x0 + tx, dx - 1);
                float v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int r = min(rb + A_NW * u, nrows - 1);
                    const int z = r / dy, y = r - z * dy;
                    v[u] = load_clean(in + (size_t)(bz + z) * sZ + (size_t)(by + y) * sY + bx + x);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int r = rb + A_NW * u;
                    if (r < nrows && x0 + tx < dx) tile[r * dx + x0 + tx] = v[u];
                }
            }
        }
    }
}

// Sample one 8 x 8 x 64 output tile from its staged box (or from global memory when the box did not fit).
template <typename TIN, int INTERP, int BOUNDARY>
__device__ __forceinline__ void sample_tile(const TIN* __restrict__ in, float* __restrict__ out, const AffineParams& p,
                                            const TileBox& b, const float* tile) {
    const int tx = threadIdx.x & 63, ty = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ox0 = b.ox0, oy0 = b.oy0, oz0 = b.oz0;
    const int bz = b.org[0], by = b.org[1], bx = b.org[2];
    const int dz = b.ext[0], dy = b.ext[1], dx = b.ext[2];
    const int64_t nbox = (int64_t)dz * dy * dx;
    const int ox = ox0 + tx;
    if (ox >= p.Xo) return;
    if (nbox == 0) {  // no source voxel can contribute to this tile
        for (int yy = ty; yy < ATY; yy += A_NW) {
            const int oy = oy0 + yy;
            if (oy < p.Yo)
                for (int k = 0; k < ATZ && oz0 + k < p.Zo; ++k) out[((size_t)(oz0 + k) * p.Yo + oy) * p.Xo + ox] = p.cval;
        }
        return;
    }
    const bool staged = nbox <= p.lds_floats;
    const size_t sY = (size_t)p.Xi, sZ = (size_t)p.Yi * p.Xi;
    const int fsz = staged ? dy * dx : 0, fsy = staged ? dx : 0;
    auto fetch = [&](int iz, int iy, int ix) -> float {  // indices inside the volume (and inside the box when staged)
        if (staged) return load_clean(tile + ((iz - bz) * fsz + (iy - by) * fsy + (ix - bx)));  // np.nan_to_num per tap
        return load_clean(in + (size_t)iz * sZ + (size_t)iy * sY + ix);
    };
    const int dims[3] = {p.Zi, p.Yi, p.Xi};
    if (INTERP == BH_INTERP_LINEAR && BOUNDARY != BH_BOUNDARY_ZEROS) {
        // Linear with edge clamp (ITK / SciPy rules).  Coordinates in Q32.32 fixed point: the integer part is
        // floor(c) for free and the next z is one 64-bit add per axis; the eight taps are combined by lerp8.  Tiles
        // whose source box is strictly inside the volume (the bulk of a registration warp) take a branch-free loop
        // with paired LDS reads; the arithmetic is identical in both branches, so a voxel's value does not depend on
        // which tile (or crop) computed it.
        const bool interior = staged && b.interior[0] && b.interior[1] && b.interior[2];
        const size_t ostep = (size_t)p.Yo * p.Xo;
        const int nk = min(ATZ, p.Zo - oz0);
        if (interior) {
            const int dx4 = dx * 4, sxy4 = dy * dx * 4;
            for (int yy = ty; yy < ATY; yy += A_NW) {
                const int oy = oy0 + yy;
                if (oy >= p.Yo) break;
                long long c0[3];  // relative to the box origin: the high words index the staged copy directly
#pragma unroll
                for (int a = 0; a < 3; ++a)
                    c0[a] = p.mq[4 * a] * (long long)(oz0 + p.cz) + p.mq[4 * a + 1] * (long long)(oy + p.cy) +
                            p.mq[4 * a + 2] * (long long)(ox + p.cx) + p.mq[4 * a + 3] - ((long long)b.org[a] << 32);
                float* o = out + ((size_t)oz0 * p.Yo + oy) * p.Xo;
                const char* tb = (const char*)tile;
                auto one = [&](int k) {
                    const int iz = (int)(c0[0] >> 32), iy = (int)(c0[1] >> 32), ix = (int)(c0[2] >> 32);
                    const char* t00 = tb + ((iz * dy + iy) * dx4 + (ix << 2));
                    const char* t01 = t00 + dx4;
                    const char* t10 = t00 + sxy4;
                    const char* t11 = t10 + dx4;
                    const f2 P0 = {((const float*)t00)[0], ((const float*)t00)[1]};
                    const f2 P1 = {((const float*)t01)[0], ((const float*)t01)[1]};
                    const f2 P2 = {((const float*)t10)[0], ((const float*)t10)[1]};
                    const f2 P3 = {((const float*)t11)[0], ((const float*)t11)[1]};
                    float r = lerp8(P0, P1, P2, P3, (unsigned)c0[0], (unsigned)c0[1], (unsigned)c0[2]);
                    if (__builtin_expect(!__builtin_isfinite(r), 0))  // a NaN / inf tap: redo on np.nan_to_num'd taps
                        r = lerp8(clean2(P0), clean2(P1), clean2(P2), clean2(P3), (unsigned)c0[0], (unsigned)c0[1],
                                  (unsigned)c0[2]);
                    o[k * ostep + ox] = r;
#pragma unroll
                    for (int a = 0; a < 3; ++a) c0[a] += p.mq[4 * a];
                };
                if (nk == ATZ) {
#pragma unroll
                    for (int k = 0; k < ATZ; ++k) one(k);
                } else {
                    for (int k = 0; k < nk; ++k) one(k);
                }
            }
            return;
        }
        for (int yy = ty; yy < ATY; yy += A_NW) {
            const int oy = oy0 + yy;
            if (oy >= p.Yo) break;
            long long c0[3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
                c0[a] = p.mq[4 * a] * (long long)(oz0 + p.cz) + p.mq[4 * a + 1] * (long long)(oy + p.cy) +
                        p.mq[4 * a + 2] * (long long)(ox + p.cx) + p.mq[4 * a + 3];
            float* o = out + ((size_t)oz0 * p.Yo + oy) * p.Xo + ox;
            for (int k = 0; k < nk; ++k) {
                const int iz = (int)(c0[0] >> 32), iy = (int)(c0[1] >> 32), ix = (int)(c0[2] >> 32);
                // The inside/outside decision at the volume faces uses the float64 coordinate in numpy / ITK
                // association (ties such as c == -0.5 exactly must fall like the reference's); only boundary
                // tiles pay for it.
                bool inside = true;
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
                float r = p.cval;
                if (inside) {
                    const int z0 = max(0, min(iz, p.Zi - 1)), z1 = max(0, min(iz + 1, p.Zi - 1));
                    const int y0 = max(0, min(iy, p.Yi - 1)), y1 = max(0, min(iy + 1, p.Yi - 1));
                    const int x0 = max(0, min(ix, p.Xi - 1)), x1 = max(0, min(ix + 1, p.Xi - 1));
                    const f2 P0 = {fetch(z0, y0, x0), fetch(z0, y0, x1)};
                    const f2 P1 = {fetch(z0, y1, x0), fetch(z0, y1, x1)};
                    const f2 P2 = {fetch(z1, y0, x0), fetch(z1, y0, x1)};
                    const f2 P3 = {fetch(z1, y1, x0), fetch(z1, y1, x1)};
                    r = lerp8(P0, P1, P2, P3, (unsigned)c0[0], (unsigned)c0[1], (unsigned)c0[2]);
                }
                o[k * ostep] = r;
#pragma unroll
                for (int a = 0; a < 3; ++a) c0[a] += p.mq[4 * a];
            }
        }
        return;
    }
    for (int yy = ty; yy < ATY; yy += A_NW) {
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

// One output tile per workgroup; four workgroups fit a CU's LDS, so the staging of one overlaps the sampling of the
// others without in-kernel double buffering (a persistent double-buffered variant measured 1.8x slower: coarser
// barriers at the same occupancy).  Workgroups are dealt round-robin to the 8 XCDs, so block b takes tile
// (b % 8) * ntiles/8 + b / 8: each XCD walks its own contiguous run of tiles (x fastest) and x / y neighbours share
// their halo lines in that XCD's L2.
// INTERP / BOUNDARY are compile-time so the sampling loop carries no mode branches.
template <typename TIN, int INTERP, int BOUNDARY>
__global__ __launch_bounds__(A_NT) void affine_kernel(const TIN* __restrict__ in, float* __restrict__ out,
                                                     AffineParams p, int ntx, int nty, int ntiles, int per_xcd) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    __shared__ TileBox box;
    const int t = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || t >= ntiles) return;
    compute_box(p, t, ntx, nty, &box);
    __syncthreads();
    stage_box(in, p, box, tile);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the LDS-DMA is not counted by the compiler
    __syncthreads();
    sample_tile<TIN, INTERP, BOUNDARY>(in, out, p, box, tile);
}

// Strongly coupled warps (a rotation of more than a few degrees that mixes z with x or y): the source box of a 64 x 8 x 8
// output tile no longer fits LDS, and affine_kernel's fallback — every lane of a 64-voxel x row gathering its eight taps from
// wherever the map sends it — touches up to 64 planes per wave instruction (20 degrees about y: 33 ms, 45 degrees about an
// oblique axis: 94 ms for an 8.6-GB volume; tools/affine_angle_sweep.py).  Here a workgroup owns a COMPACT block of the output,
// 8 (z) x 8 (y) x 16 (x) voxels, a wavefront 16 x-voxels of 4 rows of one plane and then the next three planes: the source
// footprint of a workgroup is a small rotated box whatever the matrix — a few thousand voxels, staged in LDS once per block
// (through load_clean), so every tap is an LDS read; stores are 64-byte segments.
// The per-voxel arithmetic is sample_tile's, branch for branch (Q32.32 + lerp8 for linear with an edge clamp, the generic float64
// path otherwise): results are bit-identical to the tile kernel's.
// Block geometry G: GX x GY lanes of a plane, GZL plane groups of GK planes each (GX * GY * GZL = 256 threads).
//   G = 0:  8 (z) x 8 (y) x 16 (x)  the round-3 shape
//   G = 1: 16 (z) x 4 (y) x 32 (x)  twice the voxels on a footprint that is square in the z-x plane: a rotated block's bounding
//   G = 2: 16 (z) x 8 (y) x 32 (x)  box over-fetches by 1 + (a/b + b/a) sin cos for a block of a x b — least for a = b — and its
//                                   rows are twice as long for the staging loads (source rows of ~40 instead of ~20 voxels)
template <int G>
struct GGeo {
    static constexpr int GX = G == 0 ? 16 : 32, GY = G == 1 ? 4 : 8, GZL = 256 / (GX * GY), GK = G == 0 ? 4 : (G == 1 ? 8 : 16);
    static constexpr int BZ = GZL * GK;
};
template <typename TIN, int INTERP, int BOUNDARY, int G>
__global__ __launch_bounds__(256) void affine_gather_kernel(const TIN* __restrict__ in, float* __restrict__ out, AffineParams p,
                                                            int nbx, int nby, int nblocks, int per_xcd) {
    constexpr int GX = GGeo<G>::GX, GY = GGeo<G>::GY, GZL = GGeo<G>::GZL, GK = GGeo<G>::GK;
    extern __shared__ __attribute__((aligned(16))) float gtile[];
    __shared__ int gorg[3], gext[3];
    const int b = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);  // every XCD walks its own contiguous run of blocks (x fastest)
    if ((int)(blockIdx.x >> 3) >= per_xcd || b >= nblocks) return;
    const int bz = b / (nbx * nby), rem = b - bz * (nbx * nby), byi = rem / nbx, bxi = rem - byi * nbx;
    const int t = threadIdx.x;
    const int ox = bxi * GX + (t % GX), oy = byi * GY + ((t / GX) % GY), ozb = bz * (GZL * GK) + (t / (GX * GY)) * GK;
    const size_t sY = (size_t)p.Xi, sZ = (size_t)p.Yi * p.Xi;
    // The block's source box (compute_box's bound on the block's own extents), staged in LDS through load_clean when it fits
    // the launch's capacity: a compact block's box is a few thousand voxels at any angle, so every tap comes from LDS and each
    // source voxel is fetched once per block instead of once per tap.
    if (t < 3) {
        const int a = t, oz0 = bz * (GZL * GK), oy0 = byi * GY, ox0 = bxi * GX;
        const int z1 = min(oz0 + GZL * GK, p.Zo) - 1, y1 = min(oy0 + GY, p.Yo) - 1, x1 = min(ox0 + GX, p.Xo) - 1;
        const double base = p.m[4 * a] * (double)(oz0 + p.cz) + p.m[4 * a + 1] * (double)(oy0 + p.cy) +
                            p.m[4 * a + 2] * (double)(ox0 + p.cx) + p.m[4 * a + 3];
        const double ez = p.m[4 * a] * (double)(z1 - oz0), ey = p.m[4 * a + 1] * (double)(y1 - oy0), ex = p.m[4 * a + 2] * (double)(x1 - ox0);
        double lo = base + fmin(ez, 0.0) + fmin(ey, 0.0) + fmin(ex, 0.0);
        double hi = base + fmax(ez, 0.0) + fmax(ey, 0.0) + fmax(ex, 0.0);
        const double slack = 1e-9 * (fabs(lo) + fabs(hi) + 1.0);
        lo -= slack;
        hi += slack;
        const int n = a == 0 ? p.Zi : (a == 1 ? p.Yi : p.Xi);
        const double l = fmax(floor(lo), 0.0), h = fmin(floor(hi) + 1.0, (double)(n - 1));
        gorg[a] = (int)l;
        gext[a] = (h >= l) ? (int)(h - l) + 1 : 0;
    }
    __syncthreads();
    const int gz0 = gorg[0], gy0 = gorg[1], gx0 = gorg[2], gez = gext[0], gey = gext[1], gex = gext[2];
    const long gbox = (long)gez * gey * gex;
    const bool staged = gbox > 0 && gbox <= (long)p.lds_floats;
    if (staged) {
        // the box as a flat list, four loads in flight per thread; i / gex and r / gey by multiply-high with ceil(2^32 / d)
        // (exact while i * d < 2^32: i < 2^14 here) — two integer divisions per staged voxel cost more than sampling one
        const unsigned nb = (unsigned)gbox, uex = (unsigned)gex, uey = (unsigned)gey;
        const unsigned rx = 0xffffffffu / uex + 1u, ry = 0xffffffffu / uey + 1u;
        const TIN* gbase = in + (size_t)gz0 * sZ + (size_t)gy0 * sY + gx0;
        for (unsigned i0 = t; i0 < nb; i0 += 1024) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned i = min(i0 + 256u * u, nb - 1u);
                const unsigned r = uex == 1u ? i : __umulhi(i, rx), x = i - r * uex;
                const unsigned z = uey == 1u ? r : __umulhi(r, ry), y = r - z * uey;
                v[u] = load_clean(gbase + ((size_t)z * sZ + (size_t)y * sY + x));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + 256u * u < nb) gtile[i0 + 256u * u] = v[u];
        }
        __syncthreads();
    }
    if (ox >= p.Xo || oy >= p.Yo) return;
    auto fetch = [&](int iz, int iy, int ix) -> float {
        if (staged) return gtile[((iz - gz0) * gey + (iy - gy0)) * gex + (ix - gx0)];  // cleaned when it was staged
        return load_clean(in + (size_t)iz * sZ + (size_t)iy * sY + ix);
    };
    const int dims[3] = {p.Zi, p.Yi, p.Xi};
    if (INTERP == BH_INTERP_LINEAR && BOUNDARY != BH_BOUNDARY_ZEROS) {
        // the tile kernel's linear path for tiles that are not interior, voxel for voxel: Q32.32 coordinates (exact integer
        // arithmetic: the value does not depend on how the walk through z is organised), the inside decision on the float64
        // coordinate in numpy / ITK association, clamped taps through load_clean, lerp8 — bit-identical results
        long long c0[3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
            c0[a] = p.mq[4 * a] * (long long)(ozb + p.cz) + p.mq[4 * a + 1] * (long long)(oy + p.cy) +
                    p.mq[4 * a + 2] * (long long)(ox + p.cx) + p.mq[4 * a + 3];
        for (int k = 0; k < GK; ++k) {
            const int oz = ozb + k;
            if (oz >= p.Zo) break;
            const int iz = (int)(c0[0] >> 32), iy = (int)(c0[1] >> 32), ix = (int)(c0[2] >> 32);
            bool inside = true;
            {
#pragma clang fp contract(off)
                const double zd = (double)(oz + p.cz), yd = (double)(oy + p.cy), xd = (double)(ox + p.cx);
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const double ca = p.m[4 * a] * zd + p.m[4 * a + 1] * yd + p.m[4 * a + 2] * xd + p.m[4 * a + 3];
                    if (BOUNDARY == BH_BOUNDARY_ITK)
                        inside = inside && ca >= -0.5 && ca < (double)dims[a] - 0.5;
                    else
                        inside = inside && ca >= 0.0 && ca <= (double)(dims[a] - 1);
                }
            }
            float r = p.cval;
            if (inside) {
                const int z0 = max(0, min(iz, p.Zi - 1)), z1 = max(0, min(iz + 1, p.Zi - 1));
                const int y0 = max(0, min(iy, p.Yi - 1)), y1 = max(0, min(iy + 1, p.Yi - 1));
                const int x0 = max(0, min(ix, p.Xi - 1)), x1 = max(0, min(ix + 1, p.Xi - 1));
                // (one unaligned 8-byte load per x pair instead of two 4-byte ones was tried: 2-3x SLOWER)
                const f2 P0 = {fetch(z0, y0, x0), fetch(z0, y0, x1)};
                const f2 P1 = {fetch(z0, y1, x0), fetch(z0, y1, x1)};
                const f2 P2 = {fetch(z1, y0, x0), fetch(z1, y0, x1)};
                const f2 P3 = {fetch(z1, y1, x0), fetch(z1, y1, x1)};
                r = lerp8(P0, P1, P2, P3, (unsigned)c0[0], (unsigned)c0[1], (unsigned)c0[2]);
            }
            out[((size_t)oz * p.Yo + oy) * p.Xo + ox] = r;
#pragma unroll
            for (int a = 0; a < 3; ++a) c0[a] += p.mq[4 * a];
        }
        return;
    }
    double py[3], px[3];
    {
#pragma clang fp contract(off)
        const double yd = (double)(oy + p.cy), xd = (double)(ox + p.cx);
        for (int a = 0; a < 3; ++a) {
            py[a] = p.m[4 * a + 1] * yd;
            px[a] = p.m[4 * a + 2] * xd;
        }
    }
    for (int k = 0; k < GK; ++k) {
        const int oz = ozb + k;
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
        } else {
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

template <typename TIN, int G>
static int launch_affine_gather_g(bh_ctx* ctx, const TIN* in, float* out, const AffineParams& p) {
    constexpr int GX = GGeo<G>::GX, GY = GGeo<G>::GY, BZ = GGeo<G>::BZ;
    const int64_t nbx = ceil_div(p.Xo, GX), nby = ceil_div(p.Yo, GY), nbz = ceil_div(p.Zo, BZ);
    const int64_t nblocks = nbx * nby * nbz;
    BH_REQUIRE(nblocks < (1ll << 31) - 8, "affine output too large");
    const int per_xcd = (int)ceil_div(nblocks, (int64_t)8);
    const int grid = per_xcd * 8;
    // LDS for the source box of a full block (the bound of bh_affine for the tile kernel, on the block's extents): 48 KiB at
    // most for the small block (three workgroups per CU), 64 KiB for the larger ones (two); blocks whose box is larger gather
    // from global memory
    AffineParams q = p;
    {
        const int T[3] = {BZ, GY, GX};
        const double cap = G == 0 ? 12288.0 : 16384.0;
        double nb = 1.0;
        for (int a = 0; a < 3; ++a) {
            double span = 0.0;
            for (int j = 0; j < 3; ++j) span += std::fabs(p.m[4 * a + j]) * (double)(T[j] - 1);
            nb *= std::floor(span * (1.0 + 1e-6)) + 4.0;
        }
        q.lds_floats = nb < cap ? (int)nb : (int)cap;
    }
    auto run = [&](auto kern) -> int {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), (size_t)q.lds_floats * sizeof(float), ctx->stream, in, out, q, (int)nbx, (int)nby,
                           (int)nblocks, per_xcd);
        BH_CHECK_HIP(hipGetLastError());
        return BH_OK;
    };
#define BH_AFF(I, B) return run(affine_gather_kernel<TIN, I, B, G>)
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

template <typename TIN>
static int launch_affine_gather(bh_ctx* ctx, const TIN* in, float* out, const AffineParams& p) {
    const char* e = getenv("BH_AFFINE_GBLOCK");  // 0 / 1 / 2: block geometry (GGeo)
    const int g = e ? atoi(e) : 0;
    if (g == 1) return launch_affine_gather_g<TIN, 1>(ctx, in, out, p);
    if (g == 2) return launch_affine_gather_g<TIN, 2>(ctx, in, out, p);
    return launch_affine_gather_g<TIN, 0>(ctx, in, out, p);
}

template <typename TIN>
static int launch_affine(bh_ctx* ctx, const TIN* in, float* out, const AffineParams& p) {
    const int64_t ntx = ceil_div(p.Xo, ATX), nty = ceil_div(p.Yo, ATY), ntz = ceil_div(p.Zo, ATZ);
    const int64_t ntiles = ntx * nty * ntz;
    BH_REQUIRE(ntiles < (1ll << 31), "affine output too large");
    const int per_xcd = (int)ceil_div(ntiles, (int64_t)8);
    const int grid = per_xcd * 8;
    auto run = [&](auto kern) -> int {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(A_NT), (size_t)p.lds_floats * sizeof(float), ctx->stream, in, out, p, (int)ntx, (int)nty, (int)ntiles, per_xcd);
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

#include "affine_zwalk.inc"
#include "affine_zoblique.inc"

// spline.hip: prefilter + 64-tap gather (SciPy order 3, mode "constant")
int affine_cubic(bh_ctx* ctx, const void* in, int in_dtype, int64_t Zi, int64_t Yi, int64_t Xi, const double matrix[12], float cval,
                 float* out, int64_t Zo, int64_t Yo, int64_t Xo, const int64_t crop_lo[3]);

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
    BH_REQUIRE(interpolation == BH_INTERP_NEAREST || interpolation == BH_INTERP_LINEAR || interpolation == BH_INTERP_CUBIC,
               "unknown interpolation %d", interpolation);
    BH_REQUIRE(boundary >= BH_BOUNDARY_ITK && boundary <= BH_BOUNDARY_ZEROS, "unknown boundary %d", boundary);
    BH_REQUIRE(interpolation != BH_INTERP_CUBIC || boundary == BH_BOUNDARY_SCIPY_CONSTANT,
               "cubic B-spline interpolation is defined for the SciPy \"constant\" boundary only (got boundary %d)", boundary);
    for (int i = 0; i < 12; ++i) BH_REQUIRE(matrix[i] == matrix[i], "matrix contains NaN");
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    if (interpolation == BH_INTERP_CUBIC) {
        for (int i = 0; i < 12; ++i) BH_REQUIRE(std::fabs(matrix[i]) < 1073741824.0, "matrix entry %d out of range", i);
        ScopedTimer timer(ctx, T_AFFINE);
        return affine_cubic(ctx, in, in_dtype, Zi, Yi, Xi, matrix, cval, out, Zo, Yo, Xo, crop_lo);
    }
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
    p.x4 = (in_dtype == BH_DT_F32 && Xi % 4 == 0 && ((uintptr_t)in & 15) == 0) ? 1 : 0;
    p.x8 = ((in_dtype == BH_DT_U16 || in_dtype == BH_DT_I16) && Xi % 8 == 0 && ((uintptr_t)in & 15) == 0) ? 1 : 0;
    bool box_fits = true;
    // Upper bound of any tile's source box (compute_box: hi - lo <= sum |m| * (T - 1), then floor / floor + 1 and the
    // slack add at most 3): the launch asks for exactly that much LDS, so a gentle warp (a stabilisation shift needs
    // 10 x 10 x 66 floats) runs six workgroups per CU instead of four.  Tiles that exceed it gather from global memory.
    {
        const int T[3] = {ATZ, ATY, ATX};
        double nb = 1.0;
        for (int a = 0; a < 3; ++a) {
            double span = 0.0;
            for (int j = 0; j < 3; ++j) span += std::fabs(matrix[4 * a + j]) * (double)(T[j] - 1);
            double e = std::floor(span * (1.0 + 1e-6)) + 4.0;
            if (a == 2 && p.x4) e = std::floor((e + 6.0) / 4.0) * 4.0;
            if (a == 2 && p.x8) e = std::floor((e + 14.0) / 8.0) * 8.0;
            nb *= e;
        }
        const int cap = p.x8 ? A_LDS_FLOATS_X8 : A_LDS_FLOATS;
        p.lds_floats = nb < (double)cap ? (int)nb : cap;
        // The tile kernel stages every tile whose own box fits and gathers for the others, which it does well as long as a
        // row of 64 x-voxels stays within a few source planes.  The compact-block kernel takes over when a full tile's box
        // does not fit AND the tile's x extent crosses 2 planes or more (|m_zx| * 63: ~2 degrees about y; measured per angle
        // and axis by tools/affine_angle_sweep.py, profiles/r04al_affine_angle_sweep.txt — below that, and for z-y coupling
        // at any angle, the tile kernel is faster; round 3's compact blocks paid two integer divisions per staged voxel and
        // the threshold was 4).
        const double zx_min = getenv("BH_AFFINE_GATHER_ZX") ? atof(getenv("BH_AFFINE_GATHER_ZX")) : 2.0;
        box_fits = nb < (double)cap || std::fabs(matrix[2]) * (double)(ATX - 1) < zx_min;
    }
    p.zslot = 0;
    if ((p.x4 || p.x8) && getenv("BH_ZW_NOLDS") == nullptr) {
        // the largest source box of a wave's 64 x RY outputs (as for lds_floats above), whole 16-B quads; 4 KiB at most
        const double E = p.x4 ? 4.0 : 8.0, size = p.x4 ? 4.0 : 2.0;
        const double ey = std::floor((std::fabs(matrix[5]) * (zw::RY - 1) + std::fabs(matrix[6]) * 63.0) * (1.0 + 1e-6)) + 3.0;
        const double ex = std::floor((std::fabs(matrix[9]) * (zw::RY - 1) + std::fabs(matrix[10]) * 63.0) * (1.0 + 1e-6)) + 3.0;
        const double bytes = ey * (std::floor((ex + 2.0 * (E - 1.0)) / E) * E) * size;
        if (bytes <= 4096.0) p.zslot = ((int)bytes + 15) & ~15;
    }
    ScopedTimer timer(ctx, T_AFFINE);
    if (zw::takes(p)) {  // z-separable linear warp: wave-private z walk (affine_zwalk.inc)
        switch (in_dtype) {
            case BH_DT_F32: return zw::launch(ctx, (const float*)in, out, p);
            case BH_DT_U16: return zw::launch(ctx, (const uint16_t*)in, out, p);
            case BH_DT_U8: return zw::launch(ctx, (const uint8_t*)in, out, p);
            case BH_DT_I16: return zw::launch(ctx, (const int16_t*)in, out, p);
            default: BH_REQUIRE(false, "unsupported input dtype code %d", in_dtype);
        }
    }
    if (in_dtype == BH_DT_F32 || in_dtype == BH_DT_U16 || in_dtype == BH_DT_I16) {
        // weak z coupling: the z walk with per-lane source planes (affine_zoblique.inc)
        int64_t zchunk = 0;
        const int slot = zo::plan(p, &zchunk);
        if (slot && in_dtype == BH_DT_F32) return zo::launch(ctx, (const float*)in, out, p, slot, zchunk);
        if (slot && in_dtype == BH_DT_U16) return zo::launch(ctx, (const uint16_t*)in, out, p, slot, zchunk);
        if (slot) return zo::launch(ctx, (const int16_t*)in, out, p, slot, zchunk);
    }
    // a full tile's source box does not fit LDS: compact blocks gathering through the caches (BH_AFFINE_GATHER=0, and
    // BH_AFFINE_NOZWALK=1 — "everything on the tile kernel", the reference of the bit-identity tests —: the tile kernel)
    if (!box_fits && !(getenv("BH_AFFINE_GATHER") && atoi(getenv("BH_AFFINE_GATHER")) == 0) && getenv("BH_AFFINE_NOZWALK") == nullptr) {
        switch (in_dtype) {
            case BH_DT_F32: return launch_affine_gather(ctx, (const float*)in, out, p);
            case BH_DT_U16: return launch_affine_gather(ctx, (const uint16_t*)in, out, p);
            case BH_DT_U8: return launch_affine_gather(ctx, (const uint8_t*)in, out, p);
            case BH_DT_I16: return launch_affine_gather(ctx, (const int16_t*)in, out, p);
            default: BH_REQUIRE(false, "unsupported input dtype code %d", in_dtype);
        }
    }
    switch (in_dtype) {
        case BH_DT_F32: return launch_affine(ctx, (const float*)in, out, p);
        case BH_DT_U16: return launch_affine(ctx, (const uint16_t*)in, out, p);
        case BH_DT_U8: return launch_affine(ctx, (const uint8_t*)in, out, p);
        case BH_DT_I16: return launch_affine(ctx, (const int16_t*)in, out, p);
        default: BH_REQUIRE(false, "unsupported input dtype code %d", in_dtype);
    }
    return BH_OK;
}
