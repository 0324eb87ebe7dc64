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
// words hit 64 different banks), filters them and hands the coefficients back through LDS so that loads and stores both run
// with the lanes along x.  Three out-of-place passes (in -> A -> B -> A), float32 coefficients.  The 64-tap interpolation
// reads its taps from LDS-staged source boxes of 8 x 8 x 64 output tiles (gather_tile_kernel below).
// MI355X, 512 x 2048 x 2048 f32, 2 deg about z + shift: 27 ms = x pass 5.4 + y 3.3 + z 3.3 + interpolation 15.4 (round-4 first
// form: 94 ms; profiles/r04q_cubic_kernels.txt).  The interpolation is bound by its vector instructions (~195 per voxel: 64
// taps as 32 ds_read2_b32 + 53 packed FMAs, weights, float64 coordinates, addresses) at 65-70 % issue utilisation.
#include "common.hpp"

#include <cmath>
#include <cstdint>
#include <cstdlib>

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

template <typename TIN, bool VEC>
__global__ __launch_bounds__(256) void x_kernel(const TIN* __restrict__ src, float* __restrict__ dst, long rows, int X, int nchunk,
                                                int rpw, int tpr, int pitch) {
    extern __shared__ float lds[];
    const int clen = min(X, XCH);                             // outputs per chunk
    const int span = clen + 2 * R;                            // staged samples per row
    const int cpad = ((clen + BX - 1) & ~(BX - 1)) + 2 * R;   // samples the blocks read (zeros past span)
    const long unit0 = (long)blockIdx.x * rpw;                // first (row, chunk) unit of this workgroup
    const long nunits = rows * nchunk;
    const int nu = (int)min((long)rpw, nunits - unit0);
    auto where = [&](int u, long& row, int& p0) {  // (the 64-bit division only for rows longer than one chunk)
        const long unit = unit0 + u;
        if (nchunk == 1) {
            row = unit, p0 = 0;
        } else {
            row = unit / nchunk, p0 = (int)(unit - row * nchunk) * XCH;
        }
    };
    // stage, lanes along x.  VEC (X % 4 == 0, 4-sample-aligned base): groups of four samples per lane — R and XCH are
    // multiples of 4, and skew() keeps an aligned group of four contiguous.
    for (int u = 0; u < nu; ++u) {
        long row;
        int p0;
        where(u, row, p0);
        const TIN* rp = src + row * X;
        float* lr = lds + u * pitch;
        if (VEC) {
            struct alignas(4 * sizeof(TIN)) Quad { TIN v[4]; };
#pragma unroll 2
            for (int q = threadIdx.x; q < (span >> 2); q += 256) {
                const int j0 = p0 - R + 4 * q;
                float v[4];
                if (j0 >= 0 && j0 + 3 < X) {
                    const Quad g = *reinterpret_cast<const Quad*>(rp + j0);
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = clean<TIN>(g.v[k]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = clean<TIN>(rp[mirror(j0 + k, X)]);
                }
                float* w = lr + skew(4 * q);
#pragma unroll
                for (int k = 0; k < 4; ++k) w[k] = v[k];
            }
        } else {
            for (int i = threadIdx.x; i < span; i += 256) {
                int j = p0 - R + i;
                if ((unsigned)j >= (unsigned)X) j = mirror(j, X);  // (the modulo only at the row ends)
                lr[skew(i)] = clean<TIN>(rp[j]);
            }
        }
        for (int i = span + threadIdx.x; i < cpad; i += 256) lr[skew(i)] = 0.0f;
    }
    __syncthreads();
    // filter: thread (u, t) owns block t of unit u (tpr = ceil(clen / BX): exactly one block per thread).  skew(16 t + i) =
    // 17 t + i + (i >> 4): one base address per thread, the rest immediate offsets.
    const int u = threadIdx.x / tpr, t = threadIdx.x - u * tpr;
    const bool mine = u < nu && t * BX < clen;
    float* blk = lds + u * pitch + t * (BX + 1);
    float s[BX + 2 * R];
    if (mine) {
#pragma unroll
        for (int i = 0; i < BX + 2 * R; ++i) s[i] = blk[i + (i >> 4)];
        filter_block<BX>(s);
    }
    __syncthreads();  // every block has read its run-ins: the coefficients replace the samples in place
    if (mine) {
#pragma unroll
        for (int i = 0; i < BX; ++i) blk[(R + i) + ((R + i) >> 4)] = s[R + i];
    }
    __syncthreads();
    // store: lanes along x again (a thread's BX coefficients written straight from registers were 64-B strided per lane)
    for (int v = 0; v < nu; ++v) {
        long row;
        int p0;
        where(v, row, p0);
        const float* lr = lds + v * pitch;
        float* out = dst + row * X + p0;
        const int len = min(clen, X - p0);
        if (VEC) {
#pragma unroll 2
            for (int q = threadIdx.x; q < (len >> 2); q += 256) {
                const float* r = lr + skew(R + 4 * q);
                *reinterpret_cast<float4*>(out + 4 * q) = make_float4(r[0], r[1], r[2], r[3]);
            }
        } else {
            for (int i = threadIdx.x; i < len; i += 256) out[i] = lr[skew(R + i)];
        }
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

// Coordinates stay float64 (a float32 coordinate near 2048 resolves the fraction to 1e-4 only); the fraction is exact in float64
// and the weight polynomials run in float32 from it: relative error ~1e-7 per weight, below the float32 accumulation of the
// 64 taps.  (SciPy evaluates the same polynomials in float64, with three divisions by 6 per axis — 9 float64 divisions per
// voxel cost more than the 64 taps.)
__device__ __forceinline__ void weights3(double c, int& base, float (&w)[4]) {
    const double fl = floor(c);
    const float x = (float)(c - fl), z = 1.0f - x;
    constexpr float S = 1.0f / 6.0f;
    base = (int)fl - 1;
    w[0] = z * z * z * S;
    w[1] = __builtin_fmaf(x * x * (x - 2.0f), 0.5f, 4.0f * S);
    w[2] = __builtin_fmaf(z * z * (z - 2.0f), 0.5f, 4.0f * S);
    w[3] = x * x * x * S;
}

// The 4 x 4 x 4 blend of one voxel in packed float32: the x taps 0, 1 and 2, 3 of a row ride in the two halves of a register
// pair (ds_read2_b32 delivers them so), weighted by (wx0, wx1) and (wx2, wx3), folded over y and z with packed FMAs, and the
// two halves add at the very end — 53 packed operations instead of 84 scalar ones.  row(dz, dy) -> the coefficient row of taps
// (iz[dz], iy[dy]); ix[] the four x indices within it.
typedef float f2 __attribute__((ext_vector_type(2)));

template <typename RowFn>
__device__ __forceinline__ float blend64(const float (&wz)[4], const float (&wy)[4], const float (&wx)[4], const int (&ix)[4], RowFn row) {
    // all 64 taps first (32 two-word LDS reads in flight), then the arithmetic: interleaved, every group of reads is waited
    // for in turn and a voxel costs 8 LDS latencies
    f2 ta[4][4], tb[4][4];
#pragma unroll
    for (int dz = 0; dz < 4; ++dz)
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) {
            const float* rowp = row(dz, dy);
            ta[dz][dy] = f2{rowp[ix[0]], rowp[ix[1]]};
            tb[dz][dy] = f2{rowp[ix[2]], rowp[ix[3]]};
        }
    __builtin_amdgcn_sched_barrier(0);
    const f2 wa = {wx[0], wx[1]}, wb = {wx[2], wx[3]};
    f2 acc = {0.0f, 0.0f};
#pragma unroll
    for (int dz = 0; dz < 4; ++dz) {
        f2 az = {0.0f, 0.0f};
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) {
            const f2 ax = __builtin_elementwise_fma(tb[dz][dy], wb, ta[dz][dy] * wa);
            const f2 w = {wy[dy], wy[dy]};
            az = dy == 0 ? ax * w : __builtin_elementwise_fma(ax, w, az);
        }
        const f2 w = {wz[dz], wz[dz]};
        acc = dz == 0 ? az * w : __builtin_elementwise_fma(az, w, acc);
    }
    return acc.x + acc.y;
}

// The 4 x 4 blend of one voxel on a plane whose z taps were combined beforehand (gather_tile_kernel<.., ZUNI>): 8 two-word
// LDS reads, 13 packed operations.  row(dy) -> the plane's row of tap iy[dy]; ix[] the four x indices within it.
template <typename RowFn>
__device__ __forceinline__ float blend16(const float (&wy)[4], const float (&wx)[4], const int (&ix)[4], RowFn row) {
    f2 ta[4], tb[4];
#pragma unroll
    for (int dy = 0; dy < 4; ++dy) {
        const float* rowp = row(dy);
        ta[dy] = f2{rowp[ix[0]], rowp[ix[1]]};
        tb[dy] = f2{rowp[ix[2]], rowp[ix[3]]};
    }
    const f2 wa = {wx[0], wx[1]}, wb = {wx[2], wx[3]};
    f2 az = {0.0f, 0.0f};
#pragma unroll
    for (int dy = 0; dy < 4; ++dy) {
        const f2 ax = __builtin_elementwise_fma(tb[dy], wb, ta[dy] * wa);
        const f2 w = {wy[dy], wy[dy]};
        az = dy == 0 ? ax * w : __builtin_elementwise_fma(ax, w, az);
    }
    return az.x + az.y;
}

// One voxel whose taps come through the vector cache (the fallback of the tile kernel, and the whole of gather_kernel).
__device__ __forceinline__ float sample_global(const float* __restrict__ coef, const GatherParams& p, double c0, double c1, double c2) {
    // SciPy "constant": a coordinate outside [0, n - 1] gives cval (no interpolation past the edge samples)
    if (!(c0 >= 0.0 && c0 <= (double)(p.Zi - 1) && c1 >= 0.0 && c1 <= (double)(p.Yi - 1) && c2 >= 0.0 && c2 <= (double)(p.Xi - 1)))
        return p.cval;
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
    return blend64(wz, wy, wx, ix, [&](int dz, int dy) { return coef + ((long)iz[dz] * p.Yi + iy[dy]) * p.Xi; });
}

// One output voxel per thread, 8 x 8 x 4 voxel blocks per workgroup (x fastest), every tap through the vector cache: the
// launch for warps whose source boxes do not fit LDS (strong rotations / scalings).
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
    out[((long)oz * p.Yo + oy) * p.Xo + ox] = sample_global(coef, p, c0, c1, c2);
}

// ---- interpolation from LDS: a workgroup owns an 8 x 8 x 64 (z, y, x) output tile -------------------------------------------
// It bounds the tile's source box (an affine map of a box is bounded per axis by base + the negative / positive edge extents;
// taps reach floor(c) - 1 .. floor(c) + 2), stages the box clipped to the volume by LDS-DMA (16 B per lane over the box as a
// flat list of quads when the rows are 16-B aligned) and reads the 64 taps of every voxel from LDS: the global gather spends
// 64 vector-cache accesses per voxel on lines its neighbours fetch too (79 ms at 512 x 2048 x 2048).  Tiles whose taps can
// neither leave the volume nor need mirroring skip every per-voxel test; the others test per voxel and fall back to the cache
// for a voxel whose mirrored taps lie outside the box.  Workgroup b takes tile (b % 8) * ntiles / 8 + b / 8: every XCD walks
// its own contiguous run of tiles and neighbours share their halo in that XCD's L2.
constexpr int GTX = 64, GTY = 8;  // GTZ = 8 or 4 and the workgroup's threads G_NT = 256 or 512: template parameters
constexpr int G_LDS_FLOATS = 20480;  // 80 KiB: two workgroups per CU; the launch asks for what the matrix can need

struct GBox {
    int org[3], ext[3], interior[3];
    unsigned rcp_l, rcp_dy;  // ceil(2^32 / (ext_x / 4)), ceil(2^32 / ext_y): exact small divisions while staging
};

template <int GTZ, int G_NT, bool ZUNI>
__global__ __launch_bounds__(G_NT) void gather_tile_kernel(const float* __restrict__ coef, float* __restrict__ out, GatherParams p,
                                                           int ntx, int nty, int ntiles, int per_xcd, int lds_floats, int x4, int pad32) {
#pragma clang fp contract(off)
    constexpr int G_NW = G_NT / 64;
    extern __shared__ __attribute__((aligned(16))) float tile[];  // (16-B: the plane-combining path reads and writes it 16 B wide)
    __shared__ GBox b;
    const int t = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || t >= ntiles) return;
    const int tzi = t / (ntx * nty), rem = t - tzi * (ntx * nty), tyi = rem / ntx, txi = rem - tyi * ntx;
    const int ox0 = txi * GTX, oy0 = tyi * GTY, oz0 = tzi * GTZ;
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        const int z1 = min(oz0 + GTZ, p.Zo) - 1, y1 = min(oy0 + GTY, p.Yo) - 1, x1 = min(ox0 + GTX, p.Xo) - 1;
        const double base = p.m[4 * a] * (double)(oz0 + p.cz) + p.m[4 * a + 1] * (double)(oy0 + p.cy) +
                            p.m[4 * a + 2] * (double)(ox0 + p.cx) + p.m[4 * a + 3];
        const double ez = p.m[4 * a] * (double)(z1 - oz0), ey = p.m[4 * a + 1] * (double)(y1 - oy0),
                     ex = p.m[4 * a + 2] * (double)(x1 - ox0);
        double lo = base + fmin(ez, 0.0) + fmin(ey, 0.0) + fmin(ex, 0.0);
        double hi = base + fmax(ez, 0.0) + fmax(ey, 0.0) + fmax(ex, 0.0);
        const double slack = 1e-9 * (fabs(lo) + fabs(hi) + 1.0);  // rounding difference to the per-voxel evaluation
        lo -= slack;
        hi += slack;
        const int n = a == 0 ? p.Zi : (a == 1 ? p.Yi : p.Xi);
        const double l = fmax(floor(lo) - 1.0, 0.0), h = fmin(floor(hi) + 2.0, (double)(n - 1));
        int org = 0, ext = 0;
        if (h >= l) org = (int)l, ext = (int)(h - l) + 1;
        if (a == 2 && x4 && ext > 0) {  // whole 16-B quads; Xi % 4 == 0, so this stays inside the row
            const int end = (org + ext + 3) & ~3;
            org &= ~3;
            ext = end - org;
            const int pq = (pad32 ? (ext + 31) & ~31 : ext) >> 2;  // quads per LDS row
            b.rcp_l = (unsigned)(0xffffffffu / (unsigned)pq) + 1u;  // wraps to 0 for 1: handled by the user
        }
        if (a == 1 && ext > 0) b.rcp_dy = (unsigned)(0xffffffffu / (unsigned)ext) + 1u;
        b.org[a] = org;
        b.ext[a] = ext;
        // interior: every voxel of the tile lies inside [0, n - 1] on this axis and none of its taps is mirrored
        b.interior[a] = (floor(lo) - 1.0 >= 0.0 && floor(hi) + 2.0 <= (double)(n - 1)) ? 1 : 0;
    }
    __syncthreads();
    const int bz0 = b.org[0], by0 = b.org[1], bx0 = b.org[2];
    const int dz = b.ext[0], dy = b.ext[1], dx = b.ext[2];
    // LDS row pitch: a multiple of 32 words, so that the bank of a tap is (x + const) mod 32 whatever row or plane the lane
    // reads — under a rotation the lanes of a 32-lane group sit on two or three source rows, and with the natural pitch (72
    // words: 8 banks on) every such group paid a 2-way conflict (half of all LDS cycles, SQ_LDS_BANK_CONFLICT)
    const int P = pad32 ? (dx + 31) & ~31 : dx;
    const long nbox = (long)dz * dy * P;
    const bool staged = nbox > 0 && nbox <= (long)lds_floats;
    const int tx = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (staged) {
        const size_t sY = (size_t)p.Xi, sZ = (size_t)p.Yi * p.Xi;
        const int nrows = dz * dy;
        const float* base = coef + (size_t)bz0 * sZ + (size_t)by0 * sY + bx0;
        if (x4) {
            // the box as a flat list of quads, row pitch P: the lane-linear LDS image IS the padded box (1 KiB per
            // instruction whatever the row length, each lane fetching from its own row); the pad quads' lanes stay idle
            const unsigned L = (unsigned)dx >> 2, LP = (unsigned)P >> 2, S = (unsigned)nrows * LP;
            for (unsigned i = (unsigned)wv * 64; i < S; i += G_NT) {
                const unsigned q = i + tx;
                // q / LP and r / dy by multiply-high with ceil(2^32 / d): exact while q * d < 2^32 (q < 10^4 here)
                const unsigned r = LP == 1 ? q : __umulhi(q, b.rcp_l), xq = q - r * LP;
                if (q < S && xq < L) {
                    const unsigned z = dy == 1 ? r : __umulhi(r, b.rcp_dy), y = r - z * (unsigned)dy;
                    const float* src = base + ((size_t)z * sZ + (size_t)y * sY + 4 * xq);
                    const unsigned lds_dst = (unsigned)(size_t)(tile + 4 * i);
                    unsigned keep;
                    asm volatile(
                        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                        : "=&s"(keep)
                        : "v"(src), "s"(lds_dst)
                        : "memory");
                }
            }
        } else {
            int z = wv / dy, y = wv - z * dy;  // row r = z * dy + y, advanced by G_NW rows per step
            const int qz = G_NW / dy, qy = G_NW - qz * dy;
            for (int r = wv; r < nrows; r += G_NW) {
                const float* rowp = base + (size_t)z * sZ + (size_t)y * sY;
                for (int x0 = 0; x0 < dx; x0 += 64) {
                    if (x0 + tx < dx) {
                        const float* src = rowp + x0 + tx;
                        const unsigned lds_dst = (unsigned)(size_t)(tile + r * P + x0);
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
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const bool interior = staged && b.interior[0] && b.interior[1] && b.interior[2];
    const int ox = ox0 + tx;
    const int pz = dy * P;  // plane pitch of the box
    if (ZUNI && staged && (pz & 3) == 0 && b.interior[0] && p.m[0] >= 1.0) {
        // The matrix leaves z alone (m01 = m02 = 0: shifts, scalings, rotations about z — registration and stabilisation
        // transforms): every voxel of an output plane has the same four source planes and z weights.  The workgroup combines
        // them ONCE per output plane, and a voxel is a 4 x 4 blend on the combined plane: 16 taps instead of 64, a third of the
        // vector instructions.  The combined planes replace the box IN PLACE, without a barrier between planes: a thread owns
        // 16-B columns of the box and walks the output planes upwards; with m00 >= 1 the first source plane b_k of output plane
        // k grows strictly, so plane b_k is read by no later output plane and takes the combination of k.  (z-interior tiles
        // only: no mirrored planes, every plane in range; the other tiles, and matrices that compress z, take the 64-tap path.)
        for (int q = threadIdx.x; q < (pz >> 2); q += G_NT) {
            float4* col = reinterpret_cast<float4*>(tile) + q;
            const int pq = pz >> 2;
#pragma unroll
            for (int kz = 0; kz < GTZ; ++kz) {
                if (oz0 + kz < p.Zo) {
                    int bz;
                    float wz[4];
                    weights3(p.m[0] * (double)(oz0 + kz + p.cz) + p.m[3], bz, wz);
                    float4* s0 = col + (bz - bz0) * pq;
                    const float4 a = s0[0], c = s0[pq], e = s0[2 * pq], g = s0[3 * pq];
                    float4 r;
                    r.x = __builtin_fmaf(wz[3], g.x, __builtin_fmaf(wz[2], e.x, __builtin_fmaf(wz[1], c.x, wz[0] * a.x)));
                    r.y = __builtin_fmaf(wz[3], g.y, __builtin_fmaf(wz[2], e.y, __builtin_fmaf(wz[1], c.y, wz[0] * a.y)));
                    r.z = __builtin_fmaf(wz[3], g.z, __builtin_fmaf(wz[2], e.z, __builtin_fmaf(wz[1], c.z, wz[0] * a.z)));
                    r.w = __builtin_fmaf(wz[3], g.w, __builtin_fmaf(wz[2], e.w, __builtin_fmaf(wz[1], c.w, wz[0] * a.w)));
                    s0[0] = r;
                }
            }
        }
        __syncthreads();
        if (ox >= p.Xo) return;
        const bool yx_interior = b.interior[1] && b.interior[2];
        const double gx = (double)(ox + p.cx);
        for (int r = wv; r < GTZ * GTY; r += G_NW) {
            const int oz = oz0 + (r >> 3), oy = oy0 + (r & 7);
            if (oy >= p.Yo || oz >= p.Zo) continue;
            const double gz = (double)(oz + p.cz), gy = (double)(oy + p.cy);
            const double c0 = p.m[0] * gz + p.m[3];  // (+ 0 * gy + 0 * gx)
            const double c1 = p.m[4] * gz + p.m[5] * gy + p.m[6] * gx + p.m[7];
            const double c2 = p.m[8] * gz + p.m[9] * gy + p.m[10] * gx + p.m[11];
            const float* buf = tile + ((int)floor(c0) - 1 - bz0) * pz;  // the combined plane of this output plane
            float v;
            if (yx_interior) {
                int by, bx;
                float wy[4], wx[4];
                weights3(c1, by, wy);
                weights3(c2, bx, wx);
                const float* q = buf + (by - by0) * P + (bx - bx0);
                const int ix[4] = {0, 1, 2, 3};
                v = blend16(wy, wx, ix, [&](int ky) { return q + ky * P; });
            } else if (!(c1 >= 0.0 && c1 <= (double)(p.Yi - 1) && c2 >= 0.0 && c2 <= (double)(p.Xi - 1))) {
                v = p.cval;
            } else {
                int by, bx;
                float wy[4], wx[4];
                weights3(c1, by, wy);
                weights3(c2, bx, wx);
                int ix[4], iy[4];
                bool inbox = true;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    ix[k] = mirror(bx + k, p.Xi) - bx0;
                    iy[k] = mirror(by + k, p.Yi) - by0;
                    inbox = inbox && (unsigned)ix[k] < (unsigned)dx && (unsigned)iy[k] < (unsigned)dy;
                }
                if (inbox)
                    v = blend16(wy, wx, ix, [&](int ky) { return buf + iy[ky] * P; });
                else
                    v = sample_global(coef, p, c0, c1, c2);
            }
            out[((long)oz * p.Yo + oy) * p.Xo + ox] = v;
        }
        return;
    }
    if (ox >= p.Xo) return;
    const double gx = (double)(ox + p.cx);
    for (int r = wv; r < GTZ * GTY; r += G_NW) {
        const int oz = oz0 + (r >> 3), oy = oy0 + (r & 7);
        if (oy >= p.Yo || oz >= p.Zo) continue;
        const double gz = (double)(oz + p.cz), gy = (double)(oy + p.cy);
        const double c0 = p.m[0] * gz + p.m[1] * gy + p.m[2] * gx + p.m[3];
        const double c1 = p.m[4] * gz + p.m[5] * gy + p.m[6] * gx + p.m[7];
        const double c2 = p.m[8] * gz + p.m[9] * gy + p.m[10] * gx + p.m[11];
        float v;
        if (interior) {
            int bz, by, bx;
            float wz[4], wy[4], wx[4];
            weights3(c0, bz, wz);
            weights3(c1, by, wy);
            weights3(c2, bx, wx);
            const float* q = tile + ((bz - bz0) * dy + (by - by0)) * P + (bx - bx0);
            const int ix[4] = {0, 1, 2, 3};
            v = blend64(wz, wy, wx, ix, [&](int kz, int ky) { return q + kz * pz + ky * P; });
        } else if (!(c0 >= 0.0 && c0 <= (double)(p.Zi - 1) && c1 >= 0.0 && c1 <= (double)(p.Yi - 1) && c2 >= 0.0 &&
                     c2 <= (double)(p.Xi - 1))) {
            v = p.cval;
        } else {
            int bz, by, bx;
            float wz[4], wy[4], wx[4];
            weights3(c0, bz, wz);
            weights3(c1, by, wy);
            weights3(c2, bx, wx);
            int ix[4], iy[4], iz[4];
            bool inbox = staged;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ix[k] = mirror(bx + k, p.Xi) - bx0;
                iy[k] = mirror(by + k, p.Yi) - by0;
                iz[k] = mirror(bz + k, p.Zi) - bz0;
                inbox = inbox && (unsigned)ix[k] < (unsigned)dx && (unsigned)iy[k] < (unsigned)dy && (unsigned)iz[k] < (unsigned)dz;
            }
            if (inbox)
                v = blend64(wz, wy, wx, ix, [&](int kz, int ky) { return tile + iz[kz] * pz + iy[ky] * P; });
            else
                v = sample_global(coef, p, c0, c1, c2);
        }
        out[((long)oz * p.Yo + oy) * p.Xo + ox] = v;
    }
}

// floats of LDS the largest source box of a full GTZ x 8 x 64 tile can take under this matrix (0: does not fit G_LDS_FLOATS)
static int gather_lds_floats(const GatherParams& p, bool x4, int gtz, bool pad32) {
    const int T[3] = {gtz, GTY, GTX};
    const int n[3] = {p.Zi, p.Yi, p.Xi};
    double need = 1.0;
    for (int a = 0; a < 3; ++a) {
        double e = 0.0;
        for (int j = 0; j < 3; ++j) e += std::fabs(p.m[4 * a + j]) * (double)(T[j] - 1);
        if (!(e < 1e6)) return 0;
        // floor(hi) + 2 - (floor(lo) - 1) + 1 <= floor(hi - lo) + 5, hi - lo <= e (1 + 2e-9) + 2e-9 (1 + 2 |base|): 1e-3 covers it
        int ext = (int)std::floor(e + 1e-3) + 5;
        if (a == 2 && x4) ext = (ext + 3 + 3) & ~3;  // origin rounded down, end rounded up to whole quads
        ext = std::min(ext, n[a]);                   // (Xi % 4 == 0 under x4)
        if (a == 2 && pad32) ext = (ext + 31) & ~31;  // the LDS row pitch
        need *= (double)ext;
    }
    return need <= (double)G_LDS_FLOATS ? (int)need : 0;
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
        const int cpad = (int)ceil_div(clen, BX) * BX + 2 * R, pitch = skew(cpad) + 1;  // what the blocks read: zeros past the staged span
        int rpw = std::max(1, 256 / tpr);
        rpw = std::min<int>(rpw, std::max(1, (int)(160 * 1024 / 4 / pitch) - 0));
        const size_t lds = (size_t)rpw * pitch * sizeof(float);
        BH_REQUIRE(lds <= 160 * 1024, "spline prefilter row chunk does not fit LDS");
        const long nunits = rows * nchunk;
        const long grid = ceil_div(nunits, rpw);
        BH_REQUIRE(grid < (1ll << 31), "volume too large for the spline prefilter");
        const bool vec = X % 4 == 0 && reinterpret_cast<uintptr_t>(in) % (4 * sizeof(TIN)) == 0 && reinterpret_cast<uintptr_t>(coef) % 16 == 0;
        auto kern = vec ? x_kernel<TIN, true> : x_kernel<TIN, false>;
        if (lds > 64 * 1024)
            BH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, in, coef, rows, (int)X, nchunk, rpw, tpr, pitch);
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
    const bool x4 = Xi % 4 == 0 && (reinterpret_cast<uintptr_t>(coef) & 15) == 0;
    const char* force = getenv("BH_SPLINE_GATHER");  // "global": every tap through the vector cache (the round-4 first form)
    // 8 output planes per tile and 512 threads (two workgroups = 16 wavefronts per CU at a registration-sized warp: 26.9 ms of
    // the whole operator at 512 x 2048 x 2048 against 27.9-30.1 for 4 planes or 256 threads); 4 planes when 8 do not fit
    const char* tz_env = getenv("BH_SPLINE_TZ");
    const char* nt_env = getenv("BH_SPLINE_NT");
    const char* pad_env = getenv("BH_SPLINE_PITCH32");  // 0: natural LDS row pitch (the first form; A/B)
    const bool pad32 = !(pad_env && pad_env[0] == '0');
    const bool global_only = force && force[0] == 'g';
    // a matrix that leaves z alone (and does not compress it) combines the four source planes once per output plane
    // (gather_tile_kernel<.., ZUNI>)
    const char* zu_env = getenv("BH_SPLINE_ZUNI");  // 0: the general 64-tap path for every matrix (A/B, parity tests)
    const bool zuni = matrix[1] == 0.0 && matrix[2] == 0.0 && matrix[0] >= 1.0 && !(zu_env && zu_env[0] == '0');
    int gtz = tz_env && atoi(tz_env) == 4 ? 4 : 8;
    int lds_floats = global_only ? 0 : sp::gather_lds_floats(p, x4, gtz, pad32);
    if (lds_floats == 0 && gtz == 8 && !global_only) {
        gtz = 4;
        lds_floats = sp::gather_lds_floats(p, x4, 4, pad32);
    }
    const int nt = nt_env ? atoi(nt_env) : (gtz == 8 ? 512 : 256);
    const int64_t ntx = ceil_div(Xo, sp::GTX), nty = ceil_div(Yo, sp::GTY), ntz = ceil_div(Zo, gtz), ntiles = ntx * nty * ntz;
    if (lds_floats > 0 && ntiles < (1ll << 30)) {
        const int per_xcd = (int)ceil_div(ntiles, (int64_t)8);
        const size_t lds = (size_t)lds_floats * sizeof(float);
        auto kern = zuni ? (gtz == 8 ? (nt == 512 ? sp::gather_tile_kernel<8, 512, true> : sp::gather_tile_kernel<8, 256, true>)
                                     : (nt == 512 ? sp::gather_tile_kernel<4, 512, true> : sp::gather_tile_kernel<4, 256, true>))
                         : (gtz == 8 ? (nt == 512 ? sp::gather_tile_kernel<8, 512, false> : sp::gather_tile_kernel<8, 256, false>)
                                     : (nt == 512 ? sp::gather_tile_kernel<4, 512, false> : sp::gather_tile_kernel<4, 256, false>));
        if (lds > 64 * 1024)
            BH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)(per_xcd * 8)), dim3(nt == 512 ? 512 : 256), lds, ctx->stream, coef, out, p, (int)ntx, (int)nty,
                           (int)ntiles, per_xcd, lds_floats, x4 ? 1 : 0, pad32 ? 1 : 0);
    } else {
        const dim3 grid((unsigned)ceil_div(Xo, 8), (unsigned)ceil_div(Yo, 8), (unsigned)ceil_div(Zo, 4));
        BH_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "affine output too large");
        hipLaunchKernelGGL(sp::gather_kernel, grid, dim3(256), 0, ctx->stream, coef, out, p);
    }
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
