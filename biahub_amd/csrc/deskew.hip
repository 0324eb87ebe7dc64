// Fused oblique-light-sheet deskew for gfx950 (MI355X).
//
// Replaces, in one kernel and one pass over HBM, the reference's chain
//   permute/flip copy (biahub/deskew.py:110) -> edge pad (:517-519) -> grid build (:113-154)
//   -> F.grid_sample (:531-533) -> mean over N (:536)
//
//   out[a, yo, xo] = (1/N) * sum_{k<N} lerp( in[:, Y-1-min(aN+k, Y-1), X-1-yo], ix(xo, aN+k) )
//
// Data layout / access pattern
//   in  (Z, Y, X)  : X contiguous.  For fixed (a, k) the kernel needs the (Z x X) plane
//                    in[:, yin, :]; a workgroup stages a [z-window][TX] tile of it with
//                    row-contiguous (coalesced) reads and stores it TRANSPOSED in LDS as
//                    [k][x][z] with an odd z-stride, so that the compute phase — lanes along
//                    the output x axis, which walks input z at px_to_scan_ratio per step —
//                    reads consecutive LDS banks.
//   out (Za, X, Xp): Xp contiguous.  Each wave writes 64 consecutive floats per store.
//   Every input voxel is read once (plus a 2-3 row overlap between neighbouring x-chunks)
//   and every output voxel written once: algorithmic bytes 4*(V_in + V_out).
//
// Coordinates reproduce the reference's float32 arithmetic operation by operation
// (oracle/oracle_np.py:deskew_coords), so sample positions are bit-identical to torch's.
#include "common.hpp"

#include <cmath>

namespace bh {

struct DeskewGeom {
    int Z, Y, X;     // input
    int Za, Xp;      // output (Za, X, Xp)
    int N;           // average_n_slices
    float px, pxct, offset, zm1;
    int XC;          // output-x chunk per workgroup (multiple of 256)
    int ZS;          // LDS z stride (odd)
    int ZC;          // max z-window length (<= ZS)
};

// The reference's sample position along the scan axis, in its float32 operation order:
//   in_z = px*x - (px*ct)*zo + offset ; g = 2*in_z/(Z-1) - 1 ; ix = ((g+1)/2)*(Z-1)
__host__ __device__ inline float deskew_ix(float px, float pxct, float offset, float zm1, int xo, int zo) {
#pragma clang fp contract(off)
    float t1 = px * (float)xo;
    float t2 = pxct * (float)zo;
    float in_z = (t1 - t2) + offset;
    float g = (2.0f * in_z) / zm1 - 1.0f;
    float ix = ((g + 1.0f) / 2.0f) * zm1;
    return ix;
}

template <typename T>
__device__ __forceinline__ float to_f32(T v) {
    return (float)v;
}

// NK > 0: N == NK known at compile time (interpolation plan kept in registers).
// NK == 0: generic N, plan recomputed per row.
template <typename TIN, int TX, int NT, int NK>
__global__ __launch_bounds__(NT) void deskew_kernel(const TIN* __restrict__ in, float* __restrict__ out,
                                                    DeskewGeom g) {
    extern __shared__ __attribute__((aligned(16))) float tile[];  // [N][TX][ZS]
    const int tid = threadIdx.x;
    const int xt0 = blockIdx.x * TX;
    const int xo0 = blockIdx.y * g.XC;
    const int a = blockIdx.z;
    const int N = NK > 0 ? NK : g.N;
    const int xoN = min(g.XC, g.Xp - xo0);
    const int zo0 = a * N;

    // z-window covering every sample of this (a, xo-chunk): ix is monotone in xo and zo
    const float ix_min = deskew_ix(g.px, g.pxct, g.offset, g.zm1, xo0, zo0 + N - 1);
    const float ix_max = deskew_ix(g.px, g.pxct, g.offset, g.zm1, xo0 + xoN - 1, zo0);
    const int zlo = (int)floorf(ix_min);
    int zcnt = (int)floorf(ix_max) + 2 - zlo;
    zcnt = min(zcnt, g.ZC);  // host guarantees zcnt <= ZC; clamp is a memory-safety net only

    // ---- stage: global (rows of TX contiguous x) -> LDS transposed [k][x][z] ----------
    const size_t plane = (size_t)g.Y * g.X;
    for (int k = 0; k < N; ++k) {
        const int yin = g.Y - 1 - min(zo0 + k, g.Y - 1);
        const TIN* src = in + (size_t)yin * g.X + xt0;
        float* dst = tile + (size_t)k * TX * g.ZS;
        const int xl = tid % TX;
        const bool xok = (xt0 + xl) < g.X;
        for (int zz = tid / TX; zz < zcnt; zz += NT / TX) {
            const int z = zlo + zz;
            float v = 0.0f;
            if (xok && z >= 0 && z < g.Z) v = to_f32(src[(size_t)z * plane + xl]);
            dst[xl * g.ZS + zz] = v;
        }
    }
    __syncthreads();

    // ---- compute: lanes along xo, 4 outputs per lane spaced by 64 --------------------
    constexpr int WAVES = NT / 64;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int WR = g.XC / 256;           // waves needed to cover one output row chunk
    const int sub = wave % WR;           // which 256-wide part of the chunk
    const int row0 = wave / WR;
    const int RG = WAVES / WR;           // rows processed concurrently
    const int xbase = xo0 + sub * 256 + lane;

    constexpr int NKK = NK > 0 ? NK : 1;
    int i0[NKK][4];
    float w0[NKK][4], w1[NKK][4];
    if (NK > 0) {
#pragma unroll
        for (int k = 0; k < NKK; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float ix = deskew_ix(g.px, g.pxct, g.offset, g.zm1, xbase + 64 * j, zo0 + k);
                const float fl = floorf(ix);
                w1[k][j] = ix - fl;
                w0[k][j] = (fl + 1.0f) - ix;
                int rel = (int)fl - zlo;
                rel = max(0, min(rel, g.ZC - 2));  // lanes past Xp may fall outside the window
                i0[k][j] = k * TX * g.ZS + rel;
            }
    }
    const float invN_is_div = (float)N;
    for (int xl = row0; xl < TX; xl += RG) {
        const int x = xt0 + xl;
        if (x >= g.X) break;
        const int yo = g.X - 1 - x;
        float* orow = out + ((size_t)a * g.X + yo) * g.Xp;
        const float* trow = tile + xl * g.ZS;
        float acc[4];
        if (NK > 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s = 0.0f;
#pragma unroll
                for (int k = 0; k < NKK; ++k) {
                    const float v0 = trow[i0[k][j]];
                    const float v1 = trow[i0[k][j] + 1];
                    const float val = v0 * w0[k][j] + v1 * w1[k][j];
                    s = (k == 0) ? val : s + val;
                }
                acc[j] = s;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s = 0.0f;
                for (int k = 0; k < N; ++k) {
                    const float ix = deskew_ix(g.px, g.pxct, g.offset, g.zm1, xbase + 64 * j, zo0 + k);
                    const float fl = floorf(ix);
                    int rel = (int)fl - zlo;
                    rel = max(0, min(rel, g.ZC - 2));
                    const float* p = trow + k * TX * g.ZS + rel;
                    const float val = p[0] * ((fl + 1.0f) - ix) + p[1] * (ix - fl);
                    s = (k == 0) ? val : s + val;
                }
                acc[j] = s;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int xo = xbase + 64 * j;
            if (xo < g.Xp) orow[xo] = (N > 1) ? acc[j] / invN_is_div : acc[j];
        }
    }
}

struct LaunchCfg {
    int TX, NT, XC;
};

static int deskew_geometry(int64_t Z, int64_t Y, int64_t X, double angle, double ratio, int keep_overhang,
                           int n, DeskewGeom* g, int64_t out_shape[3]) {
    double voxel[3];
    BH_TRY(bh_deskew_shape(Z, Y, X, angle, ratio, keep_overhang, n, 1.0, out_shape, voxel));
    // un-averaged geometry drives the shear offset (deskew.py:499-503: Z_out_full = Y)
    const double ct = std::cos(angle * M_PI / 180.0);
    const double px = ratio;
    const int64_t Xp = out_shape[2];
    const double offset = px * ct * (double)(Y - 1) / 2 - px * (double)(Xp - 1) / 2 + (double)(Z - 1) / 2;
    g->Z = (int)Z;
    g->Y = (int)Y;
    g->X = (int)X;
    g->Za = (int)out_shape[0];
    g->Xp = (int)Xp;
    g->N = n;
    g->px = (float)px;
    g->pxct = (float)(px * ct);
    g->offset = (float)offset;
    g->zm1 = (float)(Z - 1);
    return BH_OK;
}

// Exact maximum z-window over every (a, chunk) for a candidate XC, using the device formula.
static int max_window(const DeskewGeom& g, int XC) {
    int worst = 0;
    for (int a = 0; a < g.Za; ++a) {
        const int zo0 = a * g.N;
        for (int xo0 = 0; xo0 < g.Xp; xo0 += XC) {
            const int xoN = std::min(XC, g.Xp - xo0);
            const float lo = deskew_ix(g.px, g.pxct, g.offset, g.zm1, xo0, zo0 + g.N - 1);
            const float hi = deskew_ix(g.px, g.pxct, g.offset, g.zm1, xo0 + xoN - 1, zo0);
            const int cnt = (int)std::floor(hi) + 2 - (int)std::floor(lo);
            worst = std::max(worst, cnt);
        }
    }
    return worst;
}

template <typename TIN>
static int launch_deskew(bh_ctx* ctx, const TIN* in, float* out, DeskewGeom g) {
    constexpr int TX = 32;
    constexpr int NT = 256;
    // pick the largest chunk whose [N][TX][ZS] tile leaves room for two workgroups per CU
    const size_t lds_budget = 78 * 1024;
    int XC = 0, ZC = 0;
    for (int cand : {1024, 512, 256}) {
        if (cand > 256 && cand / 2 >= g.Xp) continue;  // do not over-size tiny problems
        const int zc = max_window(g, cand);
        const int zs = zc | 1;
        if ((size_t)g.N * TX * zs * sizeof(float) <= lds_budget || cand == 256) {
            XC = cand;
            ZC = zc;
            break;
        }
    }
    g.XC = XC;
    g.ZC = ZC;
    g.ZS = ZC | 1;
    const size_t lds = (size_t)g.N * TX * g.ZS * sizeof(float);
    BH_REQUIRE(lds <= 160 * 1024,
               "deskew tile needs %zu bytes of LDS (px_to_scan_ratio=%g, average_n_slices=%d) — exceeds 160 KiB",
               lds, (double)g.px, g.N);
    // waves per row chunk must divide the workgroup's wave count
    BH_REQUIRE((NT / 64) % (XC / 256) == 0, "internal: XC=%d incompatible with %d threads", XC, NT);
    dim3 grid((unsigned)ceil_div(g.X, TX), (unsigned)ceil_div(g.Xp, XC), (unsigned)g.Za);
    BH_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "deskew grid too large (%u,%u,%u)", grid.x, grid.y, grid.z);
    auto run = [&](auto kern) -> int {
        if (lds > 64 * 1024)
            BH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, grid, dim3(NT), lds, ctx->stream, in, out, g);
        BH_CHECK_HIP(hipGetLastError());
        return BH_OK;
    };
    switch (g.N) {
        case 1: return run(deskew_kernel<TIN, TX, NT, 1>);
        case 2: return run(deskew_kernel<TIN, TX, NT, 2>);
        case 3: return run(deskew_kernel<TIN, TX, NT, 3>);
        case 4: return run(deskew_kernel<TIN, TX, NT, 4>);
        default: return run(deskew_kernel<TIN, TX, NT, 0>);
    }
}

int fill_overhang_impl(bh_ctx* ctx, float* data, int64_t Z, int64_t Y, int64_t X, int fill_mode,
                       float fill_value, int iterations, float* mean_out);

}  // namespace bh

extern "C" {

int bh_deskew_shape(int64_t Z, int64_t Y, int64_t X, double ls_angle_deg, double px_to_scan_ratio,
                    int keep_overhang, int average_n_slices, double pixel_size_um, int64_t out_shape[3],
                    double voxel_size[3]) {
    BH_REQUIRE(out_shape != nullptr && voxel_size != nullptr, "NULL output argument");
    BH_REQUIRE(Z > 0 && Y > 0 && X > 0, "raw_data_shape must be positive, got (%lld,%lld,%lld)", (long long)Z,
               (long long)Y, (long long)X);
    BH_REQUIRE(average_n_slices >= 1, "average_n_slices must be >= 1, got %d", average_n_slices);
    BH_REQUIRE(px_to_scan_ratio > 0, "px_to_scan_ratio must be > 0, got %g", px_to_scan_ratio);
    const double theta = ls_angle_deg * M_PI / 180.0;
    const double st = std::sin(theta), ct = std::cos(theta);
    long long Xp;
    if (keep_overhang) {
        Xp = (long long)std::ceil(((double)Z / px_to_scan_ratio) + ((double)Y * ct));
    } else {
        Xp = (long long)std::ceil(((double)Z / px_to_scan_ratio) - ((double)Y * ct));
        // message text mirrors biahub/deskew.py:263-267
        BH_REQUIRE(Xp > 0,
                   "Dataset contains only overhang when keep_overhang=False. Computed Xp=%lld <= 0. Either set "
                   "keep_overhang=True or use a dataset with non-overhang content.",
                   Xp);
    }
    out_shape[0] = (Y + average_n_slices - 1) / average_n_slices;
    out_shape[1] = X;
    out_shape[2] = Xp;
    voxel_size[0] = average_n_slices * st * pixel_size_um;
    voxel_size[1] = pixel_size_um;
    voxel_size[2] = pixel_size_um;
    return BH_OK;
}

int bh_deskew(bh_ctx* ctx, const void* in, int in_dtype, int64_t Z, int64_t Y, int64_t X, double ls_angle_deg,
              double px_to_scan_ratio, int keep_overhang, int average_n_slices, int fill_mode, float fill_value,
              float* out, float* mean_out) {
    BH_REQUIRE(ctx != nullptr && in != nullptr && out != nullptr, "NULL argument");
    BH_REQUIRE(Z >= 2, "deskew needs at least 2 scan slices, got Z=%lld", (long long)Z);
    BH_REQUIRE(Z < (1 << 24) && Y < (1 << 24) && X < (1ll << 31), "volume too large for float32 coordinates");
    BH_REQUIRE(fill_mode >= BH_FILL_NONE && fill_mode <= BH_FILL_MEAN, "unknown fill_mode %d", fill_mode);
    bh::DeskewGeom g;
    int64_t os[3];
    BH_TRY(bh::deskew_geometry(Z, Y, X, ls_angle_deg, px_to_scan_ratio, keep_overhang, average_n_slices, &g, os));
    BH_REQUIRE(os[2] < (1 << 24), "deskewed X extent %lld too large", (long long)os[2]);
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    {
        bh::ScopedTimer t(ctx, bh::T_DESKEW);
        switch (in_dtype) {
            case BH_DT_F32: BH_TRY(bh::launch_deskew(ctx, (const float*)in, out, g)); break;
            case BH_DT_U16: BH_TRY(bh::launch_deskew(ctx, (const uint16_t*)in, out, g)); break;
            case BH_DT_U8: BH_TRY(bh::launch_deskew(ctx, (const uint8_t*)in, out, g)); break;
            case BH_DT_I16: BH_TRY(bh::launch_deskew(ctx, (const int16_t*)in, out, g)); break;
            default: BH_REQUIRE(false, "unsupported input dtype code %d", in_dtype);
        }
    }
    // reference :538 — fill only when keep_overhang and (fill == "mean" or fill != 0)
    const bool do_fill = keep_overhang && (fill_mode == BH_FILL_MEAN || (fill_mode == BH_FILL_CONSTANT && fill_value != 0.0f));
    if (do_fill) {
        BH_TRY(bh::fill_overhang_impl(ctx, out, os[0], os[1], os[2], fill_mode, fill_value, 3, mean_out));
    } else if (mean_out) {
        *mean_out = 0.0f;
    }
    return BH_OK;
}

}  // extern "C"
