// Host (CPU) form of the deskew operator: BASELINE config 1 — `biahub deskew ... --cluster debug` with the default
// `device: cpu` (biahub/settings.py:348-383, biahub/deskew.py:762-766) — runs without a GPU.  Own C++ behind the same C-ABI
// (bh_host_deskew: host pointers, no context), never the test oracle: the same float32 sample positions and the same per-output
// operation order as csrc/deskew.hip (deskew_ix, one fused multiply-add per tap pair, slices summed in order, the corrected
// multiply by 1/N), so its results are bit-identical to the GPU kernels'; the overhang fill is the 26-connected dilation of
// the exact-zero mask by `iterations` voxels (three box passes) and the float64 mean of what is left
// (biahub/deskew.py:339-368).  Planes of the output are dealt to std::threads.
#include "common.hpp"

#include <algorithm>
#include <cmath>
#include <thread>
#include <vector>

namespace bh {

static inline float host_deskew_ix(float px, float pxct, float offset, float zm1, int xo, int zo) {
#pragma clang fp contract(off)
    const float t1 = px * (float)xo;
    const float t2 = pxct * (float)zo;
    const float in_z = (t1 - t2) + offset;
    const float g = (2.0f * in_z) / zm1 - 1.0f;
    return ((g + 1.0f) / 2.0f) * zm1;
}

template <typename F>
static void parallel_for(int64_t n, int nthreads, F&& body) {
    nthreads = (int)std::max<int64_t>(1, std::min<int64_t>(nthreads, n));
    if (nthreads == 1) {
        for (int64_t i = 0; i < n; ++i) body(i);
        return;
    }
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; ++t)
        pool.emplace_back([&, t]() {
            for (int64_t i = t; i < n; i += nthreads) body(i);
        });
    for (auto& th : pool) th.join();
}

template <typename TIN>
static void host_deskew_planes(const TIN* in, float* out, int64_t Z, int64_t Y, int64_t X, int64_t Za, int64_t Xp, int N,
                               float px, float pxct, float offset, float zm1, int nthreads) {
#pragma clang fp contract(off)
    const float fN = (float)N, rN = 1.0f / fN;
    const int64_t plane = Y * X;
    parallel_for(Za, nthreads, [&](int64_t a) {
        std::vector<int> i0((size_t)N * Xp);
        std::vector<float> w0((size_t)N * Xp), w1((size_t)N * Xp);
        for (int k = 0; k < N; ++k)
            for (int64_t xo = 0; xo < Xp; ++xo) {
                const float ix = host_deskew_ix(px, pxct, offset, zm1, (int)xo, (int)(a * N + k));
                const float fl = std::floor(ix);
                i0[(size_t)k * Xp + xo] = (int)fl;
                w1[(size_t)k * Xp + xo] = ix - fl;
                w0[(size_t)k * Xp + xo] = (fl + 1.0f) - ix;
            }
        for (int64_t yo = 0; yo < X; ++yo) {
            const int64_t x = X - 1 - yo;
            float* orow = out + (a * X + yo) * Xp;
            for (int64_t xo = 0; xo < Xp; ++xo) {
                float s = 0.0f;
                for (int k = 0; k < N; ++k) {
                    const int64_t yin = Y - 1 - std::min<int64_t>(a * N + k, Y - 1);
                    const TIN* col = in + yin * X + x;
                    const int z0 = i0[(size_t)k * Xp + xo];
                    const float v0 = (z0 >= 0 && z0 < Z) ? (float)col[(int64_t)z0 * plane] : 0.0f;
                    const float v1 = (z0 + 1 >= 0 && z0 + 1 < Z) ? (float)col[(int64_t)(z0 + 1) * plane] : 0.0f;
                    const float val = std::fmaf(v1, w1[(size_t)k * Xp + xo], v0 * w0[(size_t)k * Xp + xo]);
                    s = k == 0 ? val : s + val;
                }
                if (N > 1) {
                    const float q = s * rN;
                    const float r = std::fmaf(-q, fN, s);
                    s = std::fmaf(r, rN, q);
                }
                orow[xo] = s;
            }
        }
    });
}

// dilation of a byte mask by `r` along one axis (stride / length given), in place via a scratch line per call
static void dilate_axis(std::vector<uint8_t>& m, int64_t n_lines_outer, int64_t n_lines_inner, int64_t len, int64_t stride,
                        int64_t outer_stride, int64_t inner_stride, int r, int nthreads) {
    parallel_for(n_lines_outer, nthreads, [&](int64_t o) {
        std::vector<uint8_t> line((size_t)len);
        for (int64_t i = 0; i < n_lines_inner; ++i) {
            uint8_t* p = m.data() + o * outer_stride + i * inner_stride;
            for (int64_t t = 0; t < len; ++t) line[(size_t)t] = p[t * stride];
            for (int64_t t = 0; t < len; ++t) {
                uint8_t v = 0;
                for (int64_t d = std::max<int64_t>(0, t - r); d <= std::min<int64_t>(len - 1, t + r); ++d) v |= line[(size_t)d];
                p[t * stride] = v;
            }
        }
    });
}

static void host_fill_overhang(float* data, int64_t Z, int64_t Y, int64_t X, int fill_mode, float fill_value, int iterations,
                               int nthreads, float* mean_out) {
    const int64_t V = Z * Y * X;
    std::vector<uint8_t> mask((size_t)V);
    for (int64_t i = 0; i < V; ++i) mask[(size_t)i] = data[i] == 0.0f;
    if (iterations > 0) {
        dilate_axis(mask, Z * Y, 1, X, 1, X, 0, iterations, nthreads);      // along x: one line per (z, y)
        dilate_axis(mask, Z, X, Y, X, Y * X, 1, iterations, nthreads);      // along y: lines (z, x)
        dilate_axis(mask, Y, X, Z, Y * X, X, 1, iterations, nthreads);      // along z: lines (y, x)
    }
    float fill = fill_value;
    if (fill_mode == BH_FILL_MEAN) {
        double s = 0.0;
        int64_t n = 0;
        for (int64_t i = 0; i < V; ++i)
            if (!mask[(size_t)i]) {
                s += (double)data[i];
                ++n;
            }
        fill = (float)(s / (double)n);  // 0 / 0 -> NaN like torch's empty mean
    }
    for (int64_t i = 0; i < V; ++i)
        if (mask[(size_t)i]) data[i] = fill;
    if (mean_out) *mean_out = fill;
}

}  // namespace bh

extern "C" int bh_host_deskew(const void* in, int in_dtype, int64_t Z, int64_t Y, int64_t X, double ls_angle_deg,
                              double px_to_scan_ratio, int keep_overhang, int average_n_slices, int fill_mode, float fill_value,
                              float* out, float* mean_out, int nthreads) {
    using namespace bh;
    BH_REQUIRE(in != nullptr && out != nullptr, "NULL argument");
    BH_REQUIRE(Z >= 2, "deskew needs at least 2 scan slices, got Z=%lld", (long long)Z);
    BH_REQUIRE(Z < (1 << 24) && Y < (1 << 24) && X < (1ll << 31), "volume too large for float32 coordinates");
    BH_REQUIRE(fill_mode >= BH_FILL_NONE && fill_mode <= BH_FILL_MEAN, "unknown fill_mode %d", fill_mode);
    int64_t os[3];
    double voxel[3];
    BH_TRY(bh_deskew_shape(Z, Y, X, ls_angle_deg, px_to_scan_ratio, keep_overhang, average_n_slices, 1.0, os, voxel));
    // the geometry of csrc/deskew.hip: deskew_geometry
    const double ct = std::cos(ls_angle_deg * M_PI / 180.0), px = px_to_scan_ratio;
    const double offset = px * ct * (double)(Y - 1) / 2 - px * (double)(os[2] - 1) / 2 + (double)(Z - 1) / 2;
    const float fpx = (float)px, fpxct = (float)(px * ct), foff = (float)offset, zm1 = (float)(Z - 1);
    if (nthreads <= 0) nthreads = (int)std::max(1u, std::thread::hardware_concurrency());
    const int N = average_n_slices;
    switch (in_dtype) {
        case BH_DT_F32: host_deskew_planes((const float*)in, out, Z, Y, X, os[0], os[2], N, fpx, fpxct, foff, zm1, nthreads); break;
        case BH_DT_U16: host_deskew_planes((const uint16_t*)in, out, Z, Y, X, os[0], os[2], N, fpx, fpxct, foff, zm1, nthreads); break;
        case BH_DT_U8: host_deskew_planes((const uint8_t*)in, out, Z, Y, X, os[0], os[2], N, fpx, fpxct, foff, zm1, nthreads); break;
        case BH_DT_I16: host_deskew_planes((const int16_t*)in, out, Z, Y, X, os[0], os[2], N, fpx, fpxct, foff, zm1, nthreads); break;
        default: BH_REQUIRE(false, "unsupported input dtype code %d", in_dtype);
    }
    const bool do_fill = keep_overhang && (fill_mode == BH_FILL_MEAN || (fill_mode == BH_FILL_CONSTANT && fill_value != 0.0f));
    if (do_fill) host_fill_overhang(out, os[0], os[1], os[2], fill_mode, fill_value, 3, nthreads, mean_out);
    else if (mean_out) *mean_out = 0.0f;
    return BH_OK;
}
