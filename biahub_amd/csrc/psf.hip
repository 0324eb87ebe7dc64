// Bead detection and PSF averaging on gfx950.
//
//   bh_block_peaks      detect_peaks' dense part (biahub/characterize_psf.py:622-650): F.avg_pool3d(k, stride 1, pad k/2,
//                       count_include_pad=False) fused with F.max_pool3d(block, stride block, pad block/2,
//                       return_indices=True) — one candidate (value, flat index) per block, never materialising the
//                       blurred volume.  The rest of detect_peaks (top-k, threshold, NMS on <= 2000 points) is host logic.
//   bh_patch_peaks      BeadExtractor._find_closest_peak (vendor/napari_psf_analysis/.../BeadExtractor.py:61-78): argmax of
//                       the sigma = 2 Gaussian-smoothed crop (zero outside the crop, radius 8), per bead.
//   bh_average_patches  estimate_psf_cli's reduction (biahub/estimate_psf.py:104-112): mean over beads of patch / max(patch),
//                       then (avg - min) / max.
//
// Summation orders follow torch's pooling kernels (z, y, x nested, float accumulate; first maximum in scan order wins), so
// the block peaks are bit-identical to the reference's on CPU torch (golden vectors).
#include "common.hpp"

#include <algorithm>
#include <cmath>

namespace bh {

struct PeakParams {
    int Z, Y, X;
    int k;           // blur kernel (odd)
    int bz, by, bx;  // block
    int nz, ny, nx;  // blocks per axis
};

__device__ __forceinline__ float box_mean(const float* __restrict__ in, const PeakParams& p, int z, int y, int x) {
    const int r = p.k >> 1;
    const int z0 = max(z - r, 0), z1 = min(z + r, p.Z - 1), y0 = max(y - r, 0), y1 = min(y + r, p.Y - 1),
              x0 = max(x - r, 0), x1 = min(x + r, p.X - 1);
    float s = 0.0f;
    for (int zz = z0; zz <= z1; ++zz)
        for (int yy = y0; yy <= y1; ++yy) {
            const float* row = in + ((size_t)zz * p.Y + yy) * p.X;
            for (int xx = x0; xx <= x1; ++xx) s += row[xx];
        }
    return s / (float)((z1 - z0 + 1) * (y1 - y0 + 1) * (x1 - x0 + 1));
}

// one workgroup per pooling block; lanes along x, every thread scans its voxels in increasing flat order
__global__ __launch_bounds__(256) void block_peaks_kernel(const float* __restrict__ in, PeakParams p,
                                                          float* __restrict__ values, long long* __restrict__ indices) {
    const int b = blockIdx.x;
    const int ox = b % p.nx, oy = (b / p.nx) % p.ny, oz = b / (p.nx * p.ny);
    const int z0 = max(oz * p.bz - p.bz / 2, 0), z1 = min(oz * p.bz - p.bz / 2 + p.bz, p.Z);
    const int y0 = max(oy * p.by - p.by / 2, 0), y1 = min(oy * p.by - p.by / 2 + p.by, p.Y);
    const int x0 = max(ox * p.bx - p.bx / 2, 0), x1 = min(ox * p.bx - p.bx / 2 + p.bx, p.X);
    const int wx = x1 - x0, wy = y1 - y0, wz = z1 - z0;
    float best = -INFINITY;
    long long bidx = 0x7fffffffffffffffll;
    const long long n = (long long)wx * wy * wz;
    for (long long i = threadIdx.x; i < n; i += 256) {
        const int x = x0 + (int)(i % wx);
        const long long t = i / wx;
        const int y = y0 + (int)(t % wy), z = z0 + (int)(t / wy);
        const float v = box_mean(in, p, z, y, x);
        const long long idx = ((long long)z * p.Y + y) * p.X + x;
        if (v > best || (v == best && idx < bidx)) best = v, bidx = idx;
    }
    __shared__ float sv[256];
    __shared__ long long si[256];
    sv[threadIdx.x] = best;
    si[threadIdx.x] = bidx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const float v = sv[threadIdx.x + o];
            const long long j = si[threadIdx.x + o];
            if (v > sv[threadIdx.x] || (v == sv[threadIdx.x] && j < si[threadIdx.x])) sv[threadIdx.x] = v, si[threadIdx.x] = j;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        values[b] = sv[0];
        indices[b] = si[0] == 0x7fffffffffffffffll ? 0 : si[0];
    }
}

// ------------------------------------------------------------------------------------------------ per-bead patches
struct PatchParams {
    int Z, Y, X;
    int pz, py, px;  // patch extent
    int R;
    double w[17];    // Gaussian taps 0..R (symmetric)
};

// separable Gaussian along `axis` over a batch of patches; zero outside the patch (scipy mode="constant").
// pass 0 reads the volume at the patch origin, later passes read the previous temporary.
__global__ __launch_bounds__(256) void patch_gauss_kernel(const float* __restrict__ vol, const float* __restrict__ src,
                                                          float* __restrict__ dst, const int* __restrict__ starts, int nb,
                                                          PatchParams p, int axis) {
    const long long pv = (long long)p.pz * p.py * p.px, total = pv * nb;
    const int n = axis == 0 ? p.pz : (axis == 1 ? p.py : p.px);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / pv);
        const long long r = i - (long long)b * pv;
        const int x = (int)(r % p.px), y = (int)((r / p.px) % p.py), z = (int)(r / ((long long)p.px * p.py));
        const int c = axis == 0 ? z : (axis == 1 ? y : x);
        auto at = [&](int j) -> double {
            if (j < 0 || j >= n) return 0.0;
            const int zz = axis == 0 ? j : z, yy = axis == 1 ? j : y, xx = axis == 2 ? j : x;
            if (src) return (double)src[(long long)b * pv + ((long long)zz * p.py + yy) * p.px + xx];
            return (double)vol[((size_t)(starts[3 * b] + zz) * p.Y + (starts[3 * b + 1] + yy)) * p.X + starts[3 * b + 2] + xx];
        };
        double acc = p.w[0] * at(c);
        for (int k = 1; k <= p.R; ++k) acc += (at(c - k) + at(c + k)) * p.w[k];
        dst[i] = (float)acc;
    }
}

// first maximum (flat order) of every patch of a batch -> peaks[b] = flat index inside the patch
__global__ __launch_bounds__(256) void patch_argmax_kernel(const float* __restrict__ src, long long pv,
                                                           long long* __restrict__ peaks) {
    const float* s = src + (long long)blockIdx.x * pv;
    float best = -INFINITY;
    long long bidx = 0x7fffffffffffffffll;
    for (long long i = threadIdx.x; i < pv; i += 256) {
        const float v = s[i];
        if (v > best) best = v, bidx = i;
    }
    __shared__ float sv[256];
    __shared__ long long si[256];
    sv[threadIdx.x] = best;
    si[threadIdx.x] = bidx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const float v = sv[threadIdx.x + o];
            const long long j = si[threadIdx.x + o];
            if (v > sv[threadIdx.x] || (v == sv[threadIdx.x] && j < si[threadIdx.x])) sv[threadIdx.x] = v, si[threadIdx.x] = j;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) peaks[blockIdx.x] = si[0] == 0x7fffffffffffffffll ? 0 : si[0];
}

__global__ __launch_bounds__(256) void patch_max_kernel(const float* __restrict__ vol, const int* __restrict__ starts,
                                                        PatchParams p, float* __restrict__ maxes) {
    const int b = blockIdx.x;
    const long long pv = (long long)p.pz * p.py * p.px;
    float m = -INFINITY;
    for (long long r = threadIdx.x; r < pv; r += 256) {
        const int x = (int)(r % p.px), y = (int)((r / p.px) % p.py), z = (int)(r / ((long long)p.px * p.py));
        m = fmaxf(m, vol[((size_t)(starts[3 * b] + z) * p.Y + (starts[3 * b + 1] + y)) * p.X + starts[3 * b + 2] + x]);
    }
    __shared__ float sv[256];
    sv[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sv[threadIdx.x] = fmaxf(sv[threadIdx.x], sv[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) maxes[b] = sv[0];
}

// out[v] = mean_b patch_b[v] / max_b : one thread per patch voxel, beads in order (float64 accumulate)
__global__ __launch_bounds__(256) void patch_mean_kernel(const float* __restrict__ vol, const int* __restrict__ starts,
                                                         const float* __restrict__ maxes, int nb, PatchParams p,
                                                         float* __restrict__ out) {
    const long long pv = (long long)p.pz * p.py * p.px;
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= pv) return;
    const int x = (int)(r % p.px), y = (int)((r / p.px) % p.py), z = (int)(r / ((long long)p.px * p.py));
    double acc = 0.0;
    for (int b = 0; b < nb; ++b)
        acc += (double)(vol[((size_t)(starts[3 * b] + z) * p.Y + (starts[3 * b + 1] + y)) * p.X + starts[3 * b + 2] + x] / maxes[b]);
    out[r] = (float)(acc / (double)nb);
}

// avg -= min(avg); avg /= max(avg)   (estimate_psf.py:110-112), one workgroup
__global__ __launch_bounds__(1024) void psf_normalise_kernel(float* __restrict__ a, long long n) {
    __shared__ float smin[1024], smax[1024];
    float mn = INFINITY, mx = -INFINITY;
    for (long long i = threadIdx.x; i < n; i += 1024) mn = fminf(mn, a[i]), mx = fmaxf(mx, a[i]);
    smin[threadIdx.x] = mn, smax[threadIdx.x] = mx;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + o]);
            smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + o]);
        }
        __syncthreads();
    }
    const float lo = smin[0], hi = smax[0] - smin[0];
    for (long long i = threadIdx.x; i < n; i += 1024) a[i] = (a[i] - lo) / hi;
}

static bool patches_inside(const int* starts, int n, const int patch[3], int64_t Z, int64_t Y, int64_t X) {
    for (int b = 0; b < n; ++b)
        if (starts[3 * b] < 0 || starts[3 * b + 1] < 0 || starts[3 * b + 2] < 0 || starts[3 * b] + patch[0] > Z ||
            starts[3 * b + 1] + patch[1] > Y || starts[3 * b + 2] + patch[2] > X)
            return false;
    return true;
}

}  // namespace bh

using namespace bh;

extern "C" int bh_block_peaks(bh_ctx* ctx, const float* in, int64_t Z, int64_t Y, int64_t X, int blur_kernel_size,
                              const int block[3], float* values, int64_t* indices, int64_t nblocks[3]) {
    BH_REQUIRE(block && nblocks, "null argument");
    BH_REQUIRE(Z > 0 && Y > 0 && X > 0 && Z < (1ll << 31) && Y < (1ll << 31) && X < (1ll << 31), "bad shape");
    BH_REQUIRE(blur_kernel_size >= 1 && (blur_kernel_size & 1), "kernel_size=%d must be an odd number", blur_kernel_size);
    const int64_t N[3] = {Z, Y, X};
    for (int a = 0; a < 3; ++a) {
        BH_REQUIRE(block[a] >= 1, "block size must be >= 1");
        const int64_t span = N[a] + 2 * (block[a] / 2) - block[a];
        BH_REQUIRE(span >= 0, "block larger than the padded volume");
        nblocks[a] = span / block[a] + 1;
    }
    if (!values && !indices) return BH_OK;  // geometry query
    BH_REQUIRE(ctx && in && values && indices, "null argument");
    const int64_t nb = nblocks[0] * nblocks[1] * nblocks[2];
    BH_REQUIRE(nb < (1ll << 31), "too many blocks");
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    PeakParams p{(int)Z, (int)Y, (int)X, blur_kernel_size, block[0], block[1], block[2], (int)nblocks[0], (int)nblocks[1],
                 (int)nblocks[2]};
    hipLaunchKernelGGL(block_peaks_kernel, dim3((unsigned)nb), dim3(256), 0, ctx->stream, in, p, values,
                       reinterpret_cast<long long*>(indices));
    BH_CHECK_HIP(hipGetLastError());
    return BH_OK;
}

static int fill_patch_params(PatchParams& p, int64_t Z, int64_t Y, int64_t X, const int patch[3], double sigma) {
    p.Z = (int)Z, p.Y = (int)Y, p.X = (int)X;
    p.pz = patch[0], p.py = patch[1], p.px = patch[2];
    p.R = sigma > 0 ? (int)(4.0 * sigma + 0.5) : 0;  // scipy: truncate 4.0
    BH_REQUIRE(p.R <= 16, "sigma too large (radius %d > 16)", p.R);
    double s = 0;
    for (int k = -p.R; k <= p.R; ++k) s += sigma > 0 ? std::exp(-0.5 * k * k / (sigma * sigma)) : 1.0;
    for (int k = 0; k <= p.R; ++k) p.w[k] = (sigma > 0 ? std::exp(-0.5 * k * k / (sigma * sigma)) : 1.0) / s;
    return BH_OK;
}

extern "C" int bh_patch_peaks(bh_ctx* ctx, const float* in, int64_t Z, int64_t Y, int64_t X, const int* starts, int n,
                              const int patch[3], double sigma, int64_t* peaks) {
    BH_REQUIRE(ctx && in && starts && patch && peaks && n >= 0, "null argument");
    BH_REQUIRE(Z > 0 && Y > 0 && X > 0 && Z < (1ll << 31) && Y < (1ll << 31) && X < (1ll << 31), "bad shape");
    BH_REQUIRE(patch[0] > 0 && patch[1] > 0 && patch[2] > 0, "bad patch size");
    BH_REQUIRE(patches_inside(starts, n, patch, Z, Y, X), "a patch is not fully inside the volume");
    if (n == 0) return BH_OK;
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    PatchParams p;
    BH_TRY(fill_patch_params(p, Z, Y, X, patch, sigma));
    const long long pv = (long long)patch[0] * patch[1] * patch[2];
    const int batch = (int)std::max<long long>(1, std::min<long long>(n, (256ll << 20) / (pv * 4)));  // <= 256 MiB per temp
    float *t1, *t2;
    int* dstarts;
    long long* dpeaks;
    BH_TRY(get_scratch(ctx, "psf_t1", sizeof(float) * (size_t)pv * batch, (void**)&t1));
    BH_TRY(get_scratch(ctx, "psf_t2", sizeof(float) * (size_t)pv * batch, (void**)&t2));
    const size_t off = (sizeof(int) * 3 * (size_t)n + 7) & ~(size_t)7;
    BH_TRY(get_scratch(ctx, "psf_starts", off + sizeof(long long) * (size_t)n, (void**)&dstarts));
    dpeaks = reinterpret_cast<long long*>(reinterpret_cast<char*>(dstarts) + off);
    BH_CHECK_HIP(hipMemcpyAsync(dstarts, starts, sizeof(int) * 3 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    for (int b0 = 0; b0 < n; b0 += batch) {
        const int nb = std::min(batch, n - b0);
        const int grid = (int)std::min<long long>(ceil_div(pv * nb, 256), (long long)ctx->num_cus * 32);
        // scipy.ndimage.gaussian_filter order: axis 0, 1, 2, each pass rounded to float32
        hipLaunchKernelGGL(patch_gauss_kernel, dim3(grid), dim3(256), 0, ctx->stream, in, (const float*)nullptr, t1,
                           dstarts + 3 * b0, nb, p, 0);
        hipLaunchKernelGGL(patch_gauss_kernel, dim3(grid), dim3(256), 0, ctx->stream, in, (const float*)t1, t2,
                           dstarts + 3 * b0, nb, p, 1);
        hipLaunchKernelGGL(patch_gauss_kernel, dim3(grid), dim3(256), 0, ctx->stream, in, (const float*)t2, t1,
                           dstarts + 3 * b0, nb, p, 2);
        hipLaunchKernelGGL(patch_argmax_kernel, dim3(nb), dim3(256), 0, ctx->stream, (const float*)t1, pv, dpeaks + b0);
        BH_CHECK_HIP(hipGetLastError());
    }
    BH_CHECK_HIP(hipMemcpyAsync(peaks, dpeaks, sizeof(long long) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    BH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return BH_OK;
}

extern "C" int bh_average_patches(bh_ctx* ctx, const float* in, int64_t Z, int64_t Y, int64_t X, const int* starts, int n,
                                  const int patch[3], int normalise, float* out) {
    BH_REQUIRE(ctx && in && starts && patch && out, "null argument");
    BH_REQUIRE(n >= 1, "no beads to average");
    BH_REQUIRE(Z > 0 && Y > 0 && X > 0 && Z < (1ll << 31) && Y < (1ll << 31) && X < (1ll << 31), "bad shape");
    BH_REQUIRE(patch[0] > 0 && patch[1] > 0 && patch[2] > 0, "bad patch size");
    BH_REQUIRE(patches_inside(starts, n, patch, Z, Y, X), "a patch is not fully inside the volume");
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    PatchParams p;
    BH_TRY(fill_patch_params(p, Z, Y, X, patch, 0.0));
    const long long pv = (long long)patch[0] * patch[1] * patch[2];
    int* dstarts;
    const size_t off = (sizeof(int) * 3 * (size_t)n + 7) & ~(size_t)7;
    BH_TRY(get_scratch(ctx, "psf_starts", off + sizeof(long long) * (size_t)n, (void**)&dstarts));
    float* maxes = reinterpret_cast<float*>(reinterpret_cast<char*>(dstarts) + off);
    BH_CHECK_HIP(hipMemcpyAsync(dstarts, starts, sizeof(int) * 3 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(patch_max_kernel, dim3(n), dim3(256), 0, ctx->stream, in, (const int*)dstarts, p, maxes);
    hipLaunchKernelGGL(patch_mean_kernel, dim3((unsigned)ceil_div(pv, 256)), dim3(256), 0, ctx->stream, in, (const int*)dstarts,
                       (const float*)maxes, n, p, out);
    if (normalise) hipLaunchKernelGGL(psf_normalise_kernel, dim3(1), dim3(1024), 0, ctx->stream, out, pv);
    BH_CHECK_HIP(hipGetLastError());
    BH_CHECK_HIP(hipStreamSynchronize(ctx->stream));  // `starts` staging is reused by the next call
    return BH_OK;
}
