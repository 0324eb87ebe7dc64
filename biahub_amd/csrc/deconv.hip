// FFT deconvolution on gfx950: transfer function (C1), Tikhonov inverse filter (C2) and
// Richardson-Lucy (C3).  3-D real FFTs go through hipFFT (R2C / C2R, half spectrum); every
// pointwise step between them is a fused HIP kernel here.
//
//   C1  biahub/deconvolve.py:30-43   H = |fftn(pad(psf))| / max          -> bh_transfer_function
//   C2  biahub/deconvolve.py:46-66   real(ifftn(fftn(x) * H / (H^2 + reg))) (H real >= 0)
//   C3  north-star extension         Richardson-Lucy, circular boundary
//
// All spectra are Hermitian halves (Z, Y, X/2+1) of complex64; H is real and even, so the
// half-spectrum filter is exactly the reference's full-spectrum product.
#include "common.hpp"

#include <cstdlib>
#include <cstring>

namespace bh {

typedef float2 cf;

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// sum of the PSF in double (wavefront reduction -> one partial per block -> host-free finalize)
__global__ __launch_bounds__(256) void psf_sum_kernel(const float* __restrict__ psf, int64_t n, double* out) {
    __shared__ double sh[4];
    double s = 0;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += (double)psf[i];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *out = sh[0] + sh[1] + sh[2] + sh[3];
}

// scatter the (optionally normalised, optionally origin-centred) PSF into a zeroed volume.
// before[] = leading zero-pad per axis (deconvolve.py:34-35: odd remainder goes to the END);
// roll[]   = circular shift applied to the destination index (0 for C1).
__global__ void place_psf_kernel(const float* __restrict__ psf, float* __restrict__ vol, int pz, int py, int px,
                                 int64_t Z, int64_t Y, int64_t X, int bz, int by, int bx, int rz, int ry, int rx,
                                 const double* __restrict__ psf_sum) {
    const int64_t n = (int64_t)pz * py * px;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % px), y = (int)((i / px) % py), z = (int)(i / ((int64_t)px * py));
        int64_t dz = (z + bz - rz) % Z, dy = (y + by - ry) % Y, dx = (x + bx - rx) % X;
        if (dz < 0) dz += Z;
        if (dy < 0) dy += Y;
        if (dx < 0) dx += X;
        float v = psf[i];
        if (psf_sum) {
            // normalise in double like the oracle: float(psf / sum)
            v = (float)((double)v / *psf_sum);
        }
        vol[(dz * Y + dy) * X + dx] = v;
    }
}

// |S| in place into a float array + global max (non-negative floats order like their bit patterns)
__global__ __launch_bounds__(256) void abs_max_kernel(const cf* __restrict__ spec, float* __restrict__ mag, int64_t n,
                                                      unsigned int* gmax) {
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const cf c = spec[i];
        const float a = hypotf(c.x, c.y);
        mag[i] = a;
        m = fmaxf(m, a);
    }
    __shared__ float sh[4];
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
        atomicMax(gmax, __float_as_uint(m));
    }
}

// expand the half-spectrum magnitude to the reference's full (Z,Y,X) array, dividing by the max
__global__ void tf_expand_kernel(const float* __restrict__ mag, float* __restrict__ tf, int64_t Z, int64_t Y,
                                 int64_t X, const unsigned int* gmax) {
    const int64_t Xh = X / 2 + 1;
    const int64_t n = Z * Y * X;
    const float mx = __uint_as_float(*gmax);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t x = i % X, y = (i / X) % Y, z = i / (X * Y);
        float v;
        if (x < Xh) {
            v = mag[(z * Y + y) * Xh + x];
        } else {  // Hermitian mirror: |F(k)| = |F(-k)|
            const int64_t zz = z ? Z - z : 0, yy = y ? Y - y : 0;
            v = mag[(zz * Y + yy) * Xh + (X - x)];
        }
        tf[i] = v / mx;
    }
}

// S *= H / (H^2 + reg) / V   with H read from the FULL-spectrum float array the reference passes
__global__ void tikhonov_filter_kernel(cf* __restrict__ spec, const float* __restrict__ tf, int64_t Z, int64_t Y,
                                       int64_t X, float reg, float inv_v) {
    const int64_t Xh = X / 2 + 1;
    const int64_t n = Z * Y * Xh;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t x = i % Xh;
        const int64_t zy = i / Xh;
        const float h = tf[zy * X + x];
        const float f = (h / (h * h + reg)) * inv_v;
        cf c = spec[i];
        c.x *= f;
        c.y *= f;
        spec[i] = c;
    }
}

// real parts of a complex array (the real transfer function of a point-symmetric PSF)
__global__ void real_part_kernel(const cf* __restrict__ a, float* __restrict__ re, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) re[i] = a[i].x;
}

// ---- Richardson-Lucy pointwise kernels (float4 / 2x complex per lane) ---------------------
__global__ void scale_spectrum_kernel(cf* __restrict__ s, int64_t n, float f) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        cf c = s[i];
        c.x *= f;
        c.y *= f;
        s[i] = c;
    }
}

template <bool CONJ>
__global__ __launch_bounds__(256) void cmul_kernel(cf* __restrict__ s, const cf* __restrict__ otf, int64_t n2) {
    // n2 = number of complex PAIRS (float4); caller handles an odd tail element separately
    float4* s4 = reinterpret_cast<float4*>(s);
    const float4* o4 = reinterpret_cast<const float4*>(otf);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
        float4 a = s4[i];
        const float4 b = o4[i];
        float4 r;
        if (CONJ) {
            r.x = a.x * b.x + a.y * b.y;
            r.y = a.y * b.x - a.x * b.y;
            r.z = a.z * b.z + a.w * b.w;
            r.w = a.w * b.z - a.z * b.w;
        } else {
            r.x = a.x * b.x - a.y * b.y;
            r.y = a.x * b.y + a.y * b.x;
            r.z = a.z * b.z - a.w * b.w;
            r.w = a.z * b.w + a.w * b.z;
        }
        s4[i] = r;
    }
}

template <bool CONJ>
__global__ void cmul_tail_kernel(cf* __restrict__ s, const cf* __restrict__ otf, int64_t i) {
    const cf a = s[i], b = otf[i];
    cf r;
    if (CONJ) {
        r.x = a.x * b.x + a.y * b.y;
        r.y = a.y * b.x - a.x * b.y;
    } else {
        r.x = a.x * b.x - a.y * b.y;
        r.y = a.x * b.y + a.y * b.x;
    }
    s[i] = r;
}

// blur <- d / max(blur, eps)
__global__ __launch_bounds__(256) void ratio_kernel(float* __restrict__ blur, const float* __restrict__ d, int64_t n,
                                                    float eps) {
    const int64_t n4 = n / 4;
    float4* b4 = reinterpret_cast<float4*>(blur);
    const float4* d4 = reinterpret_cast<const float4*>(d);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 b = b4[i];
        const float4 v = d4[i];
        b.x = v.x / fmaxf(b.x, eps);
        b.y = v.y / fmaxf(b.y, eps);
        b.z = v.z / fmaxf(b.z, eps);
        b.w = v.w / fmaxf(b.w, eps);
        b4[i] = b;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = n4 * 4 + threadIdx.x;
        blur[i] = d[i] / fmaxf(blur[i], eps);
    }
}

// est <- max(est * corr, 0)
__global__ __launch_bounds__(256) void update_kernel(float* __restrict__ est, const float* __restrict__ corr, int64_t n) {
    const int64_t n4 = n / 4;
    float4* e4 = reinterpret_cast<float4*>(est);
    const float4* c4 = reinterpret_cast<const float4*>(corr);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 e = e4[i];
        const float4 c = c4[i];
        e.x = fmaxf(e.x * c.x, 0.0f);
        e.y = fmaxf(e.y * c.y, 0.0f);
        e.z = fmaxf(e.z * c.z, 0.0f);
        e.w = fmaxf(e.w * c.w, 0.0f);
        e4[i] = e;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = n4 * 4 + threadIdx.x;
        est[i] = fmaxf(est[i] * corr[i], 0.0f);
    }
}

// ---- pad / fold: exact circular convolution at an awkward size N through FFTs of a friendly size P >= N + K - 1 ----
// The zero-padded estimate convolved at size P is the LINEAR convolution; its samples that stick out of [0, N) by up to
// elo voxels below and ehi above are folded back (n + N and n - N) to give the circular result at size N.
struct FoldDims {
    int64_t N[3], P[3];
    int elo[3], ehi[3];  // reach of the kernel below index 0 / above index N-1
};

// dst (P-volume) <- zero-padded src (N-volume)
__global__ __launch_bounds__(256) void pad_volume_kernel(const float* __restrict__ src, float* __restrict__ dst, FoldDims f) {
    const int64_t total = f.P[0] * f.P[1] * f.P[2];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t x = i % f.P[2], y = (i / f.P[2]) % f.P[1], z = i / (f.P[2] * f.P[1]);
        dst[i] = (z < f.N[0] && y < f.N[1] && x < f.N[2]) ? src[(z * f.N[1] + y) * f.N[2] + x] : 0.0f;
    }
}

__device__ __forceinline__ float fold_at(const float* __restrict__ lin, const FoldDims& f, int64_t z, int64_t y, int64_t x) {
    // up to three source indices per axis: n, n + N (the part above N-1), P + n - N (the part below 0)
    int64_t iz[3], iy[3], ix[3];
    int nz = 0, ny = 0, nx = 0;
    iz[nz++] = z;
    if (z < f.ehi[0]) iz[nz++] = z + f.N[0];
    if (z >= f.N[0] - f.elo[0]) iz[nz++] = f.P[0] + z - f.N[0];
    iy[ny++] = y;
    if (y < f.ehi[1]) iy[ny++] = y + f.N[1];
    if (y >= f.N[1] - f.elo[1]) iy[ny++] = f.P[1] + y - f.N[1];
    ix[nx++] = x;
    if (x < f.ehi[2]) ix[nx++] = x + f.N[2];
    if (x >= f.N[2] - f.elo[2]) ix[nx++] = f.P[2] + x - f.N[2];
    float acc = 0.0f;
    for (int a = 0; a < nz; ++a)
        for (int b = 0; b < ny; ++b)
            for (int c = 0; c < nx; ++c) acc += lin[(iz[a] * f.P[1] + iy[b]) * f.P[2] + ix[c]];
    return acc;
}

// next (P-volume, zero outside the N box) <- d / max(fold(lin), eps)
__global__ __launch_bounds__(256) void fold_ratio_kernel(const float* __restrict__ lin, const float* __restrict__ d,
                                                         float* __restrict__ next, FoldDims f, float eps) {
    const int64_t total = f.P[0] * f.P[1] * f.P[2];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t x = i % f.P[2], y = (i / f.P[2]) % f.P[1], z = i / (f.P[2] * f.P[1]);
        float r = 0.0f;
        if (z < f.N[0] && y < f.N[1] && x < f.N[2])
            r = d[(z * f.N[1] + y) * f.N[2] + x] / fmaxf(fold_at(lin, f, z, y, x), eps);
        next[i] = r;
    }
}

// est (N-volume) <- max(est * fold(lin), 0)
__global__ __launch_bounds__(256) void fold_update_kernel(const float* __restrict__ lin, float* __restrict__ est, FoldDims f) {
    const int64_t total = f.N[0] * f.N[1] * f.N[2];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t x = i % f.N[2], y = (i / f.N[2]) % f.N[1], z = i / (f.N[2] * f.N[1]);
        est[i] = fmaxf(est[i] * fold_at(lin, f, z, y, x), 0.0f);
    }
}

// ---- wrap-padding for Richardson-Lucy on the fused engine at a power-of-two box (richardson_lucy_engine_padded) ----
// dst voxel t (box D) takes src voxel soff + wrap_N(t - doff) when t - doff lies within [-lo, N - 1 + hi] on every axis,
// else 0.  With an N-pitched source this builds the wrap-padded box; with a box-pitched source it re-wraps the margins
// from the interior; with D = N it crops.
struct RemapDims {
    int64_t D[3], S[3];  // destination box, source pitch box
    int N[3], doff[3], soff[3], lo[3], hi[3];
};

template <bool CLIP>
__global__ __launch_bounds__(256) void remap_kernel(const float* __restrict__ src, float* __restrict__ dst, RemapDims r) {
    const int64_t rows = r.D[0] * r.D[1];
    for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
        const int y = (int)(row % r.D[1]), z = (int)(row / r.D[1]);
        int qz = z - r.doff[0], qy = y - r.doff[1];
        const bool in_zy = qz >= -r.lo[0] && qz < r.N[0] + r.hi[0] && qy >= -r.lo[1] && qy < r.N[1] + r.hi[1];
        qz += qz < 0 ? r.N[0] : (qz >= r.N[0] ? -r.N[0] : 0);
        qy += qy < 0 ? r.N[1] : (qy >= r.N[1] ? -r.N[1] : 0);
        const float* srow = src + ((int64_t)(qz + r.soff[0]) * r.S[1] + (qy + r.soff[1])) * r.S[2] + r.soff[2];
        float* drow = dst + row * r.D[2];
        for (int x = blockIdx.x * 256 + threadIdx.x; x < r.D[2]; x += gridDim.x * 256) {
            int qx = x - r.doff[2];
            float v = 0.0f;
            if (in_zy && qx >= -r.lo[2] && qx < r.N[2] + r.hi[2]) {
                qx += qx < 0 ? r.N[2] : (qx >= r.N[2] ? -r.N[2] : 0);
                v = srow[qx];
                if (CLIP) v = fmaxf(v, 0.0f);
            }
            drow[x] = v;
        }
    }
}

// Update step of the padded engine path.  corr: linear correlation of the zero-padded ratio on the box P (volume voxel n at
// n + off); est_old / est_new: the estimate on the same box, wrap-extended by lo below and hi above the volume.  Every box
// voxel inside the extended volume gets  max(est_old[n] * fold(corr)[n], 0)  for the volume voxel n it stands for — fold
// adds the tails the linear correlation left below 0 and above N - 1 (n + N when n < hi, n - N when n >= N - lo; both
// inside the box because P >= N + lo + hi) — everything else 0.
struct FoldBox {
    int64_t P[3];
    int N[3], off[3], lo[3], hi[3];
};

__global__ __launch_bounds__(256) void fold_update_rewrap_kernel(const float* __restrict__ corr, const float* __restrict__ est_old,
                                                                 float* __restrict__ est_new, FoldBox f) {
    const int64_t rows = f.P[0] * f.P[1];
    for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
        const int y = (int)(row % f.P[1]), z = (int)(row / f.P[1]);
        int qz = z - f.off[0], qy = y - f.off[1];
        const bool in_zy = qz >= -f.lo[0] && qz < f.N[0] + f.hi[0] && qy >= -f.lo[1] && qy < f.N[1] + f.hi[1];
        qz += qz < 0 ? f.N[0] : (qz >= f.N[0] ? -f.N[0] : 0);
        qy += qy < 0 ? f.N[1] : (qy >= f.N[1] ? -f.N[1] : 0);
        // fold taps along z and y: the voxel itself and at most one wrapped tail (lo + hi < N)
        int tz[2] = {qz + f.off[0], 0}, ty[2] = {qy + f.off[1], 0};
        int nz = 1, ny = 1;
        if (qz < f.hi[0]) tz[nz++] = qz + f.N[0] + f.off[0];
        else if (qz >= f.N[0] - f.lo[0]) tz[nz++] = qz - f.N[0] + f.off[0];
        if (qy < f.hi[1]) ty[ny++] = qy + f.N[1] + f.off[1];
        else if (qy >= f.N[1] - f.lo[1]) ty[ny++] = qy - f.N[1] + f.off[1];
        const float* erow = est_old + ((int64_t)tz[0] * f.P[1] + ty[0]) * f.P[2] + f.off[2];
        float* drow = est_new + row * f.P[2];
        // four voxels per thread (the box's rows are multiples of 4 long); a group whose voxels are all their own volume
        // voxel, away from the x tails, on a row without z / y tails reads and writes whole float4s at its own position
        const bool plain_row = in_zy && nz == 1 && ny == 1;
        const float* crow0 = corr + ((int64_t)tz[0] * f.P[1] + ty[0]) * f.P[2];
        for (int x = 4 * (blockIdx.x * 256 + threadIdx.x); x < f.P[2]; x += 4 * gridDim.x * 256) {
            const int q0 = x - f.off[2];
            if (plain_row && q0 >= f.hi[2] && q0 + 3 < f.N[2] - f.lo[2]) {
                const float4 e = *reinterpret_cast<const float4*>(erow + q0);
                const float4 c = *reinterpret_cast<const float4*>(crow0 + x);
                *reinterpret_cast<float4*>(drow + x) = make_float4(fmaxf(e.x * c.x, 0.0f), fmaxf(e.y * c.y, 0.0f),
                                                                    fmaxf(e.z * c.z, 0.0f), fmaxf(e.w * c.w, 0.0f));
                continue;
            }
            float out4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int qx = q0 + k;
                float v = 0.0f;
                if (in_zy && qx >= -f.lo[2] && qx < f.N[2] + f.hi[2]) {
                    qx += qx < 0 ? f.N[2] : (qx >= f.N[2] ? -f.N[2] : 0);
                    int tx[2] = {qx + f.off[2], 0};
                    int nx = 1;
                    if (qx < f.hi[2]) tx[nx++] = qx + f.N[2] + f.off[2];
                    else if (qx >= f.N[2] - f.lo[2]) tx[nx++] = qx - f.N[2] + f.off[2];
                    float acc = 0.0f;
                    for (int a = 0; a < nz; ++a)
                        for (int b = 0; b < ny; ++b) {
                            const float* crow = corr + ((int64_t)tz[a] * f.P[1] + ty[b]) * f.P[2];
                            for (int c = 0; c < nx; ++c) acc += crow[tx[c]];
                        }
                    v = fmaxf(erow[qx] * acc, 0.0f);
                }
                out4[k] = v;
            }
            *reinterpret_cast<float4*>(drow + x) = make_float4(out4[0], out4[1], out4[2], out4[3]);
        }
    }
}

// out <- max(in, 0)
__global__ void clip_copy_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = fmaxf(in[i], 0.0f);
}

// ---- phase cross-correlation (estimate_stabilization.py:199-256) -------------------------------------------
// prod = F1 * conj(F2) / norm / V   (the 1/V makes the C2R a normalised irfftn)
// dst may be either factor's buffer (every thread reads its bin of both before it writes)
__global__ void pcc_product_kernel(const cf* f1, const cf* f2, cf* dst, int64_t n, int mode, float inv_v) {
    const float eps = 1.1920929e-07f;  // np.finfo(complex64).eps
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const cf a = f1[i], b = f2[i];
        cf p = make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
        float nrm = 1.0f;
        if (mode == BH_PCC_NORM_MAGNITUDE) nrm = fmaxf(hypotf(p.x, p.y), eps);
        if (mode == BH_PCC_NORM_CLASSIC) nrm = hypotf(a.x, a.y) * hypotf(b.x, b.y);
        p.x = (p.x / nrm) * inv_v;
        p.y = (p.y / nrm) * inv_v;
        dst[i] = p;
    }
}

__device__ __forceinline__ ArgMax better(ArgMax a, ArgMax b) {  // np.argmax: first occurrence of the maximum
    if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
    return a;
}
// |corr| argmax (first occurrence) + optional fftshift(|corr|) output, one pass
__global__ __launch_bounds__(256) void pcc_argmax_kernel(const float* __restrict__ corr, float* __restrict__ shifted,
                                                         int64_t Z, int64_t Y, int64_t X, ArgMax* partial) {
    const int64_t n = Z * Y * X;
    ArgMax best = {-1.0f, 0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float a = fabsf(corr[i]);
        if (a > best.v) best = ArgMax{a, i};  // i ascends within a thread: '>' keeps the first occurrence
        if (shifted) {
            const int64_t x = i % X, y = (i / X) % Y, z = i / (X * Y);
            const int64_t sx = (x + X / 2) % X, sy = (y + Y / 2) % Y, sz = (z + Z / 2) % Z;  // np.fft.fftshift
            shifted[(sz * Y + sy) * X + sx] = a;
        }
    }
    __shared__ ArgMax sh[256];
    sh[threadIdx.x] = best;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] = better(sh[threadIdx.x], sh[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(256) void pcc_argmax_final_kernel(const ArgMax* partial, int n, ArgMax* out) {
    __shared__ ArgMax sh[256];
    ArgMax best = {-1.0f, 0};
    for (int i = threadIdx.x; i < n; i += 256) best = better(best, partial[i]);
    sh[threadIdx.x] = best;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] = better(sh[threadIdx.x], sh[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = sh[0];
}

static inline dim3 grid_for(bh_ctx* ctx, int64_t n, int tb = 256) {
    int64_t g = ceil_div(n, tb);
    const int64_t cap = (int64_t)ctx->num_cus * 8;
    return dim3((unsigned)std::max<int64_t>(1, std::min(g, cap)));
}

static void pad_before(int64_t p, int64_t S, int* before) { *before = (int)((S - p) / 2); }

// fused FFT-convolution engine (fftconv.hip)
struct ConvPlan;
bool fftconv_supported(int64_t Z, int64_t Y, int64_t X);
bool fftconv_supported_ex(int64_t Z, int64_t Y, int64_t X, bool radix3);
bool fftconv_rows_wave_private(int64_t Y, int64_t X);
void fftconv_arm_rowsums(ConvPlan& pl, double* dst);
bool fftconv_rowsums_taken(ConvPlan& pl);
int fftconv_plan(bh_ctx* ctx, int64_t Z, int64_t Y, int64_t X, ConvPlan** out);
size_t fftconv_spectrum_elems(const ConvPlan& pl);
int fftconv_plan_tag(const ConvPlan& pl);
int fftconv_make_otf(bh_ctx* ctx, const ConvPlan& pl, const float* padded_psf, cf* otf);
int fftconv_apply(bh_ctx* ctx, const ConvPlan& pl, const float* in, const cf* otf, bool correlate, cf* spec,
                  int epilogue, const float* aux, float eps, float* out);
int fftconv_richardson_lucy(bh_ctx* ctx, const ConvPlan& pl, const float* d, const cf* otf, bool otf_real, cf* spec, int iterations,
                            float eps, float* est);
int fftconv_rl_iteration_padded(bh_ctx* ctx, const ConvPlan& pl, const float* est_p, const float* d_p, const cf* otf,
                                bool otf_real, cf* spec, float eps, float* corr_p);
bool fftconv_rl_wrap_supported(const ConvPlan& pl, const int64_t N[3], const int64_t K[3], const int64_t P[3]);
int fftconv_richardson_lucy_wrap(bh_ctx* ctx, const ConvPlan& pl, const float* d_p, const cf* otf, bool otf_real, cf* spec_a,
                                 cf* spec_b, float* est_a, float* est_b, const int64_t N[3], const int64_t K[3], int iterations,
                                 float eps, float* out);
int fftconv_forward(bh_ctx* ctx, const ConvPlan& pl, const float* in, cf* spec);
int fftconv_inverse(bh_ctx* ctx, const ConvPlan& pl, cf* spec, float* out);
int fftconv_tune_spectrum(bh_ctx* ctx, const ConvPlan& pl, float* est, size_t bytes, cf** spec);
bool fftconv_pcc_peak_only(const ConvPlan& pl);
int fftconv_pcc_apply(bh_ctx* ctx, const ConvPlan& pl, const float* img, cf* fixed, bool fixed_is_mov, bool roll, cf* s2, int norm,
                      float scale, float* corr, ArgMax* partial, int* npartial);
int fftconv_tikhonov(bh_ctx* ctx, const ConvPlan& pl, const float* in, const float* tf_full, float reg, cf* spec,
                     float* filt, float* out);

static bool use_fused_engine(int64_t Z, int64_t Y, int64_t X) {
    if (const char* e = getenv("BH_FFT_BACKEND"))
        if (strcmp(e, "hipfft") == 0) return false;
    return fftconv_supported(Z, Y, X);
}
// callers whose spectral arithmetic treats every coefficient alike may also take z / y of 3 * 2^k (radix-3 column passes)
static bool use_fused_engine_any_order(int64_t Z, int64_t Y, int64_t X) {
    if (const char* e = getenv("BH_FFT_BACKEND"))
        if (strcmp(e, "hipfft") == 0) return false;
    return fftconv_supported_ex(Z, Y, X, getenv("BH_FC_NORADIX3") == nullptr);
}

// order-sensitive 64-bit content hash of a small device array (one block; the PSF is a few thousand floats).  With `kept`
// (the device copy of the PSF whose OTF is cached) out[1] also says whether the two arrays are equal word for word: a hash
// match alone never validates the cache.  out[2]: the array equals its point mirror bit for bit (flat index i <-> n - 1 - i).
__global__ __launch_bounds__(256) void content_hash_kernel(const uint32_t* __restrict__ data, int64_t n,
                                                           const uint32_t* __restrict__ kept, unsigned long long* out) {
    __shared__ unsigned long long sh[256];
    __shared__ int differs, asym;
    if (threadIdx.x == 0) differs = asym = 0;
    __syncthreads();
    unsigned long long h = 0xcbf29ce484222325ull ^ (unsigned long long)threadIdx.x;
    int diff = 0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const uint32_t v = data[i];
        if (kept && kept[i] != v) diff = 1;
        if (data[n - 1 - i] != v) asym = 1;  // point symmetry h(-n) == h(n) of an array with odd extents: flat index i <-> n - 1 - i
        h ^= (unsigned long long)v + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);
        h *= 0x100000001b3ull;
        h ^= h >> 29;
    }
    sh[threadIdx.x] = h;
    if (diff) differs = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0x84222325cbf29ce4ull;
        for (int i = 0; i < 256; ++i) t = (t ^ sh[i]) * 0x100000001b3ull + (t >> 31);
        out[0] = t;
        out[1] = (kept && !differs) ? 1ull : 0ull;
        out[2] = asym ? 0ull : 1ull;
    }
}

// zero-padded, unit-sum, origin-centred PSF in `real` (V floats)
static int stage_rl_psf(bh_ctx* ctx, const float* psf, int64_t pz, int64_t py, int64_t px, int64_t Z, int64_t Y,
                        int64_t X, float* real, double* psum) {
    hipStream_t s = ctx->stream;
    BH_CHECK_HIP(hipMemsetAsync(real, 0, (size_t)Z * Y * X * sizeof(float), s));
    hipLaunchKernelGGL(psf_sum_kernel, dim3(1), dim3(256), 0, s, psf, pz * py * px, psum);
    int bz, by, bx;
    pad_before(pz, Z, &bz);
    pad_before(py, Y, &by);
    pad_before(px, X, &bx);
    hipLaunchKernelGGL(place_psf_kernel, grid_for(ctx, pz * py * px), dim3(256), 0, s, psf, real, (int)pz, (int)py,
                       (int)px, Z, Y, X, bz, by, bx, bz + (int)(pz / 2), by + (int)(py / 2), bx + (int)(px / 2),
                       (const double*)psum);
    BH_CHECK_HIP(hipGetLastError());
    return BH_OK;
}

// The iterations of Richardson-Lucy on the fused engine with the transfer function in hand (`otf`: NS complex, or NS floats
// when `otf_real`): spectrum scratch (auditioned once per allocation), optional event timing, no host synchronisation
// otherwise.  Shared by the one-shot entry and the prepared handle.
static int rl_engine_run(bh_ctx* ctx, ConvPlan* pl, const float* d, const void* otf, bool otf_real, int iterations, float eps,
                         float* out) {
    const size_t NS = fftconv_spectrum_elems(*pl);
    cf* spec;
    // BH_FC_SPEC_X2=1: the spectrum as the first half of an allocation of twice its size (experiment: sub-ranges of larger
    // allocations never showed the slow state of the fused update pass) — no audition then
    static const bool spec_x2 = getenv("BH_FC_SPEC_X2") && atoi(getenv("BH_FC_SPEC_X2")) != 0;
    BH_TRY(get_scratch(ctx, "fc_spec", (spec_x2 ? 2 : 1) * NS * sizeof(cf), (void**)&spec));
    if (spec != ctx->spec_tuned && !spec_x2) {  // a new allocation: audition it (fftconv_tune_spectrum)
        Scratch& sc = ctx->scratch["fc_spec"];
        // the audition may free the allocation it was handed and keep another one: the scratch table must follow it on the
        // error path too, or the next get_scratch("fc_spec") hands out a freed pointer
        const int tune_rc = fftconv_tune_spectrum(ctx, *pl, out, sc.bytes, &spec);
        sc.ptr = spec;
        ctx->spec_tuned = spec;
        BH_TRY(tune_rc);
    }
    hipStream_t s = ctx->stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ctx->timing) {
        BH_CHECK_HIP(hipEventCreate(&e0));
        BH_CHECK_HIP(hipEventCreate(&e1));
        BH_CHECK_HIP(hipEventRecord(e0, s));
    }
    BH_TRY(fftconv_richardson_lucy(ctx, *pl, d, reinterpret_cast<const cf*>(otf), otf_real, spec, iterations, eps, out));
    if (e0) {
        BH_CHECK_HIP(hipEventRecord(e1, s));
        BH_CHECK_HIP(hipEventSynchronize(e1));
        float ms = 0;
        BH_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
        ctx->ms_override[T_RL_ITER] = ms / iterations;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    return BH_OK;
}

// Richardson-Lucy on the fused engine: per iteration two 5-pass convolutions, the divide and the
// multiply/clip ride in the inverse X passes.  One-shot form: the transfer function is cached in the context and validated
// against the PSF's bytes on every call (one small kernel, a 24-byte read-back and a stream synchronisation);
// bh_richardson_lucy_create / _apply is the form without that per-call stall.
static int richardson_lucy_fused(bh_ctx* ctx, const float* d, const float* psf, int64_t pz, int64_t py, int64_t px,
                                 int64_t Z, int64_t Y, int64_t X, int iterations, float eps, float* out) {
    const int64_t V = Z * Y * X;
    if (iterations == 0) {  // e0 = max(d, 0): nothing to convolve (the fused passes write `out` only from iteration 1 on)
        hipLaunchKernelGGL(clip_copy_kernel, grid_for(ctx, V), dim3(256), 0, ctx->stream, d, out, V);
        BH_CHECK_HIP(hipGetLastError());
        return BH_OK;
    }
    ConvPlan* pl;
    BH_TRY(fftconv_plan(ctx, Z, Y, X, &pl));
    const size_t NS = fftconv_spectrum_elems(*pl);
    float* real = nullptr;
    cf* otf;
    double* psum;
    BH_TRY(get_scratch(ctx, "fc_otf", NS * sizeof(cf), (void**)&otf));
    BH_TRY(get_scratch(ctx, "rl_psum", 64, (void**)&psum));
    hipStream_t s = ctx->stream;
    ScopedTimer timer(ctx, T_RL_TOTAL);
    // The OTF only depends on the PSF and the shapes: a plate reuses one PSF for every position, so keep the OTF
    // across calls and rebuild it only when the PSF's content hash (or a shape) changes.
    unsigned long long* dhash = reinterpret_cast<unsigned long long*>(psum) + 1;
    unsigned long long hv[3] = {0, 0, 0};
    const int64_t dims[6] = {pz, py, px, Z, Y, X};
    // which spectrum layout the plan's kernels keep (an OTF only fits its own) and whether the real form is wanted too
    const int tag = fftconv_plan_tag(*pl) + (getenv("BH_RL_COMPLEX_OTF") ? 0 : 2);
    bool same_key = ctx->otf_valid && ctx->otf_tag == tag;
    for (int i = 0; i < 6; ++i) same_key = same_key && ctx->otf_dims[i] == dims[i];
    const size_t psf_bytes = (size_t)(pz * py * px) * sizeof(float);
    uint32_t* kept = nullptr;  // device copy of the PSF the cached OTF was built from
    BH_TRY(get_scratch(ctx, "rl_psf_kept", psf_bytes, (void**)&kept));
    // one small kernel + one 16-byte read-back decide the hit: the hash is only a log key, equality of the bytes decides.
    // (A pointer match would not do instead of the read-back: the adapters upload the PSF anew for every call, and a buffer
    // that kept its address may have changed its contents.)
    hipLaunchKernelGGL(content_hash_kernel, dim3(1), dim3(256), 0, s, reinterpret_cast<const uint32_t*>(psf),
                       pz * py * px, same_key ? kept : nullptr, dhash);
    BH_CHECK_HIP(hipMemcpyAsync(hv, dhash, sizeof(hv), hipMemcpyDeviceToHost, s));
    BH_CHECK_HIP(hipStreamSynchronize(s));
    const bool hit = same_key && hv[1] == 1ull;
    // A PSF with odd extents that equals its point mirror bit for bit (every theoretical PSF; the bench's Gaussian) sits
    // symmetric about the origin after staging, so its transfer function is REAL: the imaginary parts the transforms leave
    // are round-off.  The Z passes then read one float per bin instead of two (17.4 -> 8.7 GB per iteration at config 2) and
    // convolution and correlation are the same pass.  BH_RL_COMPLEX_OTF=1 keeps the general path (A/B switch).
    const bool real_otf = (pz & 1) && (py & 1) && (px & 1) && hv[2] == 1ull && getenv("BH_RL_COMPLEX_OTF") == nullptr;
    float* otf_real = nullptr;
    if (real_otf) BH_TRY(get_scratch(ctx, "fc_otf_real", NS * sizeof(float), (void**)&otf_real));
    if (!hit) {
        ctx->otf_valid = false;
        // the padded PSF is staged in the spectrum buffer's own memory? no: it must survive the forward X pass that
        // writes the OTF, so it gets a real-volume scratch that is only ever needed on a cache miss
        BH_TRY(get_scratch(ctx, "fft_real", V * sizeof(float), (void**)&real));
        BH_TRY(stage_rl_psf(ctx, psf, pz, py, px, Z, Y, X, real, psum));
        BH_TRY(fftconv_make_otf(ctx, *pl, real, otf));
        if (real_otf) {
            hipLaunchKernelGGL(real_part_kernel, grid_for(ctx, (int64_t)NS), dim3(256), 0, s, otf, otf_real, (int64_t)NS);
            BH_CHECK_HIP(hipGetLastError());
        }
        BH_CHECK_HIP(hipMemcpyAsync(kept, psf, psf_bytes, hipMemcpyDeviceToDevice, s));
        ctx->otf_hash = hv[0];
        ctx->otf_tag = tag;
        for (int i = 0; i < 6; ++i) ctx->otf_dims[i] = dims[i];
        ctx->otf_valid = true;
    }
    return rl_engine_run(ctx, pl, d, real_otf ? (const void*)otf_real : (const void*)otf, real_otf, iterations, eps, out);
}

static bool is_smooth(int64_t n) {  // only the radices hipFFT has native kernels for
    for (int p : {2, 3, 5, 7})
        while (n % p == 0) n /= p;
    return n == 1;
}
static int64_t next_smooth(int64_t n) {
    while (!is_smooth(n)) ++n;
    return n;
}
static bool is_pow2(int64_t n) { return n > 0 && (n & (n - 1)) == 0; }

// Box the fused engine could run an awkward volume at: axes it transforms as they are stay (they wrap by themselves), the
// others grow to the next 2^k (3 * 2^k for z, y) >= N + K - 1: room for the wrap-extended estimate going into the
// convolution, and for the tails of the linear correlation coming out.
static bool engine_pad_box(const int64_t N[3], const int64_t K[3], int64_t P[3]) {
    const bool radix3 = getenv("BH_FC_NORADIX3") == nullptr;
    for (int a = 0; a < 3; ++a) {  // (z, y before x: the row choice below looks at the box's y extent)
        // axes the engine transforms as they are wrap by themselves: powers of two, 3 * 2^k and 5 * 2^k
        auto alone = [&](int64_t n) { return fftconv_supported_ex(a == 0 ? n : 64, a == 1 ? n : 64, a == 2 ? n : 64, true); };
        const bool odd_native = (N[a] % 3 == 0 && is_pow2(N[a] / 3)) || (N[a] % 5 == 0 && is_pow2(N[a] / 5));
        if (is_pow2(N[a]) || (radix3 && odd_native && alone(N[a]))) {
            P[a] = N[a];
            continue;
        }
        const int64_t need = N[a] + K[a] - 1;
        P[a] = 1;
        while (P[a] < need) P[a] *= 2;
        // 5 * 2^k or 3 * 2^k (odd first step of that axis' transform) when that is enough
        if (radix3 && P[a] >= 16) {
            const int64_t p5 = 5 * (P[a] / 8), p3 = 3 * (P[a] / 4);
            // Rows of 1536 / 3072 voxels run the wave-private radix-3 X passes (fftconv_x3.inc) and, with them, the wrap-padded
            // iteration without a fold pass; rows of 5 * 2^k voxels still take the tile X passes, which cost ~1.6x as much per
            // voxel (158 against 98 ms on the config-4 box) — more than the 1.2x larger box.  BH_RL_X5=1: old choice.
            // ... provided those passes are what the plan will enable for this box (its y extent, the A/B switches)
            const bool x3_rows = a == 2 && (p3 == 1536 || p3 == 3072) && getenv("BH_RL_X5") == nullptr && fftconv_rows_wave_private(P[1], p3);
            if (p5 >= need && alone(p5) && !(x3_rows && alone(p3))) P[a] = p5;
            else if (p3 >= need && alone(p3)) P[a] = p3;
        }
        if (K[a] - 1 >= N[a]) return false;  // the wrap below assumes margins shorter than the axis
    }
    return fftconv_supported_ex(P[0], P[1], P[2], true);
}

// Richardson-Lucy for an awkward shape on the fused engine at a larger box P.  On a padded axis the volume voxel n sits at
// n + off, off = K - 1 - K/2, and the estimate is wrap-extended by off below and K/2 above it: the convolution (taps at
// -K/2 .. K - 1 - K/2) is then the circular one on the volume's own box.  The data term is zero outside that box, so the
// ratio is too, and the correlation that follows is the LINEAR correlation of the zero-padded ratio; its tails (off below,
// K/2 above, inside the box because P >= N + K - 1) are folded back — the circular correlation at size N the definition
// asks for — by the kernel that also multiplies, clips and rebuilds the wrap-extension for the next iteration.
// rl_padded_run: the iterations with the transfer function of the box in hand (NS complex, or NS floats when `otf_real`); no
// host synchronisation unless the context is timing.
static int rl_padded_run(bh_ctx* ctx, ConvPlan* pl, const float* d, const void* otf, bool otf_real, const int64_t N[3],
                         const int64_t K[3], const int64_t P[3], int iterations, float eps, float* out) {
    const int64_t VP = P[0] * P[1] * P[2];
    hipStream_t s = ctx->stream;
    const size_t NS = fftconv_spectrum_elems(*pl);
    const bool wrap = fftconv_rl_wrap_supported(*pl, N, K, P);  // no fold pass: fftconv_richardson_lucy_wrap
    float *a, *b, *c = nullptr, *dp;
    cf *spec, *spec_b = nullptr;
    BH_TRY(get_scratch(ctx, "fft_real", VP * sizeof(float), (void**)&a));
    BH_TRY(get_scratch(ctx, "rl_real2", VP * sizeof(float), (void**)&b));
    if (!wrap) BH_TRY(get_scratch(ctx, "rl_corr_p", VP * sizeof(float), (void**)&c));
    BH_TRY(get_scratch(ctx, "rl_data_p", VP * sizeof(float), (void**)&dp));
    BH_TRY(get_scratch(ctx, "fc_spec", NS * sizeof(cf), (void**)&spec));
    if (wrap) BH_TRY(get_scratch(ctx, "fc_spec_b", NS * sizeof(cf), (void**)&spec_b));
    RemapDims pad, crop;
    FoldBox fold;
    for (int i = 0; i < 3; ++i) {
        const bool padded = P[i] != N[i];
        const int lo = padded ? (int)(K[i] - 1 - K[i] / 2) : 0, hi = padded ? (int)(K[i] / 2) : 0;
        pad.D[i] = crop.S[i] = fold.P[i] = P[i];
        pad.S[i] = crop.D[i] = N[i];
        pad.N[i] = crop.N[i] = fold.N[i] = (int)N[i];
        pad.doff[i] = crop.soff[i] = fold.off[i] = lo;
        pad.soff[i] = crop.doff[i] = 0;
        pad.lo[i] = fold.lo[i] = lo;
        pad.hi[i] = fold.hi[i] = hi;
        crop.lo[i] = crop.hi[i] = 0;
    }
    RemapDims pad_d = pad;  // the data term: the volume's own box only
    for (int i = 0; i < 3; ++i) pad_d.lo[i] = pad_d.hi[i] = 0;
    auto grid2 = [&](const int64_t D[3]) {  // a block walks up to 2048 voxels of one row: the per-row index work is paid once
        return dim3((unsigned)std::min<int64_t>(ceil_div(D[2], 2048), 16), (unsigned)std::min<int64_t>(D[0] * D[1], 65535));
    };
    if (wrap) {  // the data wrap-extended like the estimate: the first X pass clips it into the estimate and transforms it
        hipLaunchKernelGGL(remap_kernel<false>, grid2(pad.D), dim3(256), 0, s, d, dp, pad);
    } else {
        hipLaunchKernelGGL(remap_kernel<false>, grid2(pad_d.D), dim3(256), 0, s, d, dp, pad_d);
        hipLaunchKernelGGL(remap_kernel<true>, grid2(pad.D), dim3(256), 0, s, d, a, pad);  // e0 = max(d, 0), wrap-extended
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ctx->timing) {
        BH_CHECK_HIP(hipEventCreate(&e0));
        BH_CHECK_HIP(hipEventCreate(&e1));
        BH_CHECK_HIP(hipEventRecord(e0, s));
    }
    float *cur = a, *nxt = b;
    if (wrap) {  // the last pass stores the result cropped: no crop kernel
        BH_TRY(fftconv_richardson_lucy_wrap(ctx, *pl, dp, reinterpret_cast<const cf*>(otf), otf_real, spec, spec_b, a, b, N, K,
                                            iterations, eps, out));
    } else {
        for (int it = 0; it < iterations; ++it) {
            BH_TRY(fftconv_rl_iteration_padded(ctx, *pl, cur, dp, reinterpret_cast<const cf*>(otf), otf_real, spec, eps, c));
            hipLaunchKernelGGL(fold_update_rewrap_kernel, grid2(fold.P), dim3(256), 0, s, (const float*)c, (const float*)cur, nxt, fold);
            std::swap(cur, nxt);
        }
        hipLaunchKernelGGL(remap_kernel<false>, grid2(crop.D), dim3(256), 0, s, (const float*)cur, out, crop);
    }
    BH_CHECK_HIP(hipGetLastError());
    if (e0) {
        BH_CHECK_HIP(hipEventRecord(e1, s));
        BH_CHECK_HIP(hipEventSynchronize(e1));
        float ms = 0;
        BH_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
        ctx->ms_override[T_RL_ITER] = ms / iterations;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    return BH_OK;
}

static int richardson_lucy_engine_padded(bh_ctx* ctx, const float* d, const float* psf, int64_t pz, int64_t py, int64_t px,
                                         int64_t Z, int64_t Y, int64_t X, const int64_t P[3], int iterations, float eps,
                                         float* out) {
    const int64_t N[3] = {Z, Y, X}, K[3] = {pz, py, px};
    const int64_t V = Z * Y * X, VP = P[0] * P[1] * P[2];
    hipStream_t s = ctx->stream;
    if (iterations == 0) {
        hipLaunchKernelGGL(clip_copy_kernel, grid_for(ctx, V), dim3(256), 0, s, d, out, V);
        BH_CHECK_HIP(hipGetLastError());
        return BH_OK;
    }
    ConvPlan* pl;
    BH_TRY(fftconv_plan(ctx, P[0], P[1], P[2], &pl));
    const size_t NS = fftconv_spectrum_elems(*pl);
    float* a;
    cf* otf;
    double* psum;
    BH_TRY(get_scratch(ctx, "fft_real", VP * sizeof(float), (void**)&a));
    BH_TRY(get_scratch(ctx, "fc_otf", NS * sizeof(cf), (void**)&otf));
    BH_TRY(get_scratch(ctx, "rl_psum", 64, (void**)&psum));
    ctx->otf_valid = false;  // fc_otf is overwritten: the cache of richardson_lucy_fused no longer holds
    ScopedTimer timer(ctx, T_RL_TOTAL);
    BH_TRY(stage_rl_psf(ctx, psf, pz, py, px, P[0], P[1], P[2], a, psum));
    BH_TRY(fftconv_make_otf(ctx, *pl, a, otf));
    return rl_padded_run(ctx, pl, d, otf, false, N, K, P, iterations, eps, out);
}

// Which transform box and back-end Richardson-Lucy uses for a shape (host logic only; bh_richardson_lucy_plan exports it).
static int rl_plan(int64_t pz, int64_t py, int64_t px, int64_t Z, int64_t Y, int64_t X, int64_t box[3]) {
    const int64_t N[3] = {Z, Y, X}, K[3] = {pz, py, px};
    box[0] = Z, box[1] = Y, box[2] = X;
    if (use_fused_engine_any_order(Z, Y, X)) return BH_RL_ENGINE;  // z / y of 3 * 2^k: radix-3 first step in the column passes
    int64_t P[3], PE[3];
    const bool nopad = getenv("BH_RL_NOPAD") != nullptr;
    for (int a = 0; a < 3; ++a) P[a] = (nopad || is_smooth(N[a])) ? N[a] : next_smooth(N[a] + K[a] - 1);
    // The fused engine at a power-of-two (or 3 * 2^k) box against hipFFT at the 7-smooth one: the engine moves a voxel of its
    // box about 2x faster (9 passes + fold at ~3.9 Gvox/s against the library path's ~1.9 Gvox/s), so it wins
    // unless its box is more than twice as large.  BH_RL_ENGINE_PAD=0 / 1 forces the choice.
    const char* force = getenv("BH_RL_ENGINE_PAD");
    const char* be = getenv("BH_FFT_BACKEND");
    const bool hipfft_forced = be != nullptr && strcmp(be, "hipfft") == 0;
    if (!hipfft_forced && !nopad && !(force && force[0] == '0') && engine_pad_box(N, K, PE)) {
        // (round 3: rows the wave-private X passes take, Y unpadded, run the 8-pass wrap-padded iteration at ~6 Gvox/s)
        const bool wrap_rows = fftconv_rows_wave_private(PE[1], PE[2]) && PE[1] == N[1] && K[2] <= 256 && getenv("BH_RL_NOWRAP") == nullptr;
        const double cost_engine = (double)PE[0] * PE[1] * PE[2] / (wrap_rows ? 6.0 : 3.9), cost_lib = (double)P[0] * P[1] * P[2] / 1.9;
        if ((force && force[0] == '1') || cost_engine < cost_lib) {
            for (int a = 0; a < 3; ++a) box[a] = PE[a];
            return BH_RL_ENGINE_PADDED;
        }
    }
    for (int a = 0; a < 3; ++a) box[a] = P[a];
    return BH_RL_LIBRARY;
}

}  // namespace bh

using namespace bh;

extern "C" {

int bh_transfer_function(bh_ctx* ctx, const float* psf, int64_t pz, int64_t py, int64_t px, int64_t Z, int64_t Y,
                         int64_t X, float* tf_full) {
    BH_REQUIRE(ctx && psf && tf_full, "NULL argument");
    BH_REQUIRE(pz > 0 && py > 0 && px > 0, "invalid PSF shape");
    BH_REQUIRE(pz <= Z && py <= Y && px <= X, "PSF (%lld,%lld,%lld) larger than volume (%lld,%lld,%lld)",
               (long long)pz, (long long)py, (long long)px, (long long)Z, (long long)Y, (long long)X);
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    ScopedTimer timer(ctx, T_TF);
    FftPlans* pl;
    BH_TRY(get_plans(ctx, Z, Y, X, &pl));
    const int64_t V = Z * Y * X, Xh = X / 2 + 1, NS = Z * Y * Xh;
    float* real;
    cf* spec;
    unsigned int* gmax;
    BH_TRY(get_scratch(ctx, "fft_real", V * sizeof(float), (void**)&real));
    BH_TRY(get_scratch(ctx, "fft_spec", NS * sizeof(cf), (void**)&spec));
    BH_TRY(get_scratch(ctx, "tf_max", 64, (void**)&gmax));
    hipStream_t s = ctx->stream;
    BH_CHECK_HIP(hipMemsetAsync(real, 0, V * sizeof(float), s));
    BH_CHECK_HIP(hipMemsetAsync(gmax, 0, 4, s));
    int bz, by, bx;
    pad_before(pz, Z, &bz);
    pad_before(py, Y, &by);
    pad_before(px, X, &bx);
    hipLaunchKernelGGL(place_psf_kernel, grid_for(ctx, pz * py * px), dim3(256), 0, s, psf, real, (int)pz, (int)py,
                       (int)px, Z, Y, X, bz, by, bx, 0, 0, 0, (const double*)nullptr);
    BH_TRY(fft_forward(pl, real, spec));
    // magnitude goes into the (now free) real buffer: NS floats <= V + slack? NS*4 <= V*4 only when Xh <= X
    float* mag;
    BH_TRY(get_scratch(ctx, "tf_mag", NS * sizeof(float), (void**)&mag));
    hipLaunchKernelGGL(abs_max_kernel, grid_for(ctx, NS), dim3(256), 0, s, spec, mag, NS, gmax);
    hipLaunchKernelGGL(tf_expand_kernel, grid_for(ctx, V), dim3(256), 0, s, mag, tf_full, Z, Y, X, gmax);
    BH_CHECK_HIP(hipGetLastError());
    return BH_OK;
}

// ---- phase cross-correlation: one image's finished spectrum is kept (scratch for the one-shot call, a pooled block for a
// prepared handle), the other image meets it inside its own transform
struct PccShape {
    int64_t Z, Y, X, Xc;  // Xc: the reference calls irfftn without a shape (estimate_stabilization.py:240), so for an odd X the
                          // correlation volume comes back with X - 1 columns; the same half spectrum is inverted by a (Z, Y, Xc) plan
    bool engine;
    ConvPlan* cp;
    size_t spec_elems;    // complex elements of one spectrum buffer
};
static int pcc_shape(bh_ctx* ctx, int64_t Z, int64_t Y, int64_t X, PccShape* sh) {
    BH_REQUIRE(Z > 0 && Y > 0 && X > 0, "invalid shape");
    sh->Z = Z, sh->Y = Y, sh->X = X, sh->Xc = (X & 1) ? X - 1 : X;
    BH_REQUIRE(sh->Xc >= 2, "X must be at least 2");
    sh->cp = nullptr;
    // power-of-two (z, y also 3 * 2^k) volume: in-place passes on the fused engine instead of hipFFT's transposing pipeline;
    // the normalised product treats every coefficient alike, so their order does not matter
    sh->engine = use_fused_engine_any_order(Z, Y, X);
    if (sh->engine) {
        BH_TRY(fftconv_plan(ctx, Z, Y, X, &sh->cp));
        sh->spec_elems = fftconv_spectrum_elems(*sh->cp);
    } else {
        sh->spec_elems = (size_t)(Z * Y * (X / 2 + 1));
        // both library plans now: a new plan's round-trip self-check runs through the shared FFT scratch, which must not
        // happen between the stored spectrum's transform and its use
        FftPlans* pl;
        BH_TRY(get_plans(ctx, Z, Y, X, &pl));
        BH_TRY(get_plans(ctx, Z, Y, sh->Xc, &pl));
    }
    return BH_OK;
}
// the stored image's spectrum
static int pcc_prepare(bh_ctx* ctx, const PccShape& sh, const float* fixed_img, cf* fixed) {
    if (sh.engine) return fftconv_forward(ctx, *sh.cp, fixed_img, fixed);
    FftPlans* pl;
    BH_TRY(get_plans(ctx, sh.Z, sh.Y, sh.X, &pl));
    return fft_forward(pl, fixed_img, fixed);
}
static int pcc_run(bh_ctx* ctx, const PccShape& sh, cf* fixed, bool fixed_is_second, bool roll, const float* img, int normalization,
                   float shift[3], float* corr_shifted) {
    const int64_t Z = sh.Z, Y = sh.Y, X = sh.X, Xc = sh.Xc, V = Z * Y * Xc;
    cf* s2;
    float* corr = nullptr;
    ArgMax *partial, *result;
    const int nblk = ctx->num_cus * 8;
    int npartial = nblk;
    BH_TRY(get_scratch(ctx, "pcc_partial", (nblk + 1) * sizeof(ArgMax), (void**)&partial));
    result = partial + nblk;
    hipStream_t s = ctx->stream;
    // Only the peak is wanted and the rows are ones the wave-private X kernels take: the last inverse pass keeps the argmax
    // candidates itself and the correlation volume never exists (no store, no search pass, no scratch volume).
    const bool unfused = getenv("BH_PCC_UNFUSED") != nullptr;
    const bool peak_only = sh.engine && !corr_shifted && !unfused && fftconv_pcc_peak_only(*sh.cp);
    if (!peak_only) BH_TRY(get_scratch(ctx, "fft_real", Z * Y * X * sizeof(float), (void**)&corr));
    BH_TRY(get_scratch(ctx, "pcc_spec2", sh.spec_elems * sizeof(cf), (void**)&s2));
    const cf* first = fixed_is_second ? s2 : fixed;
    const cf* second = fixed_is_second ? fixed : s2;
    if (sh.engine && !unfused) {
        BH_TRY(fftconv_pcc_apply(ctx, *sh.cp, img, fixed, fixed_is_second, roll, s2, normalization, (float)(2.0 / (double)V), corr,
                                 partial, &npartial));
        BH_REQUIRE(npartial <= nblk, "internal: %d argmax candidates for %d slots", npartial, nblk);
    } else {
        // library path (and the engine's A/B switch BH_PCC_UNFUSED): whole transforms, the product as a pass of its own.  It
        // is written over the stored spectrum when that one is being replaced anyway (roll), over the new one otherwise.
        FftPlans *pl = nullptr, *plc = nullptr;
        if (sh.engine) {
            BH_TRY(fftconv_forward(ctx, *sh.cp, img, s2));
        } else {
            BH_TRY(get_plans(ctx, Z, Y, X, &pl));
            BH_TRY(get_plans(ctx, Z, Y, Xc, &plc));
            BH_TRY(fft_forward(pl, img, s2));
        }
        // engine: Z * Y * XP stored coefficients (pad columns are zero), not the tail slack
        const int64_t nprod = sh.engine ? (int64_t)sh.spec_elems - 64 : (int64_t)sh.spec_elems;
        cf* prod = roll ? fixed : s2;
        cf* keep = nullptr;
        if (roll) {  // the new image's spectrum survives the inverse transform in a buffer of its own
            BH_TRY(get_scratch(ctx, "pcc_spec3", sh.spec_elems * sizeof(cf), (void**)&keep));
            BH_CHECK_HIP(hipMemcpyAsync(keep, s2, sh.spec_elems * sizeof(cf), hipMemcpyDeviceToDevice, s));
        }
        hipLaunchKernelGGL(pcc_product_kernel, grid_for(ctx, nprod), dim3(256), 0, s, first, second, prod, nprod, normalization,
                           (float)((sh.engine ? 2.0 : 1.0) / (double)V));
        if (sh.engine) BH_TRY(fftconv_inverse(ctx, *sh.cp, prod, corr));
        else BH_TRY(fft_inverse(plc, prod, corr));
        if (roll) BH_CHECK_HIP(hipMemcpyAsync(fixed, keep, sh.spec_elems * sizeof(cf), hipMemcpyDeviceToDevice, s));
    }
    if (!peak_only) hipLaunchKernelGGL(pcc_argmax_kernel, dim3(nblk), dim3(256), 0, s, corr, corr_shifted, Z, Y, Xc, partial);
    hipLaunchKernelGGL(pcc_argmax_final_kernel, dim3(1), dim3(256), 0, s, partial, peak_only ? npartial : nblk, result);
    BH_CHECK_HIP(hipGetLastError());
    ArgMax h;
    BH_CHECK_HIP(hipMemcpyAsync(&h, result, sizeof(h), hipMemcpyDeviceToHost, s));
    BH_CHECK_HIP(hipStreamSynchronize(s));
    const int64_t dims[3] = {Z, Y, Xc};
    int64_t idx[3] = {h.i / (Y * Xc), (h.i / Xc) % Y, h.i % Xc};
    for (int a = 0; a < 3; ++a) {
        // shift[shift > fix(n/2)] -= n   (estimate_stabilization.py:246-252)
        float sft = (float)idx[a];
        if (sft > (float)(dims[a] / 2)) sft -= (float)dims[a];
        shift[a] = sft;
    }
    return BH_OK;
}

int bh_phase_cross_corr(bh_ctx* ctx, const float* ref, const float* mov, int64_t Z, int64_t Y, int64_t X,
                        int normalization, float shift[3], float* corr_shifted) {
    BH_REQUIRE(ctx && ref && mov && shift, "NULL argument");
    BH_REQUIRE(normalization >= BH_PCC_NORM_NONE && normalization <= BH_PCC_NORM_CLASSIC, "unknown normalization %d",
               normalization);
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    PccShape sh;
    BH_TRY(pcc_shape(ctx, Z, Y, X, &sh));
    cf* s1;
    BH_TRY(get_scratch(ctx, sh.engine ? "fc_spec" : "fft_spec", sh.spec_elems * sizeof(cf), (void**)&s1));
    BH_TRY(pcc_prepare(ctx, sh, ref, s1));
    return pcc_run(ctx, sh, s1, false, false, mov, normalization, shift, corr_shifted);
}

// Prepared form: the stabilisation estimate correlates every timepoint with ONE image (t_reference "first") or with its
// predecessor ("previous") — estimate_stabilization.py:505-520 — so that image's three forward passes are done once (or, with
// `roll`, fall out of the previous call's Z pass) and a call costs the other image's transform only.
struct bh_pcc {
    int device = 0;
    PccShape sh;
    bool fixed_is_second = false;
    cf* spec = nullptr;
    size_t bytes = 0;
};

int bh_phase_cross_corr_create(bh_ctx* ctx, const float* fixed, int64_t Z, int64_t Y, int64_t X, int fixed_is_second,
                               bh_pcc** out) {
    BH_REQUIRE(ctx && fixed && out, "NULL argument");
    *out = nullptr;
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    PccShape sh;
    BH_TRY(pcc_shape(ctx, Z, Y, X, &sh));
    bh_pcc* h = new bh_pcc();
    h->device = ctx->device;
    h->sh = sh;
    h->fixed_is_second = fixed_is_second != 0;
    h->bytes = sh.spec_elems * sizeof(cf);
    h->spec = static_cast<cf*>(filter_pool_take(ctx->device, h->bytes));
    if (!h->spec && dev_alloc(ctx->device, h->bytes, (void**)&h->spec) != hipSuccess) {
        (void)hipGetLastError();
        set_error("out of device memory (%zu bytes for the stored spectrum)", h->bytes);
        delete h;
        return BH_ERR_HIP;
    }
    const int rc = pcc_prepare(ctx, sh, fixed, h->spec);
    if (rc != BH_OK) {
        (void)bh_phase_cross_corr_destroy(h);
        return rc;
    }
    *out = h;
    return BH_OK;
}

int bh_phase_cross_corr_apply(bh_ctx* ctx, bh_pcc* h, const float* img, int normalization, int roll, float shift[3],
                              float* corr_shifted) {
    BH_REQUIRE(ctx && h && img && shift, "NULL argument");
    BH_REQUIRE(ctx->device == h->device, "handle was created on device %d, context is on device %d", h->device, ctx->device);
    BH_REQUIRE(normalization >= BH_PCC_NORM_NONE && normalization <= BH_PCC_NORM_CLASSIC, "unknown normalization %d",
               normalization);
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    return pcc_run(ctx, h->sh, h->spec, h->fixed_is_second, roll != 0, img, normalization, shift, corr_shifted);
}

int bh_phase_cross_corr_destroy(bh_pcc* h) {
    if (!h) return BH_OK;
    if (h->spec) filter_pool_give(h->device, h->bytes, h->spec);
    delete h;
    return BH_OK;
}

int bh_tikhonov(bh_ctx* ctx, const float* in, const float* tf_full, int64_t Z, int64_t Y, int64_t X,
                double regularization_strength, float* out) {
    BH_REQUIRE(ctx && in && tf_full && out, "NULL argument");
    BH_REQUIRE(Z > 0 && Y > 0 && X > 0, "invalid shape");
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    ScopedTimer timer(ctx, T_TIKHONOV);
    if (use_fused_engine_any_order(Z, Y, X)) {
        ConvPlan* cp;
        BH_TRY(fftconv_plan(ctx, Z, Y, X, &cp));
        const size_t NSf = fftconv_spectrum_elems(*cp);
        cf* fspec;
        float* filt;
        BH_TRY(get_scratch(ctx, "fc_spec", NSf * sizeof(cf), (void**)&fspec));
        BH_TRY(get_scratch(ctx, "fc_filter", NSf * sizeof(float), (void**)&filt));
        return fftconv_tikhonov(ctx, *cp, in, tf_full, (float)regularization_strength, fspec, filt, out);
    }
    FftPlans* pl;
    BH_TRY(get_plans(ctx, Z, Y, X, &pl));
    const int64_t V = Z * Y * X, Xh = X / 2 + 1, NS = Z * Y * Xh;
    cf* spec;
    BH_TRY(get_scratch(ctx, "fft_spec", NS * sizeof(cf), (void**)&spec));
    hipStream_t s = ctx->stream;
    BH_TRY(fft_forward(pl, in, spec));
    hipLaunchKernelGGL(tikhonov_filter_kernel, grid_for(ctx, NS), dim3(256), 0, s, spec, tf_full, Z, Y, X,
                       (float)regularization_strength, (float)(1.0 / (double)V));
    BH_TRY(fft_inverse(pl, spec, out));
    BH_CHECK_HIP(hipGetLastError());
    return BH_OK;
}


// Richardson-Lucy for a volume with an awkward axis (a large prime factor makes hipFFT fall back to Bluestein: the
// deskewed (342, 1024, 1517) runs 10x slower per voxel than a power of two).  Awkward axes are zero-padded to the next
// 7-smooth P >= N + K - 1 and the wrapped-around part of the linear convolution is folded back, which is exactly the
// circular convolution at size N the definition asks for; smooth axes keep P = N and wrap by themselves.
static int richardson_lucy_padfold(bh_ctx* ctx, const float* d, const float* psf, int64_t pz, int64_t py, int64_t px,
                                   int64_t Z, int64_t Y, int64_t X, const int64_t P[3], int iterations, float eps,
                                   float* out) {
    const int64_t N[3] = {Z, Y, X}, K[3] = {pz, py, px};
    FoldDims conv, corr;
    for (int a = 0; a < 3; ++a) {
        conv.N[a] = corr.N[a] = N[a];
        conv.P[a] = corr.P[a] = P[a];
        const bool padded = P[a] != N[a];
        conv.elo[a] = padded ? (int)(K[a] / 2) : 0;             // kernel taps sit at -K/2 .. K-1-K/2
        conv.ehi[a] = padded ? (int)(K[a] - 1 - K[a] / 2) : 0;
        corr.elo[a] = conv.ehi[a];                               // the correlation kernel is the mirror image
        corr.ehi[a] = conv.elo[a];
    }
    const int64_t V = Z * Y * X, VP = P[0] * P[1] * P[2], NS = P[0] * P[1] * (P[2] / 2 + 1);
    FftPlans* pl;
    BH_TRY(get_plans(ctx, P[0], P[1], P[2], &pl));
    float *ra, *rb;
    cf *spec, *otf;
    double* psum;
    BH_TRY(get_scratch(ctx, "fft_real", VP * sizeof(float), (void**)&ra));
    BH_TRY(get_scratch(ctx, "rl_real2", VP * sizeof(float), (void**)&rb));
    BH_TRY(get_scratch(ctx, "fft_spec", NS * sizeof(cf), (void**)&spec));
    BH_TRY(get_scratch(ctx, "rl_otf", NS * sizeof(cf), (void**)&otf));
    BH_TRY(get_scratch(ctx, "rl_psum", 64, (void**)&psum));
    hipStream_t s = ctx->stream;
    ScopedTimer timer(ctx, T_RL_TOTAL);
    BH_CHECK_HIP(hipMemsetAsync(ra, 0, VP * sizeof(float), s));
    hipLaunchKernelGGL(psf_sum_kernel, dim3(1), dim3(256), 0, s, psf, pz * py * px, psum);
    hipLaunchKernelGGL(place_psf_kernel, grid_for(ctx, pz * py * px), dim3(256), 0, s, psf, ra, (int)pz, (int)py, (int)px,
                       P[0], P[1], P[2], 0, 0, 0, (int)(pz / 2), (int)(py / 2), (int)(px / 2), (const double*)psum);
    BH_TRY(fft_forward(pl, ra, otf));
    hipLaunchKernelGGL(scale_spectrum_kernel, grid_for(ctx, NS), dim3(256), 0, s, otf, NS, (float)(1.0 / (double)VP));
    hipLaunchKernelGGL(clip_copy_kernel, grid_for(ctx, V), dim3(256), 0, s, d, out, V);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ctx->timing && iterations > 0) {
        BH_CHECK_HIP(hipEventCreate(&e0));
        BH_CHECK_HIP(hipEventCreate(&e1));
        BH_CHECK_HIP(hipEventRecord(e0, s));
    }
    const int64_t n2 = NS / 2;
    for (int it = 0; it < iterations; ++it) {
        hipLaunchKernelGGL(pad_volume_kernel, grid_for(ctx, VP), dim3(256), 0, s, (const float*)out, ra, conv);
        BH_TRY(fft_forward(pl, ra, spec));
        hipLaunchKernelGGL(cmul_kernel<false>, grid_for(ctx, n2), dim3(256), 0, s, spec, otf, n2);
        if (NS & 1) hipLaunchKernelGGL(cmul_tail_kernel<false>, dim3(1), dim3(1), 0, s, spec, otf, NS - 1);
        BH_TRY(fft_inverse(pl, spec, ra));
        hipLaunchKernelGGL(fold_ratio_kernel, grid_for(ctx, VP), dim3(256), 0, s, (const float*)ra, d, rb, conv, eps);
        BH_TRY(fft_forward(pl, rb, spec));
        hipLaunchKernelGGL(cmul_kernel<true>, grid_for(ctx, n2), dim3(256), 0, s, spec, otf, n2);
        if (NS & 1) hipLaunchKernelGGL(cmul_tail_kernel<true>, dim3(1), dim3(1), 0, s, spec, otf, NS - 1);
        BH_TRY(fft_inverse(pl, spec, ra));
        hipLaunchKernelGGL(fold_update_kernel, grid_for(ctx, V), dim3(256), 0, s, (const float*)ra, out, corr);
    }
    BH_CHECK_HIP(hipGetLastError());
    if (e0) {
        BH_CHECK_HIP(hipEventRecord(e1, s));
        BH_CHECK_HIP(hipEventSynchronize(e1));
        float ms = 0;
        BH_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
        ctx->ms_override[T_RL_ITER] = ms / iterations;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    return BH_OK;
}

int bh_richardson_lucy_plan(int64_t pz, int64_t py, int64_t px, int64_t Z, int64_t Y, int64_t X, int64_t box[3],
                            int* backend) {
    BH_REQUIRE(box != nullptr && backend != nullptr, "NULL argument");
    BH_REQUIRE(pz > 0 && py > 0 && px > 0 && pz <= Z && py <= Y && px <= X, "PSF must fit inside the volume");
    *backend = rl_plan(pz, py, px, Z, Y, X, box);
    return BH_OK;
}

// ---- prepared Richardson-Lucy: the transfer function built once, applied to any number of volumes ----
// The reference computes the transfer function once per plate and hands it to every (position, t, c) unit
// (biahub/deconvolve.py:140-149, 183-191).  `create` does that part — normalise, pad, centre and transform the PSF at the box
// bh_richardson_lucy_plan picks, decide whether the transfer function is real — and may synchronise; `apply` only enqueues
// kernels on the context's stream (no read-back, no host synchronisation unless the context is timing), so the legs of an
// upload / compute / download pipeline overlap.
struct bh_rl {
    int device = 0;
    int backend = 0;
    int64_t K[3] = {0, 0, 0}, N[3] = {0, 0, 0}, box[3] = {0, 0, 0};
    bh::ConvPlan* plan = nullptr;
    void* otf = nullptr;       // engine back-ends: NS complex, or NS floats when otf_real (owned, pooled on destroy)
    size_t otf_bytes = 0;
    bool otf_real = false;
    float* psf = nullptr;      // library back-end: the PSF itself (the one-shot path rebuilds its library transfer function)
};

int bh_richardson_lucy_create(bh_ctx* ctx, const float* psf, int64_t pz, int64_t py, int64_t px, int64_t Z, int64_t Y,
                              int64_t X, bh_rl** out) {
    BH_REQUIRE(ctx && psf && out, "NULL argument");
    BH_REQUIRE(pz > 0 && py > 0 && px > 0 && pz <= Z && py <= Y && px <= X, "PSF must fit inside the volume");
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    *out = nullptr;
    bh_rl* h = new bh_rl();
    h->device = ctx->device;
    h->K[0] = pz, h->K[1] = py, h->K[2] = px;
    h->N[0] = Z, h->N[1] = Y, h->N[2] = X;
    h->backend = rl_plan(pz, py, px, Z, Y, X, h->box);
    hipStream_t s = ctx->stream;
    auto fail = [&](int rc) {
        (void)bh_richardson_lucy_destroy(h);
        return rc;
    };
    const size_t psf_bytes = (size_t)(pz * py * px) * sizeof(float);
    if (h->backend == BH_RL_LIBRARY) {
        if (hipMalloc((void**)&h->psf, psf_bytes) != hipSuccess) return (set_error("out of device memory (PSF copy)"), fail(BH_ERR_HIP));
        if (hipMemcpyAsync(h->psf, psf, psf_bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return (set_error("PSF copy failed"), fail(BH_ERR_HIP));
        *out = h;
        return BH_OK;
    }
    int rc = fftconv_plan(ctx, h->box[0], h->box[1], h->box[2], &h->plan);
    if (rc != BH_OK) return fail(rc);
    const size_t NS = fftconv_spectrum_elems(*h->plan);
    const int64_t VP = h->box[0] * h->box[1] * h->box[2];
    // real transfer function?  (a PSF with odd extents that equals its point mirror bit for bit: richardson_lucy_fused)
    double* psum;
    if ((rc = get_scratch(ctx, "rl_psum", 64, (void**)&psum)) != BH_OK) return fail(rc);
    unsigned long long* dhash = reinterpret_cast<unsigned long long*>(psum) + 1;
    unsigned long long hv[3] = {0, 0, 0};
    hipLaunchKernelGGL(content_hash_kernel, dim3(1), dim3(256), 0, s, reinterpret_cast<const uint32_t*>(psf), pz * py * px,
                       (const uint32_t*)nullptr, dhash);
    if (hipMemcpyAsync(hv, dhash, sizeof(hv), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return (set_error("PSF symmetry check failed"), fail(BH_ERR_HIP));
    h->otf_real = (pz & 1) && (py & 1) && (px & 1) && hv[2] == 1ull && getenv("BH_RL_COMPLEX_OTF") == nullptr;
    h->otf_bytes = NS * (h->otf_real ? sizeof(float) : sizeof(cf));
    if ((h->otf = filter_pool_take(ctx->device, h->otf_bytes)) == nullptr &&
        dev_alloc(ctx->device, h->otf_bytes, &h->otf) != hipSuccess) {
        h->otf = nullptr;
        (void)hipGetLastError();
        return (set_error("out of device memory (%zu bytes for the transfer function)", h->otf_bytes), fail(BH_ERR_HIP));
    }
    // the padded PSF and (for the real form) the complex transfer function are transients in the context's scratch
    float* real;
    cf* otf_c = reinterpret_cast<cf*>(h->otf);
    if ((rc = get_scratch(ctx, "fft_real", VP * sizeof(float), (void**)&real)) != BH_OK) return fail(rc);
    if (h->otf_real) {
        if ((rc = get_scratch(ctx, "fc_otf", NS * sizeof(cf), (void**)&otf_c)) != BH_OK) return fail(rc);
        ctx->otf_valid = false;  // fc_otf is overwritten: the one-shot path's cache no longer holds
    }
    if ((rc = stage_rl_psf(ctx, psf, pz, py, px, h->box[0], h->box[1], h->box[2], real, psum)) != BH_OK) return fail(rc);
    if ((rc = fftconv_make_otf(ctx, *h->plan, real, otf_c)) != BH_OK) return fail(rc);
    if (h->otf_real) {
        hipLaunchKernelGGL(real_part_kernel, grid_for(ctx, (int64_t)NS), dim3(256), 0, s, otf_c, reinterpret_cast<float*>(h->otf),
                           (int64_t)NS);
        if (hipGetLastError() != hipSuccess) return (set_error("real_part_kernel launch failed"), fail(BH_ERR_HIP));
    }
    // the transients of the set-up go back to the driver: the complex transfer function when only its real part is kept (and
    // with it the one-shot path's cache), the padded PSF when the apply path has no use for a real-volume scratch of its own
    if (h->otf_real && (rc = free_scratch(ctx, "fc_otf")) != BH_OK) return fail(rc);
    if (h->backend == BH_RL_ENGINE && (rc = free_scratch(ctx, "fft_real")) != BH_OK) return fail(rc);
    *out = h;
    return BH_OK;
}

int bh_richardson_lucy_apply(bh_ctx* ctx, const bh_rl* h, const float* in, int iterations, float eps, float* out) {
    return bh_richardson_lucy_apply_rows(ctx, h, in, iterations, eps, out, nullptr, nullptr);
}

int bh_richardson_lucy_apply_rows(bh_ctx* ctx, const bh_rl* h, const float* in, int iterations, float eps, float* out,
                                  double* row_sums, int* produced) {
    if (produced) *produced = 0;
    BH_REQUIRE(ctx && h && in && out, "NULL argument");
    BH_REQUIRE(row_sums == nullptr || produced != nullptr, "row_sums needs `produced`");
    BH_REQUIRE(iterations >= 0, "iterations must be >= 0");
    BH_REQUIRE(ctx->device == h->device, "the handle belongs to device %d, the context to device %d", h->device, ctx->device);
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    const int64_t Z = h->N[0], Y = h->N[1], X = h->N[2], V = Z * Y * X;
    hipStream_t s = ctx->stream;
    const float* d = in;
    if (in == out) {  // the estimate overwrites `out`; keep the data term
        float* dcopy;
        BH_TRY(get_scratch(ctx, "rl_data", V * sizeof(float), (void**)&dcopy));
        BH_CHECK_HIP(hipMemcpyAsync(dcopy, in, V * sizeof(float), hipMemcpyDeviceToDevice, s));
        d = dcopy;
    }
    if (h->backend == BH_RL_LIBRARY)
        return bh_richardson_lucy(ctx, d, h->psf, h->K[0], h->K[1], h->K[2], Z, Y, X, iterations, eps, out);
    if (iterations == 0) {
        hipLaunchKernelGGL(clip_copy_kernel, grid_for(ctx, V), dim3(256), 0, s, d, out, V);
        BH_CHECK_HIP(hipGetLastError());
        return BH_OK;
    }
    ScopedTimer timer(ctx, T_RL_TOTAL);
    if (h->backend == BH_RL_ENGINE) {
        // the last update pass can leave the row sums of its result behind (what a deskew with a mean fill wants of this volume)
        if (row_sums) fftconv_arm_rowsums(*h->plan, row_sums);
        const int st = rl_engine_run(ctx, h->plan, d, h->otf, h->otf_real, iterations, eps, out);
        if (row_sums) *produced = fftconv_rowsums_taken(*h->plan) && st == BH_OK ? 1 : 0;
        return st;
    }
    return rl_padded_run(ctx, h->plan, d, h->otf, h->otf_real, h->N, h->K, h->box, iterations, eps, out);
}

int bh_richardson_lucy_destroy(bh_rl* h) {
    if (!h) return BH_OK;
    if (h->otf) filter_pool_give(h->device, h->otf_bytes, h->otf);
    if (h->psf) (void)hipFree(h->psf);
    delete h;
    return BH_OK;
}

int bh_richardson_lucy_info(const bh_rl* h, int64_t box[3], int* backend, int* otf_is_real, uint64_t* otf_bytes) {
    BH_REQUIRE(h != nullptr, "NULL argument");
    if (box)
        for (int a = 0; a < 3; ++a) box[a] = h->box[a];
    if (backend) *backend = h->backend;
    if (otf_is_real) *otf_is_real = h->otf_real ? 1 : 0;
    if (otf_bytes) *otf_bytes = (uint64_t)h->otf_bytes;
    return BH_OK;
}

int bh_richardson_lucy(bh_ctx* ctx, const float* in, const float* psf, int64_t pz, int64_t py, int64_t px, int64_t Z,
                       int64_t Y, int64_t X, int iterations, float eps, float* out) {
    BH_REQUIRE(ctx && in && psf && out, "NULL argument");
    BH_REQUIRE(iterations >= 0, "iterations must be >= 0");
    BH_REQUIRE(pz > 0 && py > 0 && px > 0 && pz <= Z && py <= Y && px <= X, "PSF must fit inside the volume");
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    const int64_t V = Z * Y * X, Xh = X / 2 + 1, NS = Z * Y * Xh;
    float *real, *dcopy = nullptr;
    hipStream_t s = ctx->stream;
    const float* d = in;
    if (in == out) {  // the estimate overwrites `out`; keep the data term
        BH_TRY(get_scratch(ctx, "rl_data", V * sizeof(float), (void**)&dcopy));
        BH_CHECK_HIP(hipMemcpyAsync(dcopy, in, V * sizeof(float), hipMemcpyDeviceToDevice, s));
        d = dcopy;
    }
    int64_t box[3];
    const int backend = rl_plan(pz, py, px, Z, Y, X, box);
    if (backend == BH_RL_ENGINE) return richardson_lucy_fused(ctx, d, psf, pz, py, px, Z, Y, X, iterations, eps, out);
    if (backend == BH_RL_ENGINE_PADDED)
        return richardson_lucy_engine_padded(ctx, d, psf, pz, py, px, Z, Y, X, box, iterations, eps, out);
    if (box[0] != Z || box[1] != Y || box[2] != X)
        return richardson_lucy_padfold(ctx, d, psf, pz, py, px, Z, Y, X, box, iterations, eps, out);
    FftPlans* pl;
    BH_TRY(get_plans(ctx, Z, Y, X, &pl));
    cf *spec, *otf;
    double* psum;
    BH_TRY(get_scratch(ctx, "fft_real", V * sizeof(float), (void**)&real));
    BH_TRY(get_scratch(ctx, "fft_spec", NS * sizeof(cf), (void**)&spec));
    BH_TRY(get_scratch(ctx, "rl_otf", NS * sizeof(cf), (void**)&otf));
    BH_TRY(get_scratch(ctx, "rl_psum", 64, (void**)&psum));
    ScopedTimer timer(ctx, T_RL_TOTAL);
    // OTF = rfftn(roll(pad(psf / sum), -centre)) / V   (the 1/V makes every C2R normalised)
    BH_CHECK_HIP(hipMemsetAsync(real, 0, V * sizeof(float), s));
    hipLaunchKernelGGL(psf_sum_kernel, dim3(1), dim3(256), 0, s, psf, pz * py * px, psum);
    int bz, by, bx;
    pad_before(pz, Z, &bz);
    pad_before(py, Y, &by);
    pad_before(px, X, &bx);
    hipLaunchKernelGGL(place_psf_kernel, grid_for(ctx, pz * py * px), dim3(256), 0, s, psf, real, (int)pz, (int)py,
                       (int)px, Z, Y, X, bz, by, bx, bz + (int)(pz / 2), by + (int)(py / 2), bx + (int)(px / 2),
                       (const double*)psum);
    BH_TRY(fft_forward(pl, real, otf));
    hipLaunchKernelGGL(scale_spectrum_kernel, grid_for(ctx, NS), dim3(256), 0, s, otf, NS, (float)(1.0 / (double)V));
    // e0 = max(d, 0)
    hipLaunchKernelGGL(clip_copy_kernel, grid_for(ctx, V), dim3(256), 0, s, d, out, V);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ctx->timing && iterations > 0) {
        BH_CHECK_HIP(hipEventCreate(&e0));
        BH_CHECK_HIP(hipEventCreate(&e1));
        BH_CHECK_HIP(hipEventRecord(e0, s));
    }
    const int64_t n2 = NS / 2;
    for (int it = 0; it < iterations; ++it) {
        BH_TRY(fft_forward(pl, out, spec));
        hipLaunchKernelGGL(cmul_kernel<false>, grid_for(ctx, n2), dim3(256), 0, s, spec, otf, n2);
        if (NS & 1) hipLaunchKernelGGL(cmul_tail_kernel<false>, dim3(1), dim3(1), 0, s, spec, otf, NS - 1);
        BH_TRY(fft_inverse(pl, spec, real));
        hipLaunchKernelGGL(ratio_kernel, grid_for(ctx, V / 4 + 1), dim3(256), 0, s, real, d, V, eps);
        BH_TRY(fft_forward(pl, real, spec));
        hipLaunchKernelGGL(cmul_kernel<true>, grid_for(ctx, n2), dim3(256), 0, s, spec, otf, n2);
        if (NS & 1) hipLaunchKernelGGL(cmul_tail_kernel<true>, dim3(1), dim3(1), 0, s, spec, otf, NS - 1);
        BH_TRY(fft_inverse(pl, spec, real));
        hipLaunchKernelGGL(update_kernel, grid_for(ctx, V / 4 + 1), dim3(256), 0, s, out, real, V);
    }
    BH_CHECK_HIP(hipGetLastError());
    if (e0) {
        BH_CHECK_HIP(hipEventRecord(e1, s));
        BH_CHECK_HIP(hipEventSynchronize(e1));
        float ms = 0;
        BH_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
        ctx->ms_override[T_RL_ITER] = ms / iterations;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    return BH_OK;
}

}  // extern "C"
