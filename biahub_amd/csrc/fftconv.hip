// Fused 3-D real FFT convolution engine for gfx950 (power-of-two volumes) — the Richardson-Lucy hot loop.
//
// hipFFT runs a 3-D R2C/C2R of a 512x2048x2048 volume as 5-7 kernels (row FFTs + transposes + a slow
// strided z kernel), and every pointwise step between transforms is one more pass over HBM:
// ~29 ms forward + ~23 ms inverse + 4 x 5.4 ms pointwise per convolution pair (profiles/r01a_*).
// A convolution does not need the spectrum in natural order or natural layout, only a forward and an
// inverse that agree with the transformed kernel (OTF).  So this engine
//   * runs each axis as ONE in-place pass (no transposes), leaving every axis in the scrambled order its
//     decimation-in-frequency FFT produces (bit-reversed; Y additionally split even/odd by a radix-2 step
//     that is folded into the X pass so the Y pass fits LDS with 128-B row segments);
//   * fuses forward-Z, the OTF multiply and inverse-Z into one kernel (a tile is loaded once);
//   * fuses the real<->complex packing, and Richardson-Lucy's divide / multiply-clip, into the X passes.
// One convolution = 5 passes, 48 B/voxel of HBM traffic (two per R-L iteration: 96 B/voxel, against the
// 112 B/voxel of the 3-pass-per-FFT model and ~260 B/voxel measured for the hipFFT path).
//
// Layout of the half spectrum: S[z][y][p], p in [0, XP), XP = X/2 + 16 complex per row (128-B aligned
// column tiles; column X/2 holds the Nyquist bin, the remaining pad columns stay zero).
//
// Each pass is a persistent kernel (one 1024-thread workgroup per CU) that walks 128-KiB tiles:
// registers prefetch tile t+1 from HBM while the FFT of tile t runs out of LDS.
#include "common.hpp"

#include <cstdio>
#include <cstdlib>
#include <mutex>

#include <cmath>
#include <vector>

namespace bh {

typedef float2 cf;

#ifndef BH_FC_NT
#define BH_FC_NT 1024
#endif
constexpr int FC_NT = BH_FC_NT;    // threads per workgroup
constexpr int FC_TILE = 16384;     // complex elements per column tile (128 KiB)
#ifndef BH_FC_XR
#define BH_FC_XR 16
#endif
#ifndef BH_FC_XNT
#define BH_FC_XNT BH_FC_NT
#endif
#ifndef BH_FC_XNT8
#define BH_FC_XNT8 768  // threads per workgroup of the 8-row X passes (rows of 3072 voxels: one thread per float4 of a row;
                        // 1.40 s against 1.44 s with 1024 threads for R-L x10 at the box (768,2048,3072))
#endif
#ifndef BH_FC_R16
#define BH_FC_R16 0
#endif
constexpr bool FC_R16 = BH_FC_R16 != 0;
#ifndef BH_FC_XR16
#define BH_FC_XR16 BH_FC_R16  // the same choice for the row transforms of the X passes alone
#endif
constexpr bool FC_XR16 = BH_FC_XR16 != 0;

__device__ __forceinline__ cf cadd(cf a, cf b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cf csub(cf a, cf b) { return make_float2(a.x - b.x, a.y - b.y); }
// Complex products as TWO packed instructions (v_pk_mul_f32 + v_pk_fma_f32 with op_sel / neg modifiers picking the halves):
// the compiler's own lowering spends four to six instructions on them, a quarter of the arithmetic of the register FFT stages
// being v_mov shuffles that line operands up for packed adds.  BH_FC_PK_CMUL=0 keeps the plain C form (A/B switch).
#ifndef BH_FC_PK_CMUL
#define BH_FC_PK_CMUL 1
#endif
typedef float v2f_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cf cmul(cf a, cf b) {
#if BH_FC_PK_CMUL
    v2f_t av = {a.x, a.y}, bv = {b.x, b.y}, t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(av), "v"(bv));                     // (a.x b.x, a.x b.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(av), "v"(bv), "v"(t));  // (-a.y b.y, a.y b.x) + t
    return make_float2(r.x, r.y);
#else
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
#endif
}
__device__ __forceinline__ cf cmulc(cf a, cf b) {  // a * conj(b)
#if BH_FC_PK_CMUL
    v2f_t av = {a.x, a.y}, bv = {b.x, b.y}, t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(av), "v"(bv));         // (a.x b.x, -a.x b.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(av), "v"(bv), "v"(t));      // (a.y b.y, a.y b.x) + t
    return make_float2(r.x, r.y);
#else
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
#endif
}
// a - i b and a + i b as ONE packed add each: v_pk_add_f32 takes either half of each source for each half of the result and can
// negate it.  The compiler's own lowering materialises (-i) b with two moves first — a fifth of the register FFT stages'
// instructions were such moves (fftconv_xw.inc reg_fft, fftconv_colz.inc fwd8p / inv8p fold every rotation by -i / +i into
// the add or subtract that consumes it).
__host__ __device__ __forceinline__ cf add_mi(cf a, cf b) {
#if defined(__HIP_DEVICE_COMPILE__)
    v2f_t av = {a.x, a.y}, bv = {b.x, b.y}, r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(av), "v"(bv));
    return make_float2(r.x, r.y);
#else
    return make_float2(a.x + b.y, a.y - b.x);
#endif
}
__host__ __device__ __forceinline__ cf add_pi(cf a, cf b) {
#if defined(__HIP_DEVICE_COMPILE__)
    v2f_t av = {a.x, a.y}, bv = {b.x, b.y}, r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(av), "v"(bv));
    return make_float2(r.x, r.y);
#else
    return make_float2(a.x - b.y, a.y + b.x);
#endif
}
__device__ __forceinline__ cf cconj(cf a) { return make_float2(a.x, -a.y); }
__device__ __forceinline__ cf mul_mi(cf a) { return make_float2(a.y, -a.x); }  // a * (-i)
__device__ __forceinline__ cf mul_pi(cf a) { return make_float2(-a.y, a.x); }  // a * (+i)
__device__ __forceinline__ cf cscale(cf a, float s) { return make_float2(a.x * s, a.y * s); }

// ------------------------------------------------------------------------------------------------
// In-LDS, in-place FFT of W interleaved columns: element (n, c) at buf[n * P + c].
// Forward = decimation in frequency, natural in -> bit-reversed out.  Inverse = the mirrored
// decimation in time with conjugate twiddles, bit-reversed in -> natural out, unnormalised (x N).
// Radix-4 steps are two fused radix-2 levels; an odd log2(N) adds one radix-2 step (first fwd / last inv).
// Twiddle table (see make_twiddles): [radix-2: w_N^j, j < N/2 (only if log2 N odd)] then for each
// radix-4 step of half-size h (descending) and j < h/2: w_2h^j, w_2h^2j, w_2h^3j.
// ------------------------------------------------------------------------------------------------
// CPT = complex columns per thread (1: float2 accesses, any pitch; 2: float4 accesses, pitch even and
// 16-B aligned).  BPT = butterflies per thread and step, fully unrolled so that every LDS read of a step
// is in flight before the first butterfly is computed.
template <int CPT>
struct CV;
template <>
struct CV<1> {
    cf a;
    __device__ __forceinline__ static CV ld(const cf* p) { return CV{*p}; }
    __device__ __forceinline__ void st(cf* p) const { *p = a; }
};
template <>
struct CV<2> {
    cf a, b;
    __device__ __forceinline__ static CV ld(const cf* p) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        return CV{make_float2(v.x, v.y), make_float2(v.z, v.w)};
    }
    __device__ __forceinline__ void st(cf* p) const { *reinterpret_cast<float4*>(p) = make_float4(a.x, a.y, b.x, b.y); }
};

#define BH_CV_OP1(name, f)                                                                  \
    template <int CPT>                                                                      \
    __device__ __forceinline__ CV<CPT> name(const CV<CPT>& x);                              \
    template <>                                                                             \
    __device__ __forceinline__ CV<1> name<1>(const CV<1>& x) { return CV<1>{f(x.a)}; }      \
    template <>                                                                             \
    __device__ __forceinline__ CV<2> name<2>(const CV<2>& x) { return CV<2>{f(x.a), f(x.b)}; }
#define BH_CV_OP2(name, f)                                                                                    \
    template <int CPT>                                                                                        \
    __device__ __forceinline__ CV<CPT> name(const CV<CPT>& x, const CV<CPT>& y);                              \
    template <>                                                                                               \
    __device__ __forceinline__ CV<1> name<1>(const CV<1>& x, const CV<1>& y) { return CV<1>{f(x.a, y.a)}; }   \
    template <>                                                                                               \
    __device__ __forceinline__ CV<2> name<2>(const CV<2>& x, const CV<2>& y) {                                \
        return CV<2>{f(x.a, y.a), f(x.b, y.b)};                                                               \
    }
#define BH_CV_OPT(name, f)                                                                             \
    template <int CPT>                                                                                 \
    __device__ __forceinline__ CV<CPT> name(const CV<CPT>& x, cf t);                                   \
    template <>                                                                                        \
    __device__ __forceinline__ CV<1> name<1>(const CV<1>& x, cf t) { return CV<1>{f(x.a, t)}; }        \
    template <>                                                                                        \
    __device__ __forceinline__ CV<2> name<2>(const CV<2>& x, cf t) { return CV<2>{f(x.a, t), f(x.b, t)}; }
BH_CV_OP2(vadd, cadd)
BH_CV_OP2(vsub, csub)
BH_CV_OP1(vmul_mi, mul_mi)
BH_CV_OP1(vmul_pi, mul_pi)
BH_CV_OPT(vmul, cmul)
BH_CV_OPT(vmulc, cmulc)
#undef BH_CV_OP1
#undef BH_CV_OP2
#undef BH_CV_OPT

template <bool INV, int BPT, int CPT, int NT = FC_NT>
__device__ __forceinline__ void radix4_step(cf* buf, int N, int logW, int P, int h, const cf* t, int tid) {
    const int q = h >> 1;
    const int lw = logW - (CPT == 2 ? 1 : 0);       // log2 of column groups per row
    const int total = (N >> 2) << lw;
    const size_t qP = (size_t)q * P;
    // all threads run the same number of groups; ragged tails clamp the index and skip the store
    const bool ragged = (total % (BPT * NT)) != 0;
    for (int g0 = 0; g0 < total; g0 += BPT * NT) {
    CV<CPT> x0[BPT], x1[BPT], x2[BPT], x3[BPT];
    cf t1[BPT], t2[BPT], t3[BPT];
    cf* p0[BPT];
#pragma unroll
    for (int k = 0; k < BPT; ++k) {
        const int idx = min(g0 + tid + k * NT, total - 1);
        const int c = (idx & ((1 << lw) - 1)) * CPT;
        const int b = idx >> lw;
        const int j = b & (q - 1);
        const int i = ((b - j) << 2) + j;  // (b / q) * 2h + j
        p0[k] = buf + (size_t)i * P + c;
        x0[k] = CV<CPT>::ld(p0[k]);
        x1[k] = CV<CPT>::ld(p0[k] + qP);
        x2[k] = CV<CPT>::ld(p0[k] + 2 * qP);
        x3[k] = CV<CPT>::ld(p0[k] + 3 * qP);
        t1[k] = t[3 * j];
        t2[k] = t[3 * j + 1];
        t3[k] = t[3 * j + 2];
    }
    if (ragged) __syncthreads();  // clamped duplicates must all read before anyone writes
#pragma unroll
    for (int k = 0; k < BPT; ++k) {
        if (g0 + tid + k * NT < total) {
            if (!INV) {
                const CV<CPT> s02 = vadd<CPT>(x0[k], x2[k]), d02 = vsub<CPT>(x0[k], x2[k]);
                const CV<CPT> s13 = vadd<CPT>(x1[k], x3[k]), d13 = vmul_mi<CPT>(vsub<CPT>(x1[k], x3[k]));
                vadd<CPT>(s02, s13).st(p0[k]);
                vmul<CPT>(vsub<CPT>(s02, s13), t2[k]).st(p0[k] + qP);
                vmul<CPT>(vadd<CPT>(d02, d13), t1[k]).st(p0[k] + 2 * qP);
                vmul<CPT>(vsub<CPT>(d02, d13), t3[k]).st(p0[k] + 3 * qP);
            } else {
                const CV<CPT> u1 = vmulc<CPT>(x1[k], t2[k]), u2 = vmulc<CPT>(x2[k], t1[k]), u3 = vmulc<CPT>(x3[k], t3[k]);
                const CV<CPT> A = vadd<CPT>(x0[k], u1), B = vsub<CPT>(x0[k], u1);
                const CV<CPT> C = vadd<CPT>(u2, u3), D = vmul_pi<CPT>(vsub<CPT>(u2, u3));
                vadd<CPT>(A, C).st(p0[k]);
                vsub<CPT>(A, C).st(p0[k] + 2 * qP);
                vadd<CPT>(B, D).st(p0[k] + qP);
                vsub<CPT>(B, D).st(p0[k] + 3 * qP);
            }
        }
    }
    if (ragged) __syncthreads();
    }
}

// `rows` >= N: the buffer holds rows / N independent length-N sequences one after the other (see fft_lds)
template <bool INV, int BPT, int CPT, int NT = FC_NT>
__device__ __forceinline__ void radix2_step(cf* buf, int N, int logW, int P, const cf* t, int tid, int rows) {
    const int h = N >> 1;
    const int lw = logW - (CPT == 2 ? 1 : 0);
    const int total = (rows >> 1) << lw;
    const size_t hP = (size_t)h * P;
    constexpr int B2 = 2 * BPT;  // a radix-2 step has twice the butterflies of a radix-4 step
    const bool ragged = (total % (B2 * NT)) != 0;
    for (int g0 = 0; g0 < total; g0 += B2 * NT) {
    CV<CPT> a[B2], b[B2];
    cf w[B2];
    cf* pa[B2];
#pragma unroll
    for (int k = 0; k < B2; ++k) {
        const int idx = min(g0 + tid + k * NT, total - 1);
        const int c = (idx & ((1 << lw) - 1)) * CPT;
        const int jj = idx >> lw;
        const int j = jj & (h - 1);                 // butterfly within its sequence
        pa[k] = buf + (size_t)(((jj - j) << 1) + j) * P + c;
        a[k] = CV<CPT>::ld(pa[k]);
        b[k] = CV<CPT>::ld(pa[k] + hP);
        w[k] = t[j];
    }
    if (ragged) __syncthreads();
#pragma unroll
    for (int k = 0; k < B2; ++k) {
        if (g0 + tid + k * NT < total) {
            if (!INV) {
                vadd<CPT>(a[k], b[k]).st(pa[k]);
                vmul<CPT>(vsub<CPT>(a[k], b[k]), w[k]).st(pa[k] + hP);
            } else {
                const CV<CPT> ub = vmulc<CPT>(b[k], w[k]);
                vadd<CPT>(a[k], ub).st(pa[k]);
                vsub<CPT>(a[k], ub).st(pa[k] + hP);
            }
        }
    }
    if (ragged) __syncthreads();
    }
}

// One radix-4 butterfly in registers (same arithmetic and leg order as radix4_step).
template <bool INV>
__device__ __forceinline__ void bfly4(cf& x0, cf& x1, cf& x2, cf& x3, cf t1, cf t2, cf t3) {
    if (!INV) {
        const cf s02 = cadd(x0, x2), d02 = csub(x0, x2);
        const cf s13 = cadd(x1, x3), d13 = mul_mi(csub(x1, x3));
        x0 = cadd(s02, s13);
        x1 = cmul(csub(s02, s13), t2);
        x2 = cmul(cadd(d02, d13), t1);
        x3 = cmul(csub(d02, d13), t3);
    } else {
        const cf u1 = cmulc(x1, t2), u2 = cmulc(x2, t1), u3 = cmulc(x3, t3);
        const cf A = cadd(x0, u1), B = csub(x0, u1);
        const cf C = cadd(u2, u3), D = mul_pi(csub(u2, u3));
        x0 = cadd(A, C);
        x2 = csub(A, C);
        x1 = cadd(B, D);
        x3 = csub(B, D);
    }
}

// Radix-16 step = the two radix-4 steps of half-sizes h and h/4 fused in registers: 16 legs spaced h/8, one
// LDS round trip and one barrier instead of two.  Forward runs step h then h/4, inverse the mirror image.
// tA / tB are the twiddle tables of the radix-4 steps h and h/4.
template <bool INV, int NT = FC_NT>
__device__ __forceinline__ void radix16_step(cf* buf, int N, int logW, int P, int h, const cf* tA, const cf* tB, int tid) {
    const int qB = h >> 3;
    const int W = 1 << logW;
    const int total = (N >> 4) << logW;
    const bool ragged = (total % NT) != 0;
#pragma unroll 1
    for (int g0 = 0; g0 < total; g0 += NT) {
        const int idx = min(g0 + tid, total - 1);
        const int c = idx & (W - 1);
        const int b = idx >> logW;
        const int j = b & (qB - 1);
        const int i0 = ((b - j) << 4) + j;  // (b / qB) * 2h + j
        cf* base = buf + (size_t)i0 * P + c;
        const size_t st = (size_t)qB * P;
        cf x[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = base[k * st];
        if (ragged) __syncthreads();
        if (!INV) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const cf* t = tA + 3 * (j + a * qB);
                bfly4<false>(x[a], x[a + 4], x[a + 8], x[a + 12], t[0], t[1], t[2]);
            }
            const cf u1 = tB[3 * j], u2 = tB[3 * j + 1], u3 = tB[3 * j + 2];
#pragma unroll
            for (int m = 0; m < 4; ++m) bfly4<false>(x[4 * m], x[4 * m + 1], x[4 * m + 2], x[4 * m + 3], u1, u2, u3);
        } else {
            const cf u1 = tB[3 * j], u2 = tB[3 * j + 1], u3 = tB[3 * j + 2];
#pragma unroll
            for (int m = 0; m < 4; ++m) bfly4<true>(x[4 * m], x[4 * m + 1], x[4 * m + 2], x[4 * m + 3], u1, u2, u3);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const cf* t = tA + 3 * (j + a * qB);
                bfly4<true>(x[a], x[a + 4], x[a + 8], x[a + 12], t[0], t[1], t[2]);
            }
        }
        if (g0 + tid < total) {
#pragma unroll
            for (int k = 0; k < 16; ++k) base[k * st] = x[k];
        }
        if (ragged) __syncthreads();
    }
}

// Radix-3 step for a column of N = 3 L rows (L a power of two), two complex columns per thread.  Forward (decimation in
// frequency): rows (m, m + L, m + 2L) -> the three length-L sequences y_k[m] = (x[m] + w3^k x[m+L] + w3^2k x[m+2L]) w_N^(k m),
// left in rows [k L, (k + 1) L); their length-L transforms are X[3 j + k].  Inverse: the mirror image with conjugate
// twiddles, unnormalised.  t3[2 m] = w_N^m, t3[2 m + 1] = w_N^2m.
template <int CPT>
__device__ __forceinline__ CV<CPT> vscale(const CV<CPT>& x, float f);
template <>
__device__ __forceinline__ CV<1> vscale<1>(const CV<1>& x, float f) { return CV<1>{make_float2(x.a.x * f, x.a.y * f)}; }
template <>
__device__ __forceinline__ CV<2> vscale<2>(const CV<2>& x, float f) {
    return CV<2>{make_float2(x.a.x * f, x.a.y * f), make_float2(x.b.x * f, x.b.y * f)};
}
template <bool INV, int CPT = 2, int NT = FC_NT>
__device__ __forceinline__ void radix3_step(cf* buf, int L, int logW, int P, const cf* t3, int tid) {
    const int lw = logW - (CPT == 2 ? 1 : 0);
    const int total = L << lw;
    const size_t LP = (size_t)L * P;
    const float S3 = 0.86602540378443865f;
    for (int idx = tid; idx < total; idx += NT) {
        const int c = (idx & ((1 << lw) - 1)) * CPT;
        const int m = idx >> lw;
        cf* p0 = buf + (size_t)m * P + c;
        const CV<CPT> a = CV<CPT>::ld(p0);
        CV<CPT> b = CV<CPT>::ld(p0 + LP), cc = CV<CPT>::ld(p0 + 2 * LP);
        const cf w1 = t3[2 * m], w2 = t3[2 * m + 1];
        if (INV) {
            b = vmulc<CPT>(b, w1);
            cc = vmulc<CPT>(cc, w2);
        }
        const CV<CPT> sm = vadd<CPT>(b, cc), df = vsub<CPT>(b, cc);
        const CV<CPT> base = vsub<CPT>(a, vscale<CPT>(sm, 0.5f));
        const CV<CPT> rot = vscale<CPT>(INV ? vmul_pi<CPT>(df) : vmul_mi<CPT>(df), S3);
        vadd<CPT>(a, sm).st(p0);
        if (!INV) {
            vmul<CPT>(vadd<CPT>(base, rot), w1).st(p0 + LP);
            vmul<CPT>(vsub<CPT>(base, rot), w2).st(p0 + 2 * LP);
        } else {
            vadd<CPT>(base, rot).st(p0 + LP);
            vsub<CPT>(base, rot).st(p0 + 2 * LP);
        }
    }
}

// Radix-5 step, same conventions: rows (m, m + L, .., m + 4L) <-> the five length-L sequences y_k[m] = (sum_j x[m + jL] w5^jk) w_N^(k m)
// in rows [k L, (k + 1) L); t5[4 m + k - 1] = w_N^(k m), k = 1..4.
template <bool INV, int CPT = 2, int NT = FC_NT>
__device__ __forceinline__ void radix5_step(cf* buf, int L, int logW, int P, const cf* t5, int tid) {
    const int lw = logW - (CPT == 2 ? 1 : 0);
    const int total = L << lw;
    const size_t LP = (size_t)L * P;
    const float C1 = 0.30901699437494742f, C2 = -0.80901699437494742f;  // cos(2 pi / 5), cos(4 pi / 5)
    const float S1 = 0.95105651629515357f, S2 = 0.58778525229247313f;   // sin(2 pi / 5), sin(4 pi / 5)
    for (int idx = tid; idx < total; idx += NT) {
        const int c = (idx & ((1 << lw) - 1)) * CPT;
        const int m = idx >> lw;
        cf* p0 = buf + (size_t)m * P + c;
        const CV<CPT> x0 = CV<CPT>::ld(p0);
        CV<CPT> x1 = CV<CPT>::ld(p0 + LP), x2 = CV<CPT>::ld(p0 + 2 * LP), x3 = CV<CPT>::ld(p0 + 3 * LP), x4 = CV<CPT>::ld(p0 + 4 * LP);
        const cf w1 = t5[4 * m], w2 = t5[4 * m + 1], w3 = t5[4 * m + 2], w4 = t5[4 * m + 3];
        if (INV) {
            x1 = vmulc<CPT>(x1, w1);
            x2 = vmulc<CPT>(x2, w2);
            x3 = vmulc<CPT>(x3, w3);
            x4 = vmulc<CPT>(x4, w4);
        }
        const CV<CPT> t1 = vadd<CPT>(x1, x4), t2 = vadd<CPT>(x2, x3), t3 = vsub<CPT>(x1, x4), t4 = vsub<CPT>(x2, x3);
        const CV<CPT> a1 = vadd<CPT>(x0, vadd<CPT>(vscale<CPT>(t1, C1), vscale<CPT>(t2, C2)));
        const CV<CPT> a2 = vadd<CPT>(x0, vadd<CPT>(vscale<CPT>(t1, C2), vscale<CPT>(t2, C1)));
        const CV<CPT> b1 = vadd<CPT>(vscale<CPT>(t3, S1), vscale<CPT>(t4, S2));
        const CV<CPT> b2 = vsub<CPT>(vscale<CPT>(t3, S2), vscale<CPT>(t4, S1));
        const CV<CPT> r1 = INV ? vmul_pi<CPT>(b1) : vmul_mi<CPT>(b1), r2 = INV ? vmul_pi<CPT>(b2) : vmul_mi<CPT>(b2);
        vadd<CPT>(x0, vadd<CPT>(t1, t2)).st(p0);
        if (!INV) {
            vmul<CPT>(vadd<CPT>(a1, r1), w1).st(p0 + LP);
            vmul<CPT>(vadd<CPT>(a2, r2), w2).st(p0 + 2 * LP);
            vmul<CPT>(vsub<CPT>(a2, r2), w3).st(p0 + 3 * LP);
            vmul<CPT>(vsub<CPT>(a1, r1), w4).st(p0 + 4 * LP);
        } else {
            vadd<CPT>(a1, r1).st(p0 + LP);
            vadd<CPT>(a2, r2).st(p0 + 2 * LP);
            vsub<CPT>(a2, r2).st(p0 + 3 * LP);
            vsub<CPT>(a1, r1).st(p0 + 4 * LP);
        }
    }
}

// the odd first (forward) / last (inverse) step of an axis of rdx * 2^k, rdx = 3 or 5
template <bool INV, int CPT, int NT, int RDX>
__device__ __forceinline__ void odd_step(cf* buf, int L, int logW, int P, const cf* t, int tid) {
    if (RDX == 3) radix3_step<INV, CPT, NT>(buf, L, logW, P, t, tid);
    if (RDX == 5) radix5_step<INV, CPT, NT>(buf, L, logW, P, t, tid);
}

// R16: pair consecutive radix-4 steps into radix-16 steps (one column per thread); a leftover radix-4 step and
// the radix-2 step of an odd log2(N) use <BPT, CPT>.
// SKIP2: leave out the h = 2 radix-4 step (last forward / first inverse; its twiddles are all 1) — the convolution
// passes run it fused with the spectral multiply in registers (conv_mid_step).
// rows: total rows in the buffer when it holds several length-N sequences back to back (the thirds of a 3 * 2^k column
// after radix3_step); every step then runs over all of them at once — the step functions take their butterfly count from
// `rows` and their geometry from the half-size h.  0 = one sequence.
template <bool INV, int BPT, int CPT, bool R16 = false, bool SKIP2 = false, int NT = FC_NT>
__device__ __forceinline__ void fft_lds(cf* buf, int N, int logN, int logW, int P, const cf* tw, int tid, int rows = 0) {
    if (rows == 0) rows = N;
    const bool odd = logN & 1;
    const int H0 = odd ? (N >> 2) : (N >> 1);
    const cf* t4 = tw + (odd ? (N >> 1) : 0);
    const int L4 = (logN - (odd ? 1 : 0)) >> 1;   // radix-4 levels
    const int n16 = R16 ? (L4 >> 1) : 0;          // of which fused pairwise
    if (!INV) {
        if (odd) {
            radix2_step<false, BPT, CPT, NT>(buf, N, logW, P, tw, tid, rows);
            __syncthreads();
        }
        int h = H0;
        for (int s = 0; s < n16; ++s, h >>= 4) {
            radix16_step<false, NT>(buf, rows, logW, P, h, t4 + (2 * H0 - 2 * h), t4 + (2 * H0 - 2 * (h >> 2)), tid);
            __syncthreads();
        }
        for (; h >= (SKIP2 ? 8 : 2); h >>= 2) {
            radix4_step<false, BPT, CPT, NT>(buf, rows, logW, P, h, t4 + (2 * H0 - 2 * h), tid);
            __syncthreads();
        }
    } else {
        const int hr = H0 >> (4 * n16);  // largest half-size left to plain radix-4 steps
        for (int h = SKIP2 ? 8 : 2; h <= hr; h <<= 2) {
            radix4_step<true, BPT, CPT, NT>(buf, rows, logW, P, h, t4 + (2 * H0 - 2 * h), tid);
            __syncthreads();
        }
        for (int s = n16 - 1; s >= 0; --s) {
            const int h = H0 >> (4 * s);
            radix16_step<true, NT>(buf, rows, logW, P, h, t4 + (2 * H0 - 2 * h), t4 + (2 * H0 - 2 * (h >> 2)), tid);
            __syncthreads();
        }
        if (odd) {
            radix2_step<true, BPT, CPT, NT>(buf, N, logW, P, tw, tid, rows);
            __syncthreads();
        }
    }
}

static int twiddle_count(int N) {
    int logN = 0;
    while ((1 << logN) < N) ++logN;
    const bool odd = logN & 1;
    const int H0 = odd ? N / 4 : N / 2;
    int n = odd ? N / 2 : 0;
    for (int h = H0; h >= 2; h /= 4) n += 3 * (h / 2);
    return n;
}

static void make_twiddles(int N, std::vector<cf>& out) {
    int logN = 0;
    while ((1 << logN) < N) ++logN;
    const bool odd = logN & 1;
    const int H0 = odd ? N / 4 : N / 2;
    out.clear();
    if (odd)
        for (int j = 0; j < N / 2; ++j) {
            const double a = -2.0 * M_PI * j / N;
            out.push_back(make_float2((float)std::cos(a), (float)std::sin(a)));
        }
    for (int h = H0; h >= 2; h /= 4)
        for (int j = 0; j < h / 2; ++j)
            for (int m = 1; m <= 3; ++m) {
                const double a = -2.0 * M_PI * (double)j * m / (2.0 * h);
                out.push_back(make_float2((float)std::cos(a), (float)std::sin(a)));
            }
}

// position of frequency (M - k) when frequency k sits at bit-reversed position p
__device__ __forceinline__ int mirror_pos(int p) {
    if (p < 2) return p;
    const int top = 31 - __clz(p);
    return 3 * (1 << top) - 1 - p;
}

struct ConvDims {
    int Z, Y, X;   // real volume
    int M;         // X / 2 (complex FFT length along x)
    int XP;        // spectrum row pitch in complex elements
    int logM, logYh, logZ;  // log2 of M and of the power-of-two parts of Y/2 and Z
    int Lyh, Lz;            // those parts: Y/2 and Z themselves, or a third of them (radix-3 column passes)
    int Lm;                 // M or M / 3 (radix-3 first step of the row transforms)
};

// ================================================================================================
// Column passes (Y: two length-Y/2 halves per z; Z: fused forward x OTF x inverse)
// ================================================================================================
// COL_CONV16: COL_CONV with the multiplier stored as bfloat16 pairs (4 B per complex bin, widened in registers, f32 products)
// COL_PCC: the phase cross-correlation product in the Z pass — tile <- otf * conj(tile) / norm * scale between the forward and
// the inverse transform (otf = the reference image's finished spectrum; norm per ColParams::pcc_norm)
enum ColMode { COL_FWD = 0, COL_INV = 1, COL_FWD_SCALE = 2, COL_CONV = 3, COL_CORR = 4, COL_FILTER = 5, COL_CONV16 = 6, COL_PCC = 7 };

// one bin of the phase cross-correlation product (estimate_stabilization.py:233-238): a * conj(b) / norm * scale
__device__ __forceinline__ float2 pcc_bin(float2 a, float2 b, int mode, float scale) {
    const float eps = 1.1920929e-07f;  // np.finfo(complex64).eps
    float2 p = make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
    if (mode == BH_PCC_NORM_NONE) return make_float2(p.x * scale, p.y * scale);
    if (mode == BH_PCC_NORM_CLASSIC) {
        const float nrm = hypotf(a.x, a.y) * hypotf(b.x, b.y);
        return make_float2((p.x / nrm) * scale, (p.y / nrm) * scale);
    }
    // magnitude: p / max(|p|, eps).  |p|^2 leaves the float range for the low frequencies of a large volume, so p is brought to
    // q = p 2^-e with max(|q.x|, |q.y|) in [1/2, 1) first: p / |p| = q / |q| is one reciprocal square root (1 ulp) and two
    // products instead of hypotf and two correctly rounded divisions — a fifth of the instructions, in the Z pass whose
    // arithmetic showed (6.5 ms against 5.1 ms for the complex product on the same bytes)
    const float big = fmaxf(fabsf(p.x), fabsf(p.y));
    const int e = big > 0.0f ? __builtin_amdgcn_frexp_expf(big) : 0;
    const float qx = __builtin_amdgcn_ldexpf(p.x, -e), qy = __builtin_amdgcn_ldexpf(p.y, -e);
    const float qq = qx * qx + qy * qy;  // in [1/4, 2) unless p == 0
    const float rs = __builtin_amdgcn_rsqf(qq);
    const float mag = __builtin_amdgcn_ldexpf(qq * rs, e);  // |p| (inf beyond the float range: still >= eps)
    if (mag >= eps) return make_float2(qx * (rs * scale), qy * (rs * scale));
    const float s = scale / eps;
    return make_float2(p.x * s, p.y * s);
}

struct ColParams {
    cf* S;
    const cf* otf;
    const cf* tw;       // twiddles for length N
    int ntw;
    int N, logN, W, logW;  // column length (rows of the tile), log2 of its power-of-two part L, tile width (complex columns)
    int L;              // N (power of two) or N / 3: the radix-3 step splits a 3 * 2^k column into three length-L transforms
    const cf* tw3;      // radix-3 twiddles (2 L entries) when L != N
    int XP;             // valid columns per row
    long row_stride;    // complex elements between consecutive n
    long outer_stride;  // base(o) = (o / nsub) * outer_stride + (o % nsub) * sub_stride
    long sub_stride;
    int nsub;
    int nouter;         // number of o values
    int ncoltiles;
    float scale;
    int midfuse;        // fuse the unit-twiddle steps around the spectral product (BH_FC_NOZMID=1 turns it off)
    int pcc_norm;       // COL_PCC: BH_PCC_NORM_* of the product (`scale` multiplies it)
    int pcc_swap;       // COL_PCC: 0 = otf * conj(column) (otf holds the FIRST image's spectrum), 1 = column * conj(otf)
    cf* otf_out;        // COL_PCC, may be null: the column's forward spectrum replaces the multiplier rows it has just read
                        // (the image becomes the stored one for the next call: bh_phase_cross_corr_apply's `roll`)
};

// The prefetch registers are sixteen named float4, of which ROUNDS are used (not an array: hipcc keeps a loop-carried
// float4[] in scratch memory here even with every index constant).  BH_FOR8 applies a macro to all of them.
#define BH_FOR8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

// RDX: 1 for power-of-two columns, 3 / 5 for columns of 3 * 2^k / 5 * 2^k rows (their own instantiations: the odd step's
// registers would otherwise push the power-of-two kernels into scratch)
template <int MODE, int ROUNDS, int RDX = 1>
__global__ __launch_bounds__(FC_NT) void col_pass_kernel(ColParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* buf = reinterpret_cast<cf*>(smem);                         // [N][W]
    cf* tw = reinterpret_cast<cf*>(smem + (size_t)p.N * p.W * 8);  // twiddles
    const int tid = threadIdx.x;
    for (int i = tid; i < p.ntw; i += FC_NT) tw[i] = p.tw[i];
    const int L_ = RDX == 1 ? p.N : p.L;
    constexpr bool r3 = RDX != 1;
    cf* tw3 = tw + p.ntw;
    if (r3)
        for (int i = tid; i < (RDX - 1) * L_; i += FC_NT) tw3[i] = p.tw3[i];
    // column transform = [radix-3 step] + power-of-two transform of the 1 or 3 length-L sequences
#define BH_FFT_FWD(...)                                                     \
    {                                                                       \
        if (r3) {                                                           \
            odd_step<false, 2, FC_NT, RDX>(buf, L_, logW, W_, tw3, tid);                \
            __syncthreads();                                                \
        }                                                                   \
        fft_lds<false, __VA_ARGS__>(buf, L_, logN, logW, W_, tw, tid, N_);  \
    }
#define BH_FFT_INV(...)                                                     \
    {                                                                       \
        fft_lds<true, __VA_ARGS__>(buf, L_, logN, logW, W_, tw, tid, N_);   \
        if (r3) {                                                           \
            odd_step<true, 2, FC_NT, RDX>(buf, L_, logW, W_, tw3, tid);                 \
            __syncthreads();                                                \
        }                                                                   \
    }

    const int LPS = p.W >> 1;           // lanes per row segment (float4 = 2 complex)
    const int RPR = FC_NT / LPS;        // rows per round
    const int lane = tid % LPS;
    const int r0 = tid / LPS;
    const long ntiles = (long)p.nouter * p.ncoltiles;
    constexpr bool HAS_OTF = (MODE == COL_CONV || MODE == COL_CORR || MODE == COL_FILTER || MODE == COL_CONV16 || MODE == COL_PCC);
    const int ncoltiles = p.ncoltiles, nsub = p.nsub, W_ = p.W, N_ = p.N, logN = p.logN, logW = p.logW, XP = p.XP;
    const long outer_stride = p.outer_stride, sub_stride = p.sub_stride, row_stride = p.row_stride;
    cf* const S = p.S;
    const cf* const otf = p.otf;
    const float scale = p.scale;
    const int pcc_norm = p.pcc_norm, pcc_swap = p.pcc_swap;
    cf* const otf_out = p.otf_out;
    auto tile_base = [=](long tt) -> long {
        const long ou = tt / ncoltiles;
        const int ct = (int)(tt - ou * ncoltiles);
        return (ou / nsub) * outer_stride + (ou % nsub) * sub_stride + (long)ct * W_ + 2 * lane;
    };

    float4 v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11, v12, v13, v14, v15;
    v0 = v1 = v2 = v3 = v4 = v5 = v6 = v7 = make_float4(0.f, 0.f, 0.f, 0.f);
    v8 = v9 = v10 = v11 = v12 = v13 = v14 = v15 = v0;
    // unconditional, clamped row loads (see deskew.hip on predicated loads)
#define BH_LD(u) \
    if (u < ROUNDS) v##u = *reinterpret_cast<const float4*>(src_ + (long)min(r0 + u * RPR, N_ - 1) * row_stride);
#define BH_LOAD_TILE(SRC, T)                     \
    {                                            \
        const cf* src_ = (SRC) + tile_base(T);   \
        BH_FOR8(BH_LD)                           \
    }
    // real filter (Tikhonov): one float per complex element, same [z][y][p] indexing
#define BH_LDF(u)                                                                                          \
    if (u < ROUNDS) {                                                                                      \
        const float2 f_ = *reinterpret_cast<const float2*>(fsrc_ + (long)min(r0 + u * RPR, N_ - 1) * row_stride); \
        v##u = make_float4(f_.x, f_.x, f_.y, f_.y);                                                        \
    }
#define BH_LOAD_FILTER(T)                                                      \
    {                                                                          \
        const float* fsrc_ = reinterpret_cast<const float*>(otf) + tile_base(T); \
        BH_FOR8(BH_LDF)                                                        \
    }
    // complex multiplier stored as bfloat16 pairs (COL_CONV16): one 32-bit word per complex element
#define BH_LDH(u)                                                                                          \
    if (u < ROUNDS) {                                                                                      \
        const uint2 h_ = *reinterpret_cast<const uint2*>(hsrc_ + (long)min(r0 + u * RPR, N_ - 1) * row_stride); \
        v##u = make_float4(__uint_as_float(h_.x << 16), __uint_as_float(h_.x & 0xffff0000u),               \
                           __uint_as_float(h_.y << 16), __uint_as_float(h_.y & 0xffff0000u));              \
    }
#define BH_LOAD_FILTER16(T)                                                            \
    {                                                                                  \
        const unsigned int* hsrc_ = reinterpret_cast<const unsigned int*>(otf) + tile_base(T); \
        BH_FOR8(BH_LDH)                                                                \
    }
#define BH_TO_LDS(u)                                                                            \
    if (u < ROUNDS && r0 + u * RPR < N_)                                                        \
        *reinterpret_cast<float4*>(buf + (size_t)(r0 + u * RPR) * W_ + 2 * lane) = v##u;
#define BH_OTF_MUL(u)                                                                           \
    if (u < ROUNDS && r0 + u * RPR < N_) {                                                      \
        float4* q_ = reinterpret_cast<float4*>(buf + (size_t)(r0 + u * RPR) * W_ + 2 * lane);   \
        const float4 a = *q_;                                                                   \
        const float4 b = v##u;                                                                  \
        float4 c;                                                                               \
        if (MODE == COL_FILTER) {                                                               \
            c.x = a.x * b.x;                                                                    \
            c.y = a.y * b.y;                                                                    \
            c.z = a.z * b.z;                                                                    \
            c.w = a.w * b.w;                                                                    \
        } else if (MODE == COL_PCC) { /* first * conj(second) / norm * scale, as pcc_product_kernel; b = the stored spectrum */ \
            const float2 f0 = make_float2(pcc_swap ? a.x : b.x, pcc_swap ? a.y : b.y), s0 = make_float2(pcc_swap ? b.x : a.x, pcc_swap ? b.y : a.y); \
            const float2 f1 = make_float2(pcc_swap ? a.z : b.z, pcc_swap ? a.w : b.w), s1 = make_float2(pcc_swap ? b.z : a.z, pcc_swap ? b.w : a.w); \
            const float2 p0 = pcc_bin(f0, s0, pcc_norm, scale);                                 \
            const float2 p1 = pcc_bin(f1, s1, pcc_norm, scale);                                 \
            c = make_float4(p0.x, p0.y, p1.x, p1.y);                                            \
            if (otf_out && col_ok) *reinterpret_cast<float4*>(otf_out + base + (long)(r0 + u * RPR) * row_stride) = a; \
        } else if (MODE == COL_CONV || MODE == COL_CONV16) {                                    \
            c.x = a.x * b.x - a.y * b.y;                                                        \
            c.y = a.x * b.y + a.y * b.x;                                                        \
            c.z = a.z * b.z - a.w * b.w;                                                        \
            c.w = a.z * b.w + a.w * b.z;                                                        \
        } else {                                                                                \
            c.x = a.x * b.x + a.y * b.y;                                                        \
            c.y = a.y * b.x - a.x * b.y;                                                        \
            c.z = a.z * b.z + a.w * b.w;                                                        \
            c.w = a.w * b.z - a.z * b.w;                                                        \
        }                                                                                       \
        *q_ = c;                                                                                \
    }
#define BH_STORE(u)                                                                                        \
    if (u < ROUNDS && r0 + u * RPR < N_ && col_ok) {                                                       \
        float4 a = *reinterpret_cast<const float4*>(buf + (size_t)(r0 + u * RPR) * W_ + 2 * lane);         \
        if (MODE == COL_FWD_SCALE) {                                                                       \
            a.x *= scale;                                                                                  \
            a.y *= scale;                                                                                  \
            a.z *= scale;                                                                                  \
            a.w *= scale;                                                                                  \
        }                                                                                                  \
        *reinterpret_cast<float4*>(S + base + (long)(r0 + u * RPR) * row_stride) = a;                      \
    }

    long t = blockIdx.x;
    if (t < ntiles) BH_LOAD_TILE(S, t)
    for (; t < ntiles; t += gridDim.x) {
        BH_FOR8(BH_TO_LDS)  // registers -> LDS
        const long base = tile_base(t);
        const int ct = (int)(t % ncoltiles);
        const bool col_ok = (ct * W_ + 2 * lane) < XP;  // pad columns of a ragged last tile are never stored
        __syncthreads();
        const long tn = t + gridDim.x;
        if (HAS_OTF && MODE != COL_PCC && !FC_R16 && p.midfuse) {
            // This tile's OTF arrives behind the forward FFT, fetched in the order the fused middle step wants it:
            // butterfly b = tid / LPS + s * RPR (s = 0, 1) covers rows 4b .. 4b + 3 of this lane's two columns.
            // The h = 2 radix-4 steps at the end of the forward and the start of the inverse transform have unit
            // twiddles and touch the same four rows, so forward step, spectral multiply and inverse step happen in
            // registers: one LDS round trip and one barrier instead of three.
            const int nbf = N_ >> 2;  // butterflies per column
            {
                const long tb_ = tile_base(t);
#define BH_LDM(u)                                                                                                  \
    {                                                                                                              \
        const int row_ = min(4 * (r0 + (u >> 2) * RPR) + (u & 3), N_ - 1);                                         \
        if (MODE == COL_FILTER) {                                                                                  \
            const float2 f_ = *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(otf) + tb_ + (long)row_ * row_stride); \
            v##u = make_float4(f_.x, f_.x, f_.y, f_.y);                                                            \
        } else if (MODE == COL_CONV16) {                                                                           \
            const uint2 h_ = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned int*>(otf) + tb_ + (long)row_ * row_stride); \
            v##u = make_float4(__uint_as_float(h_.x << 16), __uint_as_float(h_.x & 0xffff0000u),                   \
                               __uint_as_float(h_.y << 16), __uint_as_float(h_.y & 0xffff0000u));                  \
        } else {                                                                                                   \
            v##u = *reinterpret_cast<const float4*>(otf + tb_ + (long)row_ * row_stride);                          \
        }                                                                                                          \
    }
                BH_LDM(0) BH_LDM(1) BH_LDM(2) BH_LDM(3) BH_LDM(4) BH_LDM(5) BH_LDM(6) BH_LDM(7)
#undef BH_LDM
            }
            BH_FFT_FWD(1, 2, false, true)
#define BH_SPEC_MUL(a, b)                                                                                          \
    (MODE == COL_FILTER ? make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w)                                  \
     : (MODE == COL_CONV || MODE == COL_CONV16) ? make_float4(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x, a.z * b.z - a.w * b.w, \
                                      a.z * b.w + a.w * b.z)                                                       \
                        : make_float4(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y, a.z * b.z + a.w * b.w,         \
                                      a.w * b.z - a.z * b.w))
#define BH_MID(S_, O0, O1, O2, O3)                                                                                 \
    if (r0 + S_ * RPR < nbf) {                                                                                     \
        float4* q_ = reinterpret_cast<float4*>(buf + (size_t)(4 * (r0 + S_ * RPR)) * W_ + 2 * lane);              \
        const int st_ = W_ >> 1; /* float4 units per row */                                                        \
        const float4 x0 = q_[0], x1 = q_[st_], x2 = q_[2 * st_], x3 = q_[3 * st_];                                 \
        /* forward h = 2 step, unit twiddles: rows 4b .. 4b + 3 <- s02+s13, s02-s13, d02+d13, d02-d13 */         \
        const float4 s02 = make_float4(x0.x + x2.x, x0.y + x2.y, x0.z + x2.z, x0.w + x2.w);                       \
        const float4 d02 = make_float4(x0.x - x2.x, x0.y - x2.y, x0.z - x2.z, x0.w - x2.w);                       \
        const float4 s13 = make_float4(x1.x + x3.x, x1.y + x3.y, x1.z + x3.z, x1.w + x3.w);                       \
        const float4 e13 = make_float4(x1.x - x3.x, x1.y - x3.y, x1.z - x3.z, x1.w - x3.w);                       \
        const float4 d13 = make_float4(e13.y, -e13.x, e13.w, -e13.z); /* * (-i) */                                 \
        const float4 f0 = make_float4(s02.x + s13.x, s02.y + s13.y, s02.z + s13.z, s02.w + s13.w);                 \
        const float4 f1 = make_float4(s02.x - s13.x, s02.y - s13.y, s02.z - s13.z, s02.w - s13.w);                 \
        const float4 f2 = make_float4(d02.x + d13.x, d02.y + d13.y, d02.z + d13.z, d02.w + d13.w);                 \
        const float4 f3 = make_float4(d02.x - d13.x, d02.y - d13.y, d02.z - d13.z, d02.w - d13.w);                 \
        const float4 y0 = BH_SPEC_MUL(f0, O0), y1 = BH_SPEC_MUL(f1, O1), y2 = BH_SPEC_MUL(f2, O2),                 \
                     y3 = BH_SPEC_MUL(f3, O3);                                                                     \
        /* inverse h = 2 step, unit twiddles: rows <- A+C, B+D, A-C, B-D */                                        \
        const float4 A = make_float4(y0.x + y1.x, y0.y + y1.y, y0.z + y1.z, y0.w + y1.w);                          \
        const float4 B = make_float4(y0.x - y1.x, y0.y - y1.y, y0.z - y1.z, y0.w - y1.w);                          \
        const float4 Cc = make_float4(y2.x + y3.x, y2.y + y3.y, y2.z + y3.z, y2.w + y3.w);                         \
        const float4 e23 = make_float4(y2.x - y3.x, y2.y - y3.y, y2.z - y3.z, y2.w - y3.w);                       \
        const float4 D = make_float4(-e23.y, e23.x, -e23.w, e23.z); /* * (+i) */                                   \
        q_[0] = make_float4(A.x + Cc.x, A.y + Cc.y, A.z + Cc.z, A.w + Cc.w);                                       \
        q_[st_] = make_float4(B.x + D.x, B.y + D.y, B.z + D.z, B.w + D.w);                                         \
        q_[2 * st_] = make_float4(A.x - Cc.x, A.y - Cc.y, A.z - Cc.z, A.w - Cc.w);                                 \
        q_[3 * st_] = make_float4(B.x - D.x, B.y - D.y, B.z - D.z, B.w - D.w);                                     \
    }
            BH_MID(0, v0, v1, v2, v3)
            BH_MID(1, v4, v5, v6, v7)
#undef BH_MID
#undef BH_SPEC_MUL
            __syncthreads();
            if (tn < ntiles) BH_LOAD_TILE(S, tn)
            BH_FFT_INV(1, 2, false, true)
        } else if (HAS_OTF) {
            // this tile's OTF arrives behind the forward FFT; the next tile's data behind the inverse FFT
            if (MODE == COL_FILTER) BH_LOAD_FILTER(t) else if (MODE == COL_CONV16) BH_LOAD_FILTER16(t) else BH_LOAD_TILE(otf, t)
            BH_FFT_FWD(1, 2, FC_R16)
            BH_FOR8(BH_OTF_MUL)
            __syncthreads();
            if (tn < ntiles) BH_LOAD_TILE(S, tn)
            BH_FFT_INV(1, 2, FC_R16)
        } else {
            if (tn < ntiles) BH_LOAD_TILE(S, tn)  // prefetch the next tile behind the FFT
            if (MODE == COL_INV) {
                BH_FFT_INV(1, 2, FC_R16)
            } else {
                BH_FFT_FWD(1, 2, FC_R16)
            }
        }
        BH_FOR8(BH_STORE)  // LDS -> global
        __syncthreads();
    }
#undef BH_FFT_FWD
#undef BH_FFT_INV
#undef BH_LD
#undef BH_LDF
#undef BH_LOAD_FILTER
#undef BH_LDH
#undef BH_LOAD_FILTER16
#undef BH_LOAD_TILE
#undef BH_TO_LDS
#undef BH_OTF_MUL
#undef BH_STORE
}

// ================================================================================================
// X passes: real rows <-> half-spectrum rows, with the Y radix-2 step across row pairs (y, y + Y/2)
// ================================================================================================
enum XEpilogue { XE_STORE = 0, XE_RATIO = 1, XE_UPDATE = 2 };

struct XParams {
    const float* in;      // forward: real input volume
    cf* S;                // spectrum
    float* out;           // inverse: real output volume
    const float* aux;     // inverse: d (ratio) or est (update)
    const cf* tw;         // twiddles for length M
    const cf* untangle;   // w_X^{brev(p)}, p < M
    const cf* twy;        // w_Y^y, y < Y/2
    const cf* tw3;        // radix-3 twiddles of the row transform (2 Lm entries) when M = 3 Lm
    int ntw;
    ConvDims d;
    float eps;
};

struct ConvPlan {
    ConvDims d;
    cf *tw_x = nullptr, *tw_y = nullptr, *tw_z = nullptr, *untangle = nullptr, *twy = nullptr;
    int ntw_x = 0, ntw_y = 0, ntw_z = 0;
    int Wy = 0, Wz = 0;
    int Lyh = 0, Lz = 0;                      // power-of-two part of Y/2 and Z (== them, or a third of them)
    cf *tw3_y = nullptr, *tw3_z = nullptr;    // radix-3 twiddles where the axis is 3 * 2^k
    cf* tw3_x = nullptr;                      // same for the rows (M = 3 Lm)
    int xr = BH_FC_XR;                        // rows per X-pass tile: which instantiation of the X passes runs
    // wave-private X passes (fftconv_xw.inc) for rows of 1024 / 2048 voxels: their tables, and the stored column of every
    // bit-reversed position (they keep the spectrum row in their own column order)
    // set by bh_richardson_lucy_apply_rows for the duration of one call: where the LAST update pass leaves the row sums of the
    // estimate it stores (xw::Params::rowsum); rl_rowsums_done says that a pass took it
    double* rl_rowsums = nullptr;
    bool rl_rowsums_done = false;
    bool xw = false;
    bool x3 = false;  // rows of 1536 / 3072 voxels: the radix-3 kernels of fftconv_x3.inc (same role, tables and column map)
    cf* xw_tab = nullptr;
    int* xw_col = nullptr;
    // register-stage column passes (fftconv_colw.inc) for columns of 256 / 512 / 1024 points: their twiddle tables
    cf *colw_y = nullptr, *colw_z = nullptr;
    cf* colz = nullptr;  // radix-8 register-stage Z pass of 512-point columns (fftconv_colz.inc)
    cf* colz3 = nullptr;  // register-stage Z pass of 384-point columns (fftconv_colz3.inc)
};

// The X passes exist for two tile heights: 16 rows (M = X/2 up to 1024) and 8 rows (M up to 1536: a 3072-voxel row, for which
// 16 rows of LDS do not fit).  Same spectrum layout either way — the tile height only groups rows.
namespace xr16 {
#define BH_XP_XR BH_FC_XR
#define BH_XP_XNT BH_FC_XNT
#include "fftconv_xpass.inc"
#undef BH_XP_XR
#undef BH_XP_XNT
}  // namespace xr16
namespace xr8 {
#define BH_XP_XR 8
#define BH_XP_XNT BH_FC_XNT8
#include "fftconv_xpass.inc"
#undef BH_XP_XR
#undef BH_XP_XNT
}  // namespace xr8

#include "fftconv_xw.inc"
#include "fftconv_x3.inc"
#include "fftconv_colw.inc"
#include "fftconv_colz.inc"
#include "fftconv_colz3.inc"

// ================================================================================================
// host side
// ================================================================================================

static int ilog2(long v) {
    int l = 0;
    while ((1l << l) < v) ++l;
    return l;
}

// rows per X-pass tile for a row length: the configured height while its LDS tile fits, 8 rows beyond M = 1024
static int x_tile_rows(int64_t X) { return X / 2 > 1024 ? 8 : BH_FC_XR; }

// Shapes the engine runs: X a power of two; Y and Z powers of two or — `radix3` — three times one (the column passes then
// start with a radix-3 step).  Callers whose spectral arithmetic knows the scrambled coefficient order (Tikhonov's filter
// staging) ask without `radix3`; order-agnostic ones (Richardson-Lucy at a padded box) may ask with it.
bool fftconv_supported_ex(int64_t Z, int64_t Y, int64_t X, bool radix3) {
    auto pow2 = [](int64_t v) { return v > 0 && (v & (v - 1)) == 0; };
    auto ok = [&](int64_t v) { return pow2(v) || (radix3 && ((v % 3 == 0 && pow2(v / 3)) || (v % 5 == 0 && pow2(v / 5)))); };
    if (!ok(Z) || !ok(Y) || !ok(X)) return false;
    if (X < 64 || X > 3072) return false;          // M = X/2 in [32, 1536]: (M+1)*(rows+1)*8 + tables <= 160 KiB
    if (!pow2(X) && X / 2 / (X % 3 == 0 ? 3 : 5) < 32) return false;  // rows of 3 * 2^k / 5 * 2^k: parts of at least 32 complex points
    if (X / 4 > (X / 2 > 1024 ? BH_FC_XNT8 : BH_FC_XNT)) return false;  // an X-pass thread owns two complex columns of a row
    if (Y < 2 * 16 || Y / 2 > 2048) return false;   // Y/2 rows x >= 8 columns per tile, whole groups of tile rows
    if (Z < 4 || Z > 2048) return false;
    const int xr = x_tile_rows(X);
    if ((Y % xr) != 0) return false;
    if (!pow2(Z) && Z / (Z % 3 == 0 ? 3 : 5) < 8) return false;        // odd-radix columns: parts of at least 8 rows
    if (!pow2(Y) && Y / 2 / (Y % 3 == 0 ? 3 : 5) < 16) return false;
    const int M = (int)X / 2, Lm = pow2(X) ? M : (X % 3 == 0 ? M / 3 : M / 5);
    const size_t xlds = (size_t)(M + 1) * (xr + 1) * 8 + (size_t)twiddle_count(Lm) * 8 + (size_t)M * 8 + (Lm != M ? (size_t)(M / Lm - 1) * Lm * 8 : 0);
    return xlds <= 160 * 1024;
}

bool fftconv_supported(int64_t Z, int64_t Y, int64_t X) { return fftconv_supported_ex(Z, Y, X, false); }

static std::map<std::tuple<int, int64_t, int64_t, int64_t>, ConvPlan> g_plans;  // twiddle tables per (device, shape): a few KiB, never freed
static std::mutex g_plans_mu;                                                    // contexts of different threads share the cache

static int upload(const std::vector<cf>& h, cf** dptr) {
    BH_CHECK_HIP(hipMalloc(dptr, h.size() * sizeof(cf) + 16));
    BH_CHECK_HIP(hipMemcpy(*dptr, h.data(), h.size() * sizeof(cf), hipMemcpyHostToDevice));
    return BH_OK;
}

// THE predicate for "rows of X voxels (Y of them per plane) run the wave-private X passes": what fftconv_plan enables, and what
// the box chooser (engine_pad_box) and the back-end cost model (rl_plan) of deconv.hip assume when they prefer rows of
// 1536 / 3072 voxels or price a wrap-padded iteration.
// BH_FC_XW=0 keeps the tile-based X passes for every shape (A/B switch, read per call: plans of both kinds can coexist)
// rows of 512 / 1024 / 2048 voxels: fftconv_xw.inc; of 1536 / 3072: fftconv_x3.inc (BH_FC_X3=0 keeps those on the tile kernels)
bool fftconv_rows_wave_private(int64_t Y, int64_t X) {
    const bool x3_rows = (X == 1536 || X == 3072) && !(getenv("BH_FC_X3") && atoi(getenv("BH_FC_X3")) == 0);
    return !(getenv("BH_FC_XW") && atoi(getenv("BH_FC_XW")) == 0) && (X == 512 || X == 1024 || X == 2048 || x3_rows) && ((Y / 2) % 4) == 0;
}

int fftconv_plan(bh_ctx* ctx, int64_t Z, int64_t Y, int64_t X, ConvPlan** out) {
    std::lock_guard<std::mutex> lock(g_plans_mu);
    const bool xw_on = fftconv_rows_wave_private(Y, X);
    auto key = std::make_tuple(ctx->device * 2 + (xw_on ? 1 : 0), Z, Y, X);
    auto it = g_plans.find(key);
    if (it != g_plans.end()) {
        *out = &it->second;
        return BH_OK;
    }
    ConvPlan pl;
    pl.d.Z = (int)Z;
    pl.d.Y = (int)Y;
    pl.d.X = (int)X;
    pl.d.M = (int)X / 2;
    pl.d.XP = (int)X / 2 + 16;
    auto pow2part = [](int64_t n) { return (int)((n & (n - 1)) == 0 ? n : (n % 3 == 0 ? n / 3 : n / 5)); };
    pl.Lyh = pow2part(Y / 2);
    pl.Lz = pow2part(Z);
    pl.xr = x_tile_rows(X);
    pl.d.Lm = pow2part(X / 2);
    pl.d.logM = ilog2(pl.d.Lm);
    pl.d.logYh = ilog2(pl.Lyh);
    pl.d.logZ = ilog2(pl.Lz);
    pl.d.Lyh = pl.Lyh;
    pl.d.Lz = pl.Lz;
    std::vector<cf> h;
    make_twiddles(pl.d.Lm, h);
    pl.ntw_x = (int)h.size();
    BH_TRY(upload(h, &pl.tw_x));
    make_twiddles(pl.Lyh, h);
    pl.ntw_y = (int)h.size();
    BH_TRY(upload(h, &pl.tw_y));
    make_twiddles(pl.Lz, h);
    pl.ntw_z = (int)h.size();
    BH_TRY(upload(h, &pl.tw_z));
    auto radix3_twiddles = [&](int n, int L, cf** dptr) -> int {  // [w_n^(k m), k = 1 .. r - 1], m < L = n / r, r = 3 or 5
        if (L == n) return BH_OK;
        const int r = n / L;
        h.resize((size_t)(r - 1) * L);
        for (int m = 0; m < L; ++m)
            for (int k = 1; k < r; ++k) {
                const double a = -2.0 * M_PI * (double)m * k / (double)n;
                h[(size_t)(r - 1) * m + k - 1] = make_float2((float)std::cos(a), (float)std::sin(a));
            }
        return upload(h, dptr);
    };
    BH_TRY(radix3_twiddles((int)Y / 2, pl.Lyh, &pl.tw3_y));
    BH_TRY(radix3_twiddles((int)Z, pl.Lz, &pl.tw3_z));
    BH_TRY(radix3_twiddles(pl.d.M, pl.d.Lm, &pl.tw3_x));
    h.resize(pl.d.M);
    for (int pp = 0; pp < pl.d.M; ++pp) {
        // frequency stored at position pp: bit-reversed within the (single, or one of three) length-Lm transform(s)
        const int third = pp / pl.d.Lm, r = pp % pl.d.Lm;
        int k = 0;
        for (int b = 0; b < pl.d.logM; ++b)
            if (r & (1 << b)) k |= 1 << (pl.d.logM - 1 - b);
        if (pl.d.Lm != pl.d.M) k = (pl.d.M / pl.d.Lm) * k + third;
        const double a = -2.0 * M_PI * k / (double)X;
        h[pp] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    BH_TRY(upload(h, &pl.untangle));
    h.resize(Y / 2);
    for (int y = 0; y < (int)Y / 2; ++y) {
        const double a = -2.0 * M_PI * y / (double)Y;
        h[y] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    BH_TRY(upload(h, &pl.twy));
    auto tile_w = [&](int64_t N) {
        int w = 1;
        while (2 * w * N <= FC_TILE) w *= 2;  // widest power-of-two tile of N rows in 128 KiB
        if (w > 64) w = 64;       // 512-B row segments are plenty
        if (w > pl.d.XP) w = 16;
        return w < 2 ? 2 : w;
    };
    pl.Wy = tile_w(Y / 2);
    pl.Wz = tile_w(Z);
    auto colw_tables = [&](int64_t n, cf** dptr) -> int {
        if (n == 256) colw::make_tables<8>(h);
        else if (n == 512) colw::make_tables<9>(h);
        else if (n == 1024) colw::make_tables<10>(h);
        else return BH_OK;
        return upload(h, dptr);
    };
    BH_TRY(colw_tables(Y / 2, &pl.colw_y));
    BH_TRY(colw_tables(Z, &pl.colw_z));
    if (Z == colz::N && pl.d.XP >= colz::W) {
        colz::make_tables(h);
        BH_TRY(upload(h, &pl.colz));
    }
    if (Z == 384 && pl.d.XP >= 32) {
        colz3::make_tables<7>(h);
        BH_TRY(upload(h, &pl.colz3));
    } else if (Z == 768 && pl.d.XP >= 16) {
        colz3::make_tables<8>(h);
        BH_TRY(upload(h, &pl.colz3));
    }
    if (xw_on) {
        std::vector<int> col;
        if (X == 3072) x3::make_tables<9>(h, col);
        else if (X == 1536) x3::make_tables<8>(h, col);
        else if (X == 2048) xw::make_tables<10>(h, col);
        else if (X == 1024) xw::make_tables<9>(h, col);
        else xw::make_tables<8>(h, col);
        pl.x3 = X == 3072 || X == 1536;
        BH_TRY(upload(h, &pl.xw_tab));
        BH_CHECK_HIP(hipMalloc(&pl.xw_col, col.size() * sizeof(int)));
        BH_CHECK_HIP(hipMemcpy(pl.xw_col, col.data(), col.size() * sizeof(int), hipMemcpyHostToDevice));
        pl.xw = true;
    }
    auto ins = g_plans.emplace(key, pl);
    *out = &ins.first->second;
    return BH_OK;
}

int fftconv_plan_tag(const ConvPlan& pl) { return pl.xw ? 1 : 0; }
// the last update pass of the next fftconv_richardson_lucy leaves the float64 row sums of its result at `dst` (wave-private X
// passes of 512 / 1024 / 2048-voxel rows only); fftconv_rowsums_taken says whether a pass did, and disarms the plan
void fftconv_arm_rowsums(ConvPlan& pl, double* dst) {
    pl.rl_rowsums = dst;
    pl.rl_rowsums_done = false;
}
bool fftconv_rowsums_taken(ConvPlan& pl) {
    const bool done = pl.rl_rowsums_done;
    pl.rl_rowsums = nullptr;
    pl.rl_rowsums_done = false;
    return done;
}

size_t fftconv_spectrum_elems(const ConvPlan& pl) {
    return (size_t)pl.d.Z * pl.d.Y * pl.d.XP + 64;  // slack: a ragged last column tile reads past its row
}

template <int LOGN>
static int launch_colw(bh_ctx* ctx, ColParams p, int mode) {
    using G = colw::Geo<LOGN>;
    p.W = G::W;
    p.ncoltiles = (int)ceil_div(p.XP, p.W);
    const long ntiles = (long)p.nouter * p.ncoltiles;
    const int grid = (int)std::min<long>(ntiles, (long)ctx->num_cus * (512 / colw::NT));
    auto run = [&](auto kern) -> int {
        BH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)G::LDS_BYTES));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(colw::NT), G::LDS_BYTES, ctx->stream, p);
        BH_CHECK_HIP(hipGetLastError());
        return BH_OK;
    };
    switch (mode) {
        case COL_FWD: return run(colw::colw_kernel<LOGN, COL_FWD>);
        case COL_INV: return run(colw::colw_kernel<LOGN, COL_INV>);
        case COL_FWD_SCALE: return run(colw::colw_kernel<LOGN, COL_FWD_SCALE>);
        case COL_CONV: return run(colw::colw_kernel<LOGN, COL_CONV>);
        case COL_FILTER: return run(colw::colw_kernel<LOGN, COL_FILTER>);
        case COL_CONV16: return run(colw::colw_kernel<LOGN, COL_CONV16>);
        default: return run(colw::colw_kernel<LOGN, COL_CORR>);
    }
}

static int launch_col(bh_ctx* ctx, const ConvPlan& pl, int mode, bool zaxis, cf* S, const cf* otf, float scale, int pcc_norm = 0,
                      int pcc_swap = 0, cf* otf_out = nullptr) {
    ColParams p;
    p.pcc_norm = pcc_norm;
    p.pcc_swap = pcc_swap;
    p.otf_out = otf_out;
    p.S = S;
    p.otf = otf;
    p.XP = pl.d.XP;
    p.scale = scale;
    if (!zaxis) {
        p.N = pl.d.Y / 2;
        p.logN = pl.d.logYh;
        p.W = pl.Wy;
        p.tw = pl.tw_y;
        p.ntw = pl.ntw_y;
        p.L = pl.Lyh;
        p.tw3 = pl.tw3_y;
        p.row_stride = pl.d.XP;
        p.outer_stride = (long)pl.d.Y * pl.d.XP;
        p.sub_stride = (long)(pl.d.Y / 2) * pl.d.XP;
        p.nsub = 2;
        p.nouter = pl.d.Z * 2;
    } else {
        p.N = pl.d.Z;
        p.logN = pl.d.logZ;
        p.W = pl.Wz;
        p.tw = pl.tw_z;
        p.ntw = pl.ntw_z;
        p.L = pl.Lz;
        p.tw3 = pl.tw3_z;
        p.row_stride = (long)pl.d.Y * pl.d.XP;
        p.outer_stride = pl.d.XP;
        p.sub_stride = 0;
        p.nsub = 1;
        p.nouter = pl.d.Y;
    }
    // 512-point Z passes with a spectral product: radix-8 register stages (BH_FC_COLZ=0 keeps the radix-4 LDS steps: A/B switch)
    if (zaxis && pl.colz && p.N == colz::N && (mode == COL_CONV || mode == COL_CORR || mode == COL_FILTER || mode == COL_CONV16 || mode == COL_PCC) &&
        !(getenv("BH_FC_COLZ") && atoi(getenv("BH_FC_COLZ")) == 0) &&
        p.row_stride * 8 * 64 < (1ll << 32)) {  // colz_kernel's lanes address their rows by 32-bit offsets from scalar row pointers
        p.W = colz::W;
        p.tw = pl.colz;
        p.ncoltiles = (int)ceil_div(p.XP, p.W);
        const long ntiles = (long)p.nouter * p.ncoltiles;
        const int grid = (int)std::min<long>(ntiles, (long)ctx->num_cus * (1024 / colz::NT));
        auto run = [&](auto kern) -> int {
            BH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)colz::LDS_BYTES));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(colz::NT), colz::LDS_BYTES, ctx->stream, p);
            BH_CHECK_HIP(hipGetLastError());
            return BH_OK;
        };
        switch (mode) {
            case COL_CONV: return run(colz::colz_kernel<COL_CONV>);
            case COL_CORR: return run(colz::colz_kernel<COL_CORR>);
            case COL_FILTER: return run(colz::colz_kernel<COL_FILTER>);
            case COL_CONV16: return run(colz::colz_kernel<COL_CONV16>);
            default: return run(colz::colz_kernel<COL_PCC>);
        }
    }
    // 384- / 768-point Z passes with a spectral product (the boxes of the deskewed config-4 / config-2 volumes): register stages
    // (BH_FC_COLZ3=0: A/B switch)
    // (lane offsets are 32-bit: 64 rows of the spectrum must span less than 4 GiB, as for colz_kernel)
    if (zaxis && pl.colz3 && (p.N == 384 || p.N == 768) && (mode == COL_CONV || mode == COL_CORR || mode == COL_FILTER) &&
        p.row_stride * 8 * 64 < (1ll << 32) && !(getenv("BH_FC_COLZ3") && atoi(getenv("BH_FC_COLZ3")) == 0)) {
        p.tw = pl.colz3;
        auto run = [&](auto kern, int w, int nt, int lds) -> int {
            p.W = w;
            p.ncoltiles = (int)ceil_div(p.XP, p.W);
            const long ntiles = (long)p.nouter * p.ncoltiles;
            const int grid = (int)std::min<long>(ntiles, ctx->num_cus);
            BH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(nt), lds, ctx->stream, p);
            BH_CHECK_HIP(hipGetLastError());
            return BH_OK;
        };
#define BH_COLZ3(LOGL_)                                                                                                       \
    switch (mode) {                                                                                                           \
        case COL_CONV: return run(colz3::colz3_kernel<LOGL_, COL_CONV>, colz3::Geo<LOGL_>::W, colz3::Geo<LOGL_>::NT, colz3::Geo<LOGL_>::LDS_BYTES); \
        case COL_CORR: return run(colz3::colz3_kernel<LOGL_, COL_CORR>, colz3::Geo<LOGL_>::W, colz3::Geo<LOGL_>::NT, colz3::Geo<LOGL_>::LDS_BYTES); \
        default: return run(colz3::colz3_kernel<LOGL_, COL_FILTER>, colz3::Geo<LOGL_>::W, colz3::Geo<LOGL_>::NT, colz3::Geo<LOGL_>::LDS_BYTES);     \
    }
        if (p.N == 384) { BH_COLZ3(7) } else { BH_COLZ3(8) }
#undef BH_COLZ3
    }
    // columns of 256 / 512 / 1024 points: the register-stage kernels (BH_FC_COLW=0 keeps the LDS-stepped ones: A/B switch)
    const cf* colw_tab = zaxis ? pl.colw_z : pl.colw_y;
    // BH_FC_COLW: 0 never, 1 always, 2 the Y passes only, 3 the Z passes only; default (4): the Y passes, and the Z pass for
    // columns of 256 points (one exchange per transform).  Measured (tools/ab_env.sh): the Z pass of 512-point columns pays
    // more for its 8 barriers per tile at 8 wavefronts than it saves in LDS round trips (6.97 against 6.52 ms at config 2;
    // 5.66 ms with the barriers compiled out), the others win.
    const int colw_mode = getenv("BH_FC_COLW") ? atoi(getenv("BH_FC_COLW")) : 4;
    const bool colw_axis = colw_mode == 1 || (colw_mode == 2 && !zaxis) || (colw_mode == 3 && zaxis) ||
                           (colw_mode == 4 && (!zaxis || p.N == 256));
    if (colw_tab && (long)p.N * p.row_stride < (1l << 31) && colw_axis && mode != COL_PCC) {
        p.tw = colw_tab;
        return p.N == 1024 ? launch_colw<10>(ctx, p, mode) : (p.N == 512 ? launch_colw<9>(ctx, p, mode) : launch_colw<8>(ctx, p, mode));
    }
    p.logW = ilog2(p.W);
    p.midfuse = getenv("BH_FC_NOZMID") == nullptr;
    p.ncoltiles = (int)ceil_div(pl.d.XP, p.W);
    BH_REQUIRE((long)p.N * p.W <= FC_TILE && (long)p.N * (p.W / 2) <= 16l * FC_NT && (FC_NT % (p.W / 2)) == 0,
               "internal: column tile %dx%d unsupported", p.N, p.W);
    const size_t lds = (size_t)p.N * p.W * 8 + (size_t)p.ntw * 8 + (p.L != p.N ? (size_t)(p.N / p.L - 1) * p.L * 8 : 0);
    const long ntiles = (long)p.nouter * p.ncoltiles;
    const int grid = (int)std::min<long>(ntiles, ctx->num_cus);
    auto run = [&](auto kern) -> int {
        BH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(FC_NT), lds, ctx->stream, p);
        BH_CHECK_HIP(hipGetLastError());
        return BH_OK;
    };
    const long per_round = (long)(FC_NT / (p.W / 2));
    const int rounds = (int)ceil_div(p.N, per_round);
#define BH_COL_DISPATCH_(R, RDX)                                           \
    switch (mode) {                                                        \
        case COL_FWD: return run(col_pass_kernel<COL_FWD, R, RDX>);        \
        case COL_INV: return run(col_pass_kernel<COL_INV, R, RDX>);        \
        case COL_FWD_SCALE: return run(col_pass_kernel<COL_FWD_SCALE, R, RDX>); \
        case COL_CONV: return run(col_pass_kernel<COL_CONV, R, RDX>);      \
        case COL_FILTER: return run(col_pass_kernel<COL_FILTER, R, RDX>);  \
        case COL_CONV16: return run(col_pass_kernel<COL_CONV16, R, RDX>);  \
        case COL_PCC: return run(col_pass_kernel<COL_PCC, R, RDX>);        \
        default: return run(col_pass_kernel<COL_CORR, R, RDX>);            \
    }
#define BH_COL_DISPATCH(R)                                 \
    if (p.N / p.L == 3) { BH_COL_DISPATCH_(R, 3) }         \
    else if (p.N / p.L == 5) { BH_COL_DISPATCH_(R, 5) }    \
    else { BH_COL_DISPATCH_(R, 1) }
    if (rounds <= 1) { BH_COL_DISPATCH(1) }
    if (rounds <= 2) { BH_COL_DISPATCH(2) }
    if (rounds <= 4) { BH_COL_DISPATCH(4) }
    if (rounds <= 8) { BH_COL_DISPATCH(8) }
    BH_COL_DISPATCH(16)
#undef BH_COL_DISPATCH
#undef BH_COL_DISPATCH_
}

template <int LOGM>
static int launch_xw_m(bh_ctx* ctx, const xw::Params& p, int mode, int* grid_out = nullptr) {
    using G = xw::Geo<LOGM>;
    const long npairs = (long)p.Z * (p.Y / 2);
    const int grid = (int)std::min<long>(ceil_div(npairs, (long)xw::NW * G::PAIRS), ctx->num_cus);
    if (grid_out) *grid_out = grid;
    auto run = [&](auto kern) -> int {
        BH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)G::LDS_BYTES));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(xw::NT), G::LDS_BYTES, ctx->stream, p);
        BH_CHECK_HIP(hipGetLastError());
        return BH_OK;
    };
    switch (mode) {
        case xw::FWD: return run(xw::xw_kernel<LOGM, xw::FWD>);
        case xw::INV_STORE: return run(xw::xw_kernel<LOGM, xw::INV_STORE>);
        case xw::INV_RATIO: return run(xw::xw_kernel<LOGM, xw::INV_RATIO>);
        case xw::INV_UPDATE: return run(xw::xw_kernel<LOGM, xw::INV_UPDATE>);
        case xw::FUSED_RATIO: return run(xw::xw_kernel<LOGM, xw::FUSED_RATIO>);
        case xw::FUSED_RATIO_WRAP: return run(xw::xw_kernel<LOGM, xw::FUSED_RATIO_WRAP>);
        case xw::FUSED_UPDATE_WRAP: return run(xw::xw_kernel<LOGM, xw::FUSED_UPDATE_WRAP>);
        case xw::INV_UPDATE_CROP: return run(xw::xw_kernel<LOGM, xw::INV_UPDATE_CROP>);
        case xw::INV_ARGMAX: return run(xw::xw_kernel<LOGM, xw::INV_ARGMAX>);
        default: return run(xw::xw_kernel<LOGM, xw::FUSED_UPDATE>);
    }
}

template <int LOGL>
static int launch_x3_m(bh_ctx* ctx, const xw::Params& p, int mode) {
    using G = x3::Geo<LOGL>;
    const long npairs = (long)p.Z * (p.Y / 2);
    const int grid = (int)std::min<long>(ceil_div(npairs, (long)x3::NW * G::PAIRS), ctx->num_cus);
    auto run = [&](auto kern) -> int {
        BH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)G::LDS_BYTES));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(x3::NT), G::LDS_BYTES, ctx->stream, p);
        BH_CHECK_HIP(hipGetLastError());
        return BH_OK;
    };
    switch (mode) {
        case xw::FWD: return run(x3::x3_kernel<LOGL, xw::FWD>);
        case xw::INV_STORE: return run(x3::x3_kernel<LOGL, xw::INV_STORE>);
        case xw::INV_RATIO: return run(x3::x3_kernel<LOGL, xw::INV_RATIO>);
        case xw::INV_UPDATE: return run(x3::x3_kernel<LOGL, xw::INV_UPDATE>);
        case xw::FUSED_RATIO: return run(x3::x3_kernel<LOGL, xw::FUSED_RATIO>);
        case xw::FUSED_RATIO_WRAP: return run(x3::x3_kernel<LOGL, xw::FUSED_RATIO_WRAP>);
        case xw::FUSED_UPDATE_WRAP: return run(x3::x3_kernel<LOGL, xw::FUSED_UPDATE_WRAP>);
        case xw::INV_UPDATE_CROP: return run(x3::x3_kernel<LOGL, xw::INV_UPDATE_CROP>);
        default: return run(x3::x3_kernel<LOGL, xw::FUSED_UPDATE>);
    }
}

static int launch_xw(bh_ctx* ctx, const ConvPlan& pl, bool inverse, int epi, const float* in, cf* S, float* out,
                     const float* aux, float eps, bool fuse_fwd, const double* norm_mean = nullptr) {
    xw::Params p;
    p.norm_mean = norm_mean;
    p.rowsum = nullptr;
    p.S_out = nullptr;
    p.wz = p.wx = xw::Params::Wrap{0, 0, 0, 0};
    p.in = in;
    p.S = S;
    p.out = out;
    p.aux = aux;
    p.tab = pl.xw_tab;
    p.twy = pl.twy;
    p.Z = pl.d.Z;
    p.Y = pl.d.Y;
    p.XP = pl.d.XP;
    p.eps = eps;
    const int mode = !inverse ? xw::FWD
                     : epi == XE_STORE ? xw::INV_STORE
                     : epi == XE_RATIO ? (fuse_fwd ? xw::FUSED_RATIO : xw::INV_RATIO)
                                       : (fuse_fwd ? xw::FUSED_UPDATE : xw::INV_UPDATE);
    if (mode == xw::INV_UPDATE && !pl.x3 && pl.rl_rowsums != nullptr) {
        p.rowsum = pl.rl_rowsums;
        const_cast<ConvPlan&>(pl).rl_rowsums_done = true;
    }
    if (pl.x3) return pl.d.M == 1536 ? launch_x3_m<9>(ctx, p, mode) : launch_x3_m<8>(ctx, p, mode);
    return pl.d.M == 1024 ? launch_xw_m<10>(ctx, p, mode) : (pl.d.M == 512 ? launch_xw_m<9>(ctx, p, mode) : launch_xw_m<8>(ctx, p, mode));
}

// inverse X pass that keeps only the first occurrence of max |.| (xw::INV_ARGMAX): `partial` receives *npartial entries
static int launch_xw_argmax(bh_ctx* ctx, const ConvPlan& pl, cf* S, ArgMax* partial, int* npartial) {
    xw::Params p;
    p.norm_mean = nullptr;
    p.rowsum = nullptr;
    p.S_out = nullptr;
    p.wz = p.wx = xw::Params::Wrap{0, 0, 0, 0};
    p.in = nullptr;
    p.S = S;
    p.out = reinterpret_cast<float*>(partial);
    p.aux = nullptr;
    p.tab = pl.xw_tab;
    p.twy = pl.twy;
    p.Z = pl.d.Z;
    p.Y = pl.d.Y;
    p.XP = pl.d.XP;
    p.eps = 0.f;
    int grid = 0;
    BH_TRY(pl.d.M == 1024 ? launch_xw_m<10>(ctx, p, xw::INV_ARGMAX, &grid)
                          : (pl.d.M == 512 ? launch_xw_m<9>(ctx, p, xw::INV_ARGMAX, &grid) : launch_xw_m<8>(ctx, p, xw::INV_ARGMAX, &grid)));
    *npartial = grid * xw::NW;
    return BH_OK;
}

static int launch_x(bh_ctx* ctx, const ConvPlan& pl, bool inverse, int epi, const float* in, cf* S, float* out,
                    const float* aux, float eps, bool fuse_fwd = false) {
    if (pl.xw) return launch_xw(ctx, pl, inverse, epi, in, S, out, aux, eps, fuse_fwd);
    return pl.xr == 8 ? xr8::launch_x(ctx, pl, inverse, epi, in, S, out, aux, eps, fuse_fwd)
                      : xr16::launch_x(ctx, pl, inverse, epi, in, S, out, aux, eps, fuse_fwd);
}

// OTF (scrambled order, scaled by 2/V so that forward -> multiply -> inverse is a normalised convolution)
int fftconv_make_otf(bh_ctx* ctx, const ConvPlan& pl, const float* padded_psf, cf* otf) {
    const double V = (double)pl.d.Z * pl.d.Y * pl.d.X;
    BH_TRY(launch_x(ctx, pl, false, 0, padded_psf, otf, nullptr, nullptr, 0.f));
    BH_TRY(launch_col(ctx, pl, COL_FWD, false, otf, nullptr, 1.f));
    BH_TRY(launch_col(ctx, pl, COL_FWD_SCALE, true, otf, nullptr, (float)(2.0 / V)));
    return BH_OK;
}

// out = epilogue( irfft( rfft(in) * OTF or conj(OTF) ) )
int fftconv_apply(bh_ctx* ctx, const ConvPlan& pl, const float* in, const cf* otf, bool correlate, cf* spec,
                  int epilogue, const float* aux, float eps, float* out) {
    BH_TRY(launch_x(ctx, pl, false, 0, in, spec, nullptr, nullptr, 0.f));
    BH_TRY(launch_col(ctx, pl, COL_FWD, false, spec, nullptr, 1.f));
    BH_TRY(launch_col(ctx, pl, correlate ? COL_CORR : COL_CONV, true, spec, otf, 1.f));
    BH_TRY(launch_col(ctx, pl, COL_INV, false, spec, nullptr, 1.f));
    BH_TRY(launch_x(ctx, pl, true, epilogue, nullptr, spec, out, aux, eps));
    return BH_OK;
}


// Tikhonov filter H/(H^2 + reg) * 2/V from the reference's natural-order full-spectrum H, written in the
// engine's scrambled half-spectrum layout.  One workgroup per spectrum row: coalesced read of tf[kz][ky][0..M],
// bit-reversal permutation through LDS, coalesced write of filt[zs][ys][0..XP).
__global__ __launch_bounds__(256) void tikhonov_filter_rows_kernel(const float* __restrict__ tf, float* __restrict__ filt,
                                                                   ConvDims d, float reg, float scale,
                                                                   const int* __restrict__ xcol) {
    extern __shared__ __attribute__((aligned(16))) float rowbuf[];  // [XP]
    constexpr int PT = 8;  // columns per thread: XP <= 2048 (X <= 3072: XP = 1552)
    const int Yh = d.Y / 2;
    // a thread's columns kx = tid + 256 j and where they go in the stored row do not depend on the row
    int ps[PT];
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        const int kx = threadIdx.x + 256 * j;
        int q = kx;  // Nyquist and pad columns keep their place
        if (kx < d.M) {
            const int rx = d.M / d.Lm, jx = kx / rx, tx = kx - rx * jx;
            q = tx * d.Lm + (int)(__brev((unsigned)jx) >> (32 - d.logM));
            if (xcol) q = xcol[q];  // the wave-private X passes store position q in column xcol[q]
        }
        ps[j] = kx < d.XP ? q : -1;
    }
    auto source_row = [&](long row) -> const float* {
        const int zs = (int)(row / d.Y), ys = (int)(row - (long)zs * d.Y);
        // stored position -> frequency: a 3 * 2^k column holds X[3 j + t] in third t at the bit-reversed j
        const int tz = zs / d.Lz, rz = zs - tz * d.Lz;
        const int jz = (int)(__brev((unsigned)rz) >> (32 - d.logZ));
        const int kz = (d.Z / d.Lz) * jz + tz;  // Z / Lz = 1, 3 or 5: frequency r j + t sits in part t at the bit-reversed j
        const int half = ys / Yh, r = ys - half * Yh;
        const int ty = r / d.Lyh, ry = r - ty * d.Lyh;
        const int jy = (int)(__brev((unsigned)ry) >> (32 - d.logYh));
        const int ky = 2 * ((Yh / d.Lyh) * jy + ty) + half;
        return tf + ((long)kz * d.Y + ky) * d.X;
    };
    const long nrows = (long)d.Z * d.Y;
    float h[PT];
    auto load_row = [&](long row) {  // unconditional, clamped: the values of columns past M are not used
        const float* src = source_row(row < nrows ? row : nrows - 1);
#pragma unroll
        for (int j = 0; j < PT; ++j) h[j] = src[min((int)threadIdx.x + 256 * j, d.M)];
    };
    long row = blockIdx.x;
    if (row < nrows) load_row(row);
    for (; row < nrows; row += gridDim.x) {
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            const int kx = threadIdx.x + 256 * j;
            if (ps[j] >= 0) rowbuf[ps[j]] = kx <= d.M ? (h[j] / (h[j] * h[j] + reg)) * scale : 0.0f;
        }
        load_row(row + gridDim.x);  // the next row travels behind this row's permutation and store
        __syncthreads();
        float4* dst = reinterpret_cast<float4*>(filt + row * d.XP);  // XP % 4 == 0, rows 16-B aligned
        for (int q = threadIdx.x; q < d.XP / 4; q += 256) dst[q] = reinterpret_cast<const float4*>(rowbuf)[q];
        __syncthreads();
    }
}

// Inverse filter of a general transfer function H (natural order, full spectrum (Z, Y, X), complex64 or float32), staged in
// the engine's scrambled half-spectrum layout:  F = conj(H) / (|H|^2 + reg) * scale.  The real part of ifftn(fftn(x) F) is
// what the reference keeps (waveorder: `.real` of the filtered inverse transform), i.e. only the Hermitian part
// F_h(k) = (F(k) + conj(F(-k))) / 2 acts on a real volume — that is what a half-spectrum product can and does apply.
// One workgroup per stored spectrum row; `bf16`: the staged value is rounded to bfloat16 pairs (round to nearest even).
__device__ __forceinline__ unsigned int f32_to_bf16_bits(float f) {
    const unsigned int u = __float_as_uint(f);
    if ((u & 0x7f800000u) == 0x7f800000u) return u >> 16;  // inf / nan: truncate (a quiet-nan payload bit survives)
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
template <bool CPLX>
__global__ __launch_bounds__(256) void inverse_filter_rows_kernel(const void* __restrict__ tf, void* __restrict__ filt,
                                                                  ConvDims d, float reg, float scale, int bf16,
                                                                  const int* __restrict__ xcol) {
    extern __shared__ cf rowc[];  // [XP]
    const int Yh = d.Y / 2;
    for (long row = blockIdx.x; row < (long)d.Z * d.Y; row += gridDim.x) {
        const int zs = (int)(row / d.Y), ys = (int)(row - (long)zs * d.Y);
        const int tz = zs / d.Lz, rz = zs - tz * d.Lz;
        const int kz = (d.Z / d.Lz) * (int)(__brev((unsigned)rz) >> (32 - d.logZ)) + tz;
        const int half = ys / Yh, r = ys - half * Yh;
        const int ty = r / d.Lyh, ry = r - ty * d.Lyh;
        const int ky = 2 * ((Yh / d.Lyh) * (int)(__brev((unsigned)ry) >> (32 - d.logYh)) + ty) + half;
        const int mz = kz ? d.Z - kz : 0, my = ky ? d.Y - ky : 0;  // -k
        const long src = ((long)kz * d.Y + ky) * d.X, msrc = ((long)mz * d.Y + my) * d.X;
        for (int kx = threadIdx.x; kx < d.XP; kx += 256) {
            cf f = make_float2(0.f, 0.f);
            if (kx <= d.M) {
                const int mx = kx ? d.X - kx : 0;
                cf h, hm;
                if (CPLX) {
                    h = reinterpret_cast<const cf*>(tf)[src + kx];
                    hm = reinterpret_cast<const cf*>(tf)[msrc + mx];
                } else {
                    h = make_float2(reinterpret_cast<const float*>(tf)[src + kx], 0.f);
                    hm = make_float2(reinterpret_cast<const float*>(tf)[msrc + mx], 0.f);
                }
                const float q = 1.0f / (h.x * h.x + h.y * h.y + reg), qm = 1.0f / (hm.x * hm.x + hm.y * hm.y + reg);
                // F(k) = conj(h) q ; conj(F(-k)) = hm qm
                f = make_float2(0.5f * (h.x * q + hm.x * qm) * scale, 0.5f * (-h.y * q + hm.y * qm) * scale);
            }
            int ps = kx;  // Nyquist and pad columns keep their place
            if (kx < d.M) {
                const int rx = d.M / d.Lm, jx = kx / rx, tx = kx - rx * jx;
                ps = tx * d.Lm + (int)(__brev((unsigned)jx) >> (32 - d.logM));
                if (xcol) ps = xcol[ps];
            }
            rowc[ps] = f;
        }
        __syncthreads();
        for (int ps = threadIdx.x; ps < d.XP; ps += 256) {
            const cf f = rowc[ps];
            if (bf16)
                reinterpret_cast<unsigned int*>(filt)[row * d.XP + ps] = f32_to_bf16_bits(f.x) | (f32_to_bf16_bits(f.y) << 16);
            else
                reinterpret_cast<cf*>(filt)[row * d.XP + ps] = f;
        }
        __syncthreads();
    }
}

// filt: NS complex (f32) or NS 32-bit words (bf16 pairs), NS = fftconv_spectrum_elems
int fftconv_stage_inverse_filter(bh_ctx* ctx, const ConvPlan& pl, const void* tf, bool tf_complex, float reg, bool bf16,
                                 void* filt) {
    const double V = (double)pl.d.Z * pl.d.Y * pl.d.X;
    const int grid = ctx->num_cus * 8;
    const int* xcol = pl.xw ? pl.xw_col : nullptr;
    if (tf_complex)
        hipLaunchKernelGGL(inverse_filter_rows_kernel<true>, dim3(grid), dim3(256), pl.d.XP * sizeof(cf), ctx->stream, tf, filt,
                           pl.d, reg, (float)(2.0 / V), bf16 ? 1 : 0, xcol);
    else
        hipLaunchKernelGGL(inverse_filter_rows_kernel<false>, dim3(grid), dim3(256), pl.d.XP * sizeof(cf), ctx->stream, tf, filt,
                           pl.d, reg, (float)(2.0 / V), bf16 ? 1 : 0, xcol);
    BH_CHECK_HIP(hipGetLastError());
    return BH_OK;
}

// whether fftconv_apply_staged_filter can take the x / mean - 1 normalisation into its first pass
bool fftconv_fuses_normalisation(const ConvPlan& pl) { return pl.xw; }

// out = irfft( rfft(in) * staged filter ): 5 passes, the product rides in the Z pass.  norm_mean (device, may be null; only
// when fftconv_fuses_normalisation): the forward X pass transforms in / *norm_mean - 1.
int fftconv_apply_staged_filter(bh_ctx* ctx, const ConvPlan& pl, const float* in, const void* filt, bool bf16, cf* spec,
                                float* out, const double* norm_mean) {
    if (norm_mean) {
        BH_REQUIRE(pl.xw, "internal: fused normalisation needs the wave-private X passes");
        BH_TRY(launch_xw(ctx, pl, false, 0, in, spec, nullptr, nullptr, 0.f, false, norm_mean));
    } else
    BH_TRY(launch_x(ctx, pl, false, 0, in, spec, nullptr, nullptr, 0.f));
    BH_TRY(launch_col(ctx, pl, COL_FWD, false, spec, nullptr, 1.f));
    BH_TRY(launch_col(ctx, pl, bf16 ? COL_CONV16 : COL_CONV, true, spec, reinterpret_cast<const cf*>(filt), 1.f));
    BH_TRY(launch_col(ctx, pl, COL_INV, false, spec, nullptr, 1.f));
    BH_TRY(launch_x(ctx, pl, true, XE_STORE, nullptr, spec, out, nullptr, 0.f));
    return BH_OK;
}

// z-slab variants: the X passes and the Y column pass work plane by plane, so a range of planes is the same launch on
// offset pointers with Z shrunk (the Z pass needs every plane and is never slabbed)
static int launch_x_slab(bh_ctx* ctx, const ConvPlan& pl, bool inverse, int epi, const float* in, cf* S, float* out,
                         const float* aux, float eps, bool fuse_fwd, int z0, int nz) {
    ConvPlan q = pl;
    q.d.Z = nz;
    const long rows = (long)z0 * pl.d.Y;
    return launch_x(ctx, q, inverse, epi, in ? in + rows * pl.d.X : nullptr, S + rows * pl.d.XP, out ? out + rows * pl.d.X : nullptr,
                    aux ? aux + rows * pl.d.X : nullptr, eps, fuse_fwd);
}
static int launch_col_y_slab(bh_ctx* ctx, const ConvPlan& pl, int mode, cf* S, int z0, int nz) {
    ConvPlan q = pl;
    q.d.Z = nz;
    return launch_col(ctx, q, mode, false, S + (long)z0 * pl.d.Y * pl.d.XP, nullptr, 1.f);
}

// Planes per slab for the MALL hand-off below (0 = off): BH_FC_SLAB_MB megabytes of spectrum, default off.
static int slab_planes(const ConvPlan& pl) {
    static const int mb = getenv("BH_FC_SLAB_MB") ? atoi(getenv("BH_FC_SLAB_MB")) : 0;
    if (mb <= 0) return 0;
    const double plane = (double)pl.d.Y * pl.d.XP * sizeof(cf);
    const int n = (int)((double)mb * 1048576.0 / plane);
    return n >= 1 && n < pl.d.Z ? n : 0;
}

// Richardson-Lucy iterations with the X passes of consecutive convolutions fused:
//   S = Xfwd(est);  repeat { Y, Z*OTF, Yinv ; [Xinv -> d/max(.,eps) -> Xfwd] ; Y, Z*conj(OTF), Yinv ;
//                            [Xinv -> est = max(est*.,0) (stored) -> Xfwd] }   (last iteration: no trailing Xfwd)
// 8 passes and 84 B/voxel per iteration instead of 10 passes and 96 B/voxel.
// `est` is output only: the first pass fills it with max(d, 0).
// With BH_FC_SLAB_MB set, the Yinv -> X -> Yfwd chain between two Z passes runs slab by slab (a few z planes at a time), so
// that each kernel finds the planes its predecessor just wrote in the 256-MB memory-side cache instead of HBM.
int fftconv_richardson_lucy(bh_ctx* ctx, const ConvPlan& pl, const float* d, const cf* otf, bool otf_real, cf* spec,
                            int iterations, float eps, float* est) {
    if (iterations <= 0) return BH_OK;
    // otf_real: `otf` holds one float per bin (the transfer function of a point-symmetric PSF); convolution and correlation
    // are then the same real product
    const int CONV = otf_real ? COL_FILTER : COL_CONV, CORR = otf_real ? COL_FILTER : COL_CORR;
    const int slab = slab_planes(pl);
    if (slab > 0) {
        const int Z = pl.d.Z;
        for (int z0 = 0; z0 < Z; z0 += slab) {
            const int nz = std::min(slab, Z - z0);
            BH_TRY(launch_x_slab(ctx, pl, false, 0, d, spec, est, nullptr, 0.f, false, z0, nz));
            BH_TRY(launch_col_y_slab(ctx, pl, COL_FWD, spec, z0, nz));
        }
        for (int it = 0; it < iterations; ++it) {
            const bool last = it + 1 == iterations;
            BH_TRY(launch_col(ctx, pl, CONV, true, spec, otf, 1.f));
            for (int z0 = 0; z0 < Z; z0 += slab) {
                const int nz = std::min(slab, Z - z0);
                BH_TRY(launch_col_y_slab(ctx, pl, COL_INV, spec, z0, nz));
                BH_TRY(launch_x_slab(ctx, pl, true, XE_RATIO, nullptr, spec, nullptr, d, eps, true, z0, nz));
                BH_TRY(launch_col_y_slab(ctx, pl, COL_FWD, spec, z0, nz));
            }
            BH_TRY(launch_col(ctx, pl, CORR, true, spec, otf, 1.f));
            for (int z0 = 0; z0 < Z; z0 += slab) {
                const int nz = std::min(slab, Z - z0);
                BH_TRY(launch_col_y_slab(ctx, pl, COL_INV, spec, z0, nz));
                BH_TRY(launch_x_slab(ctx, pl, true, XE_UPDATE, nullptr, spec, est, est, eps, !last, z0, nz));
                if (!last) BH_TRY(launch_col_y_slab(ctx, pl, COL_FWD, spec, z0, nz));
            }
        }
        return BH_OK;
    }
    BH_TRY(launch_x(ctx, pl, false, 0, d, spec, est, nullptr, 0.f));  // est = max(d, 0) written by the same pass
    for (int it = 0; it < iterations; ++it) {
        BH_TRY(launch_col(ctx, pl, COL_FWD, false, spec, nullptr, 1.f));
        BH_TRY(launch_col(ctx, pl, CONV, true, spec, otf, 1.f));
        BH_TRY(launch_col(ctx, pl, COL_INV, false, spec, nullptr, 1.f));
        BH_TRY(launch_x(ctx, pl, true, XE_RATIO, nullptr, spec, nullptr, d, eps, true));
        BH_TRY(launch_col(ctx, pl, COL_FWD, false, spec, nullptr, 1.f));
        BH_TRY(launch_col(ctx, pl, CORR, true, spec, otf, 1.f));
        BH_TRY(launch_col(ctx, pl, COL_INV, false, spec, nullptr, 1.f));
        BH_TRY(launch_x(ctx, pl, true, XE_UPDATE, nullptr, spec, est, est, eps, it + 1 < iterations));
    }
    return BH_OK;
}

// Where the driver puts the spectrum matters to ONE kernel: the fused update X pass (S read and written, the estimate read and
// written: four streams) takes 6.0 to 7.1 ms depending on the physical pages behind `spec` — the same virtual address
// re-allocated gives 35.0 to 36.2 ms per iteration at config 2, offsets inside one allocation give the same time to
// +-0.05 ms, every other kernel is indifferent (tools/ctx_probe.py).  So a new spectrum allocation of a large
// volume is auditioned once: the pass is timed on it and on up to four more allocations held at the same time, until a fast and a
// slow one have both been seen, and the fastest stays (30-70 ms and up to 35 GB of transient memory once per context and shape;
// BH_FC_TUNE_ALLOC=0 skips it).
// `est` is any V-float buffer the caller is about to overwrite (the pass reads and writes it), `bytes` the allocation size.
int fftconv_tune_spectrum(bh_ctx* ctx, const ConvPlan& pl, float* est, size_t bytes, cf** spec) {
    const double V = (double)pl.d.Z * pl.d.Y * pl.d.X;
    // Round 3: with the workspace's default layout (2-MiB physical chunks in a shuffled order, context.hip dev_alloc) the pass
    // reads 5.87-6.5 ms on every allocation tried and the audition is off; it stays for the hipMalloc layout
    // (BH_ALLOC_VMM_MB=0), where the two states are 5.95 and 7.07 ms.  BH_FC_TUNE_ALLOC=0 / 1 forces it off / on.
    const bool tune = getenv("BH_FC_TUNE_ALLOC") ? atoi(getenv("BH_FC_TUNE_ALLOC")) != 0 : !(dev_alloc_is_shuffled() && dev_block_is_vmm(*spec));
    if (!pl.xw || V < (double)(1u << 28) || !tune) return BH_OK;
    hipEvent_t e0, e1;
    BH_CHECK_HIP(hipEventCreate(&e0));
    BH_CHECK_HIP(hipEventCreate(&e1));
    auto audition = [&](cf* s, float* ms) -> int {
        for (int rep = 0; rep < 2; ++rep) {  // the second launch is the measurement
            BH_CHECK_HIP(hipEventRecord(e0, ctx->stream));
            BH_TRY(launch_x(ctx, pl, true, XE_UPDATE, nullptr, s, est, est, 1e-6f, true));
            BH_CHECK_HIP(hipEventRecord(e1, ctx->stream));
            BH_CHECK_HIP(hipEventSynchronize(e1));
            BH_CHECK_HIP(hipEventElapsedTime(ms, e0, e1));
        }
        return BH_OK;
    };
    constexpr int NC = 5;  // four of six simultaneous allocations measured slow: five tries leave ~13 % of the draws without a fast one
    cf* cand[NC] = {*spec, nullptr, nullptr, nullptr, nullptr};
    float ms[NC] = {0.f, 0.f, 0.f, 0.f, 0.f};
    int n = 1, best = 0, rc = audition(cand[0], &ms[0]);
    float slowest = ms[0];
    for (; rc == BH_OK && n < NC; ++n) {
        if (ms[best] < 0.93f * slowest) break;  // both levels seen: the fast one is in hand
        if (dev_alloc(ctx->device, bytes, (void**)&cand[n]) != hipSuccess) {
            (void)hipGetLastError();
            cand[n] = nullptr;
            break;
        }
        rc = audition(cand[n], &ms[n]);
        if (rc != BH_OK) {
            ++n;
            break;
        }
        if (ms[n] < ms[best]) best = n;
        if (ms[n] > slowest) slowest = ms[n];
    }
    if (getenv("BH_DEBUG_SCRATCH"))
        fprintf(stderr, "[bh tune] fused update pass on %d spectrum allocation(s): %.3f %.3f %.3f %.3f %.3f ms -> #%d\n", n, ms[0], ms[1],
                ms[2], ms[3], ms[4], best);
    for (int i = 0; i < n; ++i)
        if (i != best && cand[i]) (void)dev_free(cand[i]);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *spec = cand[best];
    return rc;
}

// The transform passes of one Richardson-Lucy iteration on a volume the CALLER keeps padded (deconv.hip:
// richardson_lucy_engine_padded): forward X of est_p, convolution, [inverse X -> d_p / max(., eps) -> forward X] fused,
// correlation, inverse X stored to corr_p.  est_p is wrap-extended so that the convolution is right on the volume's own box;
// d_p is zero outside that box, which zeroes the ratio there, so corr_p is the LINEAR correlation of the zero-padded ratio:
// the caller folds its wrapped-around tails back, multiplies, and rebuilds est_p.  9 transform passes (8 when nothing is
// padded and update -> forward can stay fused).
int fftconv_rl_iteration_padded(bh_ctx* ctx, const ConvPlan& pl, const float* est_p, const float* d_p, const cf* otf,
                                bool otf_real, cf* spec, float eps, float* corr_p) {
    const int COL_CONV = otf_real ? bh::COL_FILTER : bh::COL_CONV, COL_CORR = otf_real ? bh::COL_FILTER : bh::COL_CORR;
    BH_TRY(launch_x(ctx, pl, false, 0, est_p, spec, nullptr, nullptr, 0.f));
    BH_TRY(launch_col(ctx, pl, COL_FWD, false, spec, nullptr, 1.f));
    BH_TRY(launch_col(ctx, pl, COL_CONV, true, spec, otf, 1.f));
    BH_TRY(launch_col(ctx, pl, COL_INV, false, spec, nullptr, 1.f));
    BH_TRY(launch_x(ctx, pl, true, XE_RATIO, nullptr, spec, nullptr, d_p, eps, true));
    BH_TRY(launch_col(ctx, pl, COL_FWD, false, spec, nullptr, 1.f));
    BH_TRY(launch_col(ctx, pl, COL_CORR, true, spec, otf, 1.f));
    BH_TRY(launch_col(ctx, pl, COL_INV, false, spec, nullptr, 1.f));
    BH_TRY(launch_x(ctx, pl, true, XE_STORE, nullptr, spec, corr_p, nullptr, 0.f));
    return BH_OK;
}

// Richardson-Lucy at a wrap-padded box WITHOUT a fold pass (rows the wave-private X passes take: fftconv_xw.inc / fftconv_x3.inc;
// Y unpadded).
// The estimate going into the convolution is wrap-extended by (lo below, hi above) the volume on the padded axes (z, x); the
// ratio going into the correlation is wrap-extended too — by (hi below, lo above), the mirror image, which is what the
// correlation's taps reach — instead of zero-padded, so the correlation is already the circular one on the volume's own box
// and nothing has to be folded back.  Both extensions are made by the fused X pass that produces the values: along x inside
// the row (two strips of at most K - 1 floats through LDS), along z by letting the wavefront of a margin plane re-read the
// spectrum and the epilogue operand of the plane it mirrors and redo that plane's row — no pass over memory of its own.  Because
// a margin plane reads what another wavefront overwrites, the X passes run out of place here (two spectrum buffers, two
// estimate buffers).  One iteration = the 8 passes of the unpadded path instead of 9 transform passes + a fold / rewrap pass.
bool fftconv_rl_wrap_supported(const ConvPlan& pl, const int64_t N[3], const int64_t K[3], const int64_t P[3]) {
    // any wave-private row length, Y unpadded; the two x strips (K - 1 floats) travel through the head of a row's LDS buffer
    return pl.xw && N[1] == P[1] && K[2] <= 256 && getenv("BH_RL_NOWRAP") == nullptr;
}

static int launch_x3_wrap(bh_ctx* ctx, const ConvPlan& pl, int mode, const cf* S_in, cf* S_out, float* out, const float* aux,
                          float eps, xw::Params::Wrap wz, xw::Params::Wrap wx) {
    xw::Params p;
    p.norm_mean = nullptr;
    p.rowsum = nullptr;
    p.in = nullptr;
    p.S = const_cast<cf*>(S_in);
    p.S_out = S_out;
    p.out = out;
    p.aux = aux;
    p.tab = pl.xw_tab;
    p.twy = pl.twy;
    p.Z = pl.d.Z;
    p.Y = pl.d.Y;
    p.XP = pl.d.XP;
    p.eps = eps;
    p.wz = wz;
    p.wx = wx;
    if (!pl.x3) return pl.d.M == 1024 ? launch_xw_m<10>(ctx, p, mode) : (pl.d.M == 512 ? launch_xw_m<9>(ctx, p, mode) : launch_xw_m<8>(ctx, p, mode));
    return pl.d.M == 1536 ? launch_x3_m<9>(ctx, p, mode) : launch_x3_m<8>(ctx, p, mode);
}

// d_p: the data on the box, wrap-extended like the estimate (lo below, hi above): the first pass clips it into est_a
// (e0 = max(d, 0)) and transforms it in one go.  est_a / est_b alternate; the last update is stored straight into `out`, the
// UNPADDED (N[0], N[1], N[2]) result volume.
int fftconv_richardson_lucy_wrap(bh_ctx* ctx, const ConvPlan& pl, const float* d_p, const cf* otf, bool otf_real, cf* spec_a,
                                 cf* spec_b, float* est_a, float* est_b, const int64_t N[3], const int64_t K[3], int iterations,
                                 float eps, float* out) {
    const int CONV = otf_real ? COL_FILTER : COL_CONV, CORR = otf_real ? COL_FILTER : COL_CORR;
    const int64_t P[3] = {pl.d.Z, pl.d.Y, pl.d.X};
    xw::Params::Wrap we[3], wr[3];  // extension of the estimate (lo below, hi above) and of the ratio (hi below, lo above)
    for (int a = 0; a < 3; ++a) {
        const bool padded = P[a] != N[a];
        const int lo = padded ? (int)(K[a] - 1 - K[a] / 2) : 0, hi = padded ? (int)(K[a] / 2) : 0;
        we[a] = xw::Params::Wrap{(int)N[a], lo, lo, hi};
        wr[a] = xw::Params::Wrap{(int)N[a], lo, hi, lo};
    }
    float *cur = est_a, *nxt = est_b;
    BH_TRY(launch_x(ctx, pl, false, 0, d_p, spec_a, cur, nullptr, 0.f));  // est = max(d, 0) written by the same pass
    for (int it = 0; it < iterations; ++it) {
        BH_TRY(launch_col(ctx, pl, COL_FWD, false, spec_a, nullptr, 1.f));
        BH_TRY(launch_col(ctx, pl, CONV, true, spec_a, otf, 1.f));
        BH_TRY(launch_col(ctx, pl, COL_INV, false, spec_a, nullptr, 1.f));
        BH_TRY(launch_x3_wrap(ctx, pl, xw::FUSED_RATIO_WRAP, spec_a, spec_b, nullptr, d_p, eps, wr[0], wr[2]));
        BH_TRY(launch_col(ctx, pl, COL_FWD, false, spec_b, nullptr, 1.f));
        BH_TRY(launch_col(ctx, pl, CORR, true, spec_b, otf, 1.f));
        BH_TRY(launch_col(ctx, pl, COL_INV, false, spec_b, nullptr, 1.f));
        if (it + 1 == iterations) {  // the last update is needed on the volume's own voxels only: stored cropped
            BH_TRY(launch_x3_wrap(ctx, pl, xw::INV_UPDATE_CROP, spec_b, nullptr, out, cur, eps, we[0], we[2]));
        } else {
            BH_TRY(launch_x3_wrap(ctx, pl, xw::FUSED_UPDATE_WRAP, spec_b, spec_a, nxt, cur, eps, we[0], we[2]));
            std::swap(cur, nxt);
        }
    }
    return BH_OK;
}

// Bare transform pair for callers that do their own spectral arithmetic (phase cross-correlation): forward leaves the
// true DFT coefficients in the engine's scrambled half-spectrum layout, inverse returns real space scaled by V/2
// (multiply the spectrum by 2/V first for a normalised irfftn).  Any element-wise operation that treats both spectra
// alike is layout-agnostic.
int fftconv_forward(bh_ctx* ctx, const ConvPlan& pl, const float* in, cf* spec) {
    BH_TRY(launch_x(ctx, pl, false, 0, in, spec, nullptr, nullptr, 0.f));
    BH_TRY(launch_col(ctx, pl, COL_FWD, false, spec, nullptr, 1.f));
    BH_TRY(launch_col(ctx, pl, COL_FWD, true, spec, nullptr, 1.f));
    return BH_OK;
}
int fftconv_inverse(bh_ctx* ctx, const ConvPlan& pl, cf* spec, float* out) {
    BH_TRY(launch_col(ctx, pl, COL_INV, true, spec, nullptr, 1.f));
    BH_TRY(launch_col(ctx, pl, COL_INV, false, spec, nullptr, 1.f));
    BH_TRY(launch_x(ctx, pl, true, XE_STORE, nullptr, spec, out, nullptr, 0.f));
    return BH_OK;
}

// corr = irfft( rfft(ref) * conj(rfft(mov)) / norm ) with the product inside the Z pass of ONE image's transform, the other
// image's finished spectrum (fftconv_forward) being the multiplier: 5 passes per call once that spectrum is in hand, 8 with
// it (instead of 10 for two forward transforms, a product pass and an inverse transform).  Spectrum layout and scaling as
// fftconv_forward / fftconv_inverse: scale = 2 / V.
// fixed: the finished spectrum of the stored image; fixed_is_mov says whether that image is the product's second (conjugated)
// factor.  roll: the Z pass also writes img's forward spectrum over `fixed` (each thread replaces the rows it has just read), so
// that img is the stored image of the next call — the "previous timepoint" reference of the stabilisation estimate for one
// extra store stream.  corr == nullptr (rows the wave-private xw kernels take: fftconv_pcc_peak_only): the correlation volume
// is not stored — the last pass leaves *npartial argmax candidates of |corr| in `partial` (at most 8 per compute unit) for the
// caller's final reduction.
bool fftconv_pcc_peak_only(const ConvPlan& pl) { return pl.xw && !pl.x3 && getenv("BH_PCC_NO_FUSED_PEAK") == nullptr; }
int fftconv_pcc_apply(bh_ctx* ctx, const ConvPlan& pl, const float* img, cf* fixed, bool fixed_is_mov, bool roll, cf* s2, int norm,
                      float scale, float* corr, ArgMax* partial, int* npartial) {
    BH_REQUIRE(corr || (partial && npartial && fftconv_pcc_peak_only(pl)), "internal: correlation volume or peak buffer required");
    BH_TRY(launch_x(ctx, pl, false, 0, img, s2, nullptr, nullptr, 0.f));
    BH_TRY(launch_col(ctx, pl, COL_FWD, false, s2, nullptr, 1.f));
    BH_TRY(launch_col(ctx, pl, COL_PCC, true, s2, fixed, scale, norm, fixed_is_mov ? 1 : 0, roll ? fixed : nullptr));
    BH_TRY(launch_col(ctx, pl, COL_INV, false, s2, nullptr, 1.f));
    if (!corr) return launch_xw_argmax(ctx, pl, s2, partial, npartial);
    BH_TRY(launch_x(ctx, pl, true, XE_STORE, nullptr, s2, corr, nullptr, 0.f));
    return BH_OK;
}
int fftconv_pcc(bh_ctx* ctx, const ConvPlan& pl, const float* ref, const float* mov, cf* s1, cf* s2, int norm, float scale, float* corr,
                ArgMax* partial, int* npartial) {
    BH_TRY(fftconv_forward(ctx, pl, ref, s1));
    return fftconv_pcc_apply(ctx, pl, mov, s1, false, false, s2, norm, scale, corr, partial, npartial);
}

// out = irfft( rfft(in) * H/(H^2+reg) ), H = tf_full (natural order, real, even)
int fftconv_tikhonov(bh_ctx* ctx, const ConvPlan& pl, const float* in, const float* tf_full, float reg, cf* spec,
                     float* filt, float* out) {
    const double V = (double)pl.d.Z * pl.d.Y * pl.d.X;
    const int grid = ctx->num_cus * 8;
    hipLaunchKernelGGL(tikhonov_filter_rows_kernel, dim3(grid), dim3(256), pl.d.XP * sizeof(float), ctx->stream, tf_full,
                       filt, pl.d, reg, (float)(2.0 / V), pl.xw ? pl.xw_col : nullptr);
    BH_CHECK_HIP(hipGetLastError());
    BH_TRY(launch_x(ctx, pl, false, 0, in, spec, nullptr, nullptr, 0.f));
    BH_TRY(launch_col(ctx, pl, COL_FWD, false, spec, nullptr, 1.f));
    BH_TRY(launch_col(ctx, pl, COL_FILTER, true, spec, reinterpret_cast<const cf*>(filt), 1.f));
    BH_TRY(launch_col(ctx, pl, COL_INV, false, spec, nullptr, 1.f));
    BH_TRY(launch_x(ctx, pl, true, XE_STORE, nullptr, spec, out, nullptr, 0.f));
    return BH_OK;
}

}  // namespace bh
