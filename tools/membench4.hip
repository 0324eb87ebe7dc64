// Micro-benchmark 4: read-only / write-only / copy, outstanding-request sweep.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

template <int NT, int U, int MODE>  // MODE 0 copy, 1 read-only, 2 write-only
__global__ __launch_bounds__(NT) void k(const float4* in, float4* out, size_t ch16, float4* sink) {
    const float4* src = in + (size_t)blockIdx.x * ch16;
    float4* dst = out + (size_t)blockIdx.x * ch16;
    float4 acc = {0, 0, 0, 0};
    for (size_t i = threadIdx.x; i < ch16; i += (size_t)NT * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (MODE != 2) v[u] = src[i + (size_t)u * NT]; else v[u] = make_float4(1.f, 2.f, 3.f, (float)u);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (MODE != 1) dst[i + (size_t)u * NT] = v[u]; else { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
    }
    if (MODE == 1 && acc.x == 123.456f) *sink = acc;
}
template <typename F> double timeit(F f, double bytes) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipEventRecord(a)); for (int r = 0; r < 4; ++r) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); CK(hipGetLastError());
    return bytes / (ms / 4 * 1e-3) / 1e12;
}
template <int NT, int U> void row(const float4* in, float4* out, size_t total, size_t ch, float4* sink) {
    const size_t ch16 = ch / 16; const unsigned nb = (unsigned)(total / ch);
    printf("  %4d thr U%-2d chunk %7zu: copy %.2f  read %.2f  write %.2f\n", NT, U, ch,
           timeit([&]{ k<NT, U, 0><<<nb, NT>>>(in, out, ch16, sink); }, 2.0 * total),
           timeit([&]{ k<NT, U, 1><<<nb, NT>>>(in, out, ch16, sink); }, 1.0 * total),
           timeit([&]{ k<NT, U, 2><<<nb, NT>>>(in, out, ch16, sink); }, 1.0 * total));
}
int main() {
    const size_t total = (size_t)8 << 30;
    char *in, *out; float4* sink; CK(hipMalloc(&in, total)); CK(hipMalloc(&out, total)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(in, 1, total)); CK(hipMemset(out, 0, total));
    const float4* I = (const float4*)in; float4* O = (float4*)out;
    const size_t ch = 128 << 10;
    row<256, 1>(I, O, total, ch, sink); row<256, 2>(I, O, total, ch, sink); row<256, 4>(I, O, total, ch, sink); row<256, 8>(I, O, total, ch, sink);
    row<1024, 1>(I, O, total, ch, sink); row<1024, 2>(I, O, total, ch, sink); row<1024, 4>(I, O, total, ch, sink); row<1024, 8>(I, O, total, ch, sink);
    row<512, 1>(I, O, total, ch, sink); row<512, 2>(I, O, total, ch, sink);
    row<256, 1>(I, O, total, 16 << 10, sink); row<1024, 1>(I, O, total, 16 << 10, sink);  row<64, 1>(I, O, total, 16 << 10, sink); row<64, 4>(I, O, total, 16 << 10, sink);
    return 0;
}
