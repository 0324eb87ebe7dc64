// Micro-benchmark 3: does "each workgroup owns a private contiguous chunk" cap HBM throughput?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

template <int NT>
__global__ __launch_bounds__(NT) void stream_copy(const float4* in, float4* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) out[i] = in[i];
}
// WG b copies chunk b (CH bytes), U float4 per thread per trip
template <int NT, int U>
__global__ __launch_bounds__(NT) void chunk_copy(const float4* in, float4* out, size_t ch16, int lds_dummy) {
    extern __shared__ char pad[];
    if (lds_dummy == 12345) pad[threadIdx.x] = 1;
    const float4* src = in + (size_t)blockIdx.x * ch16;
    float4* dst = out + (size_t)blockIdx.x * ch16;
    for (size_t i = threadIdx.x; i < ch16; i += (size_t)NT * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + (size_t)u * NT < ch16) v[u] = src[i + (size_t)u * NT];
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + (size_t)u * NT < ch16) dst[i + (size_t)u * NT] = v[u];
    }
}
template <typename F> double timeit(F f, double bytes) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipEventRecord(a)); for (int r = 0; r < 4; ++r) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); CK(hipGetLastError());
    return bytes / (ms / 4 * 1e-3) / 1e12;
}
int main() {
    const size_t total = (size_t)8 << 30;
    char *in, *out; CK(hipMalloc(&in, total)); CK(hipMalloc(&out, total));
    CK(hipMemset(in, 1, total)); CK(hipMemset(out, 0, total));
    const size_t n16 = total / 16;
    printf("stream 256thr grid 65536          : %.2f\n", timeit([&]{ stream_copy<256><<<65536, 256>>>((float4*)in, (float4*)out, n16); }, 2.0 * total));
    printf("stream 1024thr grid 16384         : %.2f\n", timeit([&]{ stream_copy<1024><<<16384, 1024>>>((float4*)in, (float4*)out, n16); }, 2.0 * total));
    printf("stream 256thr grid 2048 (persist.): %.2f\n", timeit([&]{ stream_copy<256><<<2048, 256>>>((float4*)in, (float4*)out, n16); }, 2.0 * total));
    for (size_t ch : {(size_t)16 << 10, (size_t)64 << 10, (size_t)128 << 10, (size_t)512 << 10, (size_t)2 << 20}) {
        const size_t ch16 = ch / 16; const unsigned nb = (unsigned)(total / ch);
        printf("chunk %7zu B: 256thr U1 %.2f", ch, timeit([&]{ chunk_copy<256, 1><<<nb, 256, 0>>>((float4*)in, (float4*)out, ch16, 0); }, 2.0 * total));
        printf("  256thr U4 %.2f", timeit([&]{ chunk_copy<256, 4><<<nb, 256, 0>>>((float4*)in, (float4*)out, ch16, 0); }, 2.0 * total));
        printf("  1024thr U1 %.2f", timeit([&]{ chunk_copy<1024, 1><<<nb, 1024, 0>>>((float4*)in, (float4*)out, ch16, 0); }, 2.0 * total));
        printf("  1024thr U8 %.2f", timeit([&]{ chunk_copy<1024, 8><<<nb, 1024, 0>>>((float4*)in, (float4*)out, ch16, 0); }, 2.0 * total));
        printf("  1024thr U8 lds72K %.2f\n", timeit([&]{ chunk_copy<1024, 8><<<nb, 1024, 72 * 1024>>>((float4*)in, (float4*)out, ch16, 0); }, 2.0 * total));
    }
    // read-only and write-only rates
    return 0;
}
