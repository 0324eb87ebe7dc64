// Micro-benchmark: HBM throughput of strided tile copies (the access pattern of a non-contiguous
// FFT axis pass / the deskew stage) versus a plain streaming copy.  Build: hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

__global__ __launch_bounds__(256) void stream_copy(const float4* in, float4* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}

// array viewed as [outer][N][inner bytes]; tile = N rows x SEG bytes; one WG (NT threads) per tile
template <int SEG, int NT, int LDSB>
__global__ __launch_bounds__(NT) void tile_copy(const char* in, char* out, int N, size_t inner, size_t tiles_per_outer) {
    extern __shared__ char pad[];  // occupancy limiter only
    if (LDSB && threadIdx.x == 9999) pad[0] = 1;
    constexpr int LPS = SEG / 16;          // lanes per segment
    constexpr int RPI = NT / LPS;          // rows per iteration
    const size_t tile = blockIdx.x;
    const size_t outer = tile / tiles_per_outer, t = tile % tiles_per_outer;
    const size_t base = outer * (size_t)N * inner + t * SEG;
    const int lane = threadIdx.x % LPS, r0 = threadIdx.x / LPS;
    float4 v[16];
    const int iters = N / RPI;  // must be <= 16
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (i < iters) v[i] = *(const float4*)(in + base + (size_t)(r0 + i * RPI) * inner + lane * 16);
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (i < iters) *(float4*)(out + base + (size_t)(r0 + i * RPI) * inner + lane * 16) = v[i];
}

template <int SEG, int NT, int LDSB>
double run_tile(const char* in, char* out, size_t total, int N, size_t inner, int reps) {
    size_t tiles_per_outer = inner / SEG;
    size_t nouter = total / ((size_t)N * inner);
    size_t ntiles = nouter * tiles_per_outer;
    if (LDSB > 64 * 1024) CK(hipFuncSetAttribute((const void*)tile_copy<SEG, NT, LDSB>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    tile_copy<SEG, NT, LDSB><<<ntiles, NT, LDSB>>>(in, out, N, inner, tiles_per_outer);
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) tile_copy<SEG, NT, LDSB><<<ntiles, NT, LDSB>>>(in, out, N, inner, tiles_per_outer);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return 2.0 * nouter * N * inner / (ms / reps * 1e-3) / 1e12;
}

int main() {
    const size_t total = (size_t)8 << 30;  // 8 GiB each way
    char *in, *out; CK(hipMalloc(&in, total)); CK(hipMalloc(&out, total));
    CK(hipMemset(in, 1, total)); CK(hipMemset(out, 0, total));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int g : {2048, 8192, 65536}) {
        stream_copy<<<g, 256>>>((float4*)in, (float4*)out, total / 16);
        CK(hipEventRecord(a));
        for (int r = 0; r < 5; ++r) stream_copy<<<g, 256>>>((float4*)in, (float4*)out, total / 16);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("stream_copy grid %6d: %.2f TB/s (r+w)\n", g, 2.0 * total / (ms / 5 * 1e-3) / 1e12);
    }
    // Y-axis-like: inner = 8192 B (1024 complex), N rows = 1024 or 2048 ; Z-axis-like: inner = 16 MiB, N = 512
    struct Cfg { int N; size_t inner; const char* name; } cfgs[] = {
        {1024, 8192, "N=1024 stride 8 KiB"}, {2048, 8192, "N=2048 stride 8 KiB"}, {512, (size_t)16 << 20, "N=512 stride 16 MiB"}};
    for (auto& c : cfgs) {
        printf("%s\n", c.name);
        if (c.N <= 1024) {
            printf("  seg  64 B, 256 thr, 4 WG/CU : %.2f TB/s\n", run_tile<64, 256, 36 * 1024>(in, out, total, c.N, c.inner, 5));
        }
        printf("  seg  64 B, 1024 thr, 2 WG/CU: %.2f TB/s\n", run_tile<64, 1024, 72 * 1024>(in, out, total, c.N, c.inner, 5));
        printf("  seg 128 B, 1024 thr, 2 WG/CU: %.2f TB/s\n", run_tile<128, 1024, 72 * 1024>(in, out, total, c.N, c.inner, 5));
        printf("  seg 128 B, 1024 thr, 1 WG/CU: %.2f TB/s\n", run_tile<128, 1024, 130 * 1024>(in, out, total, c.N, c.inner, 5));
        printf("  seg 256 B, 1024 thr, 1 WG/CU: %.2f TB/s\n", run_tile<256, 1024, 130 * 1024>(in, out, total, c.N, c.inner, 5));
        printf("  seg 256 B, 1024 thr, 2 WG/CU: %.2f TB/s\n", run_tile<256, 1024, 72 * 1024>(in, out, total, c.N, c.inner, 5));
        printf("  seg 512 B, 1024 thr, 2 WG/CU: %.2f TB/s\n", run_tile<512, 1024, 72 * 1024>(in, out, total, c.N, c.inner, 5));
    }
    return 0;
}
