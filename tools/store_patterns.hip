// Which store pattern reaches the rate of a plain fill (6.7-6.9 TB/s) on MI355X?  17-GB float buffer, 16-byte stores.
//   A  persistent, every wavefront walks its own contiguous range (1 KB per step)
//   B  one 256-thread block per 16 KB, no loop (the shape of torch's fill_)
//   C  persistent, grid-stride: wavefront w writes the 1-KB pieces w, w + nw, w + 2 nw, ...
//   D  persistent, grid-stride with 16-KB pieces per block
// hipcc -O3 --offload-arch=gfx950 tools/store_patterns.hip -o /tmp/sp && /tmp/sp
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void kA(float4* d, long n4, float v) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
    const long per = (n4 / 64 + nw - 1) / nw * 64;
    const float4 v4 = make_float4(v, v, v, v);
    for (long i = wave * per + lane; i < min(n4, (wave + 1) * per); i += 64) d[i] = v4;
}
__global__ __launch_bounds__(256) void kB(float4* d, long n4, float v) {
    const float4 v4 = make_float4(v, v, v, v);
    const long base = (long)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (base + k * 256 < n4) d[base + k * 256] = v4;
}
__global__ __launch_bounds__(256) void kC(float4* d, long n4, float v) {
    const float4 v4 = make_float4(v, v, v, v);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) d[i] = v4;
}
__global__ __launch_bounds__(256) void kD(float4* d, long n4, float v) {
    const float4 v4 = make_float4(v, v, v, v);
    for (long b = (long)blockIdx.x * 1024; b < n4; b += (long)gridDim.x * 1024) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (b + threadIdx.x + k * 256 < n4) d[b + threadIdx.x + k * 256] = v4;
    }
}
int main() {
    const long n4 = 683l * 2048 * 3034 / 4;
    float4* d;
    hipMalloc(&d, n4 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char* name, auto launch) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%-40s %.3f ms = %.2f TB/s\n", name, ms, (double)n4 * 16 / ms / 1e9);
    };
    for (int grid : {512, 2048, 8192}) {
        char nm[64];
        snprintf(nm, 64, "A contiguous per wave, grid %d", grid);
        time(nm, [&] { hipLaunchKernelGGL(kA, dim3(grid), dim3(256), 0, 0, d, n4, 1.f); });
        snprintf(nm, 64, "C grid-stride 4 KB per block, grid %d", grid);
        time(nm, [&] { hipLaunchKernelGGL(kC, dim3(grid), dim3(256), 0, 0, d, n4, 1.f); });
        snprintf(nm, 64, "D grid-stride 16 KB per block, grid %d", grid);
        time(nm, [&] { hipLaunchKernelGGL(kD, dim3(grid), dim3(256), 0, 0, d, n4, 1.f); });
    }
    time("B one block per 16 KB", [&] { hipLaunchKernelGGL(kB, dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, 0, d, n4, 1.f); });
    return 0;
}
