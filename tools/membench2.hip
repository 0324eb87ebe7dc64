// Micro-benchmark 2: which side of a transposing pass may be strided?
//   mode 0: strided tile read  -> strided tile write (same tile)          [in-place axis pass]
//   mode 1: strided tile read  -> contiguous write                         [gather side strided]
//   mode 2: contiguous read    -> strided tile write                       [scatter side strided]
// array [outer][N][inner bytes]; tile = N rows x SEG bytes. Contiguous side = the same bytes packed
// as a [tile][N][SEG] block (what a WG would write after transposing into "axis-contiguous" layout).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

template <int SEG, int NT, int MODE, int UU>
__global__ __launch_bounds__(NT) void tile_pass(const char* in, char* out, int N, size_t inner, size_t tiles_per_outer, int lds_dummy) {
    extern __shared__ char pad[];
    if (lds_dummy == 12345) pad[threadIdx.x] = 1;
    constexpr int LPS = SEG / 16;
    constexpr int RPI = NT / LPS;
    const size_t tile = blockIdx.x;
    const size_t outer = tile / tiles_per_outer, t = tile % tiles_per_outer;
    const size_t sbase = outer * (size_t)N * inner + t * SEG;   // strided view
    const size_t cbase = tile * (size_t)N * SEG;                // packed view
    const int lane = threadIdx.x % LPS, r0 = threadIdx.x / LPS;
    for (int rr = 0; rr < N; rr += RPI * UU) {
        float4 v[UU];
#pragma unroll
        for (int i = 0; i < UU; ++i) {
            const int r = rr + r0 + i * RPI;
            if (MODE == 2) v[i] = *(const float4*)(in + cbase + (size_t)r * SEG + lane * 16);
            else v[i] = *(const float4*)(in + sbase + (size_t)r * inner + lane * 16);
        }
#pragma unroll
        for (int i = 0; i < UU; ++i) {
            const int r = rr + r0 + i * RPI;
            if (MODE == 1) *(float4*)(out + cbase + (size_t)r * SEG + lane * 16) = v[i];
            else *(float4*)(out + sbase + (size_t)r * inner + lane * 16) = v[i];
        }
    }
}

template <int SEG, int NT, int MODE, int UU>
double run(const char* in, char* out, size_t total, int N, size_t inner, int ldsb) {
    size_t tiles_per_outer = inner / SEG;
    size_t nouter = total / ((size_t)N * inner);
    size_t ntiles = nouter * tiles_per_outer;
    auto k = tile_pass<SEG, NT, MODE, UU>;
    if (ldsb > 64 * 1024) CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k<<<ntiles, NT, ldsb>>>(in, out, N, inner, tiles_per_outer, 0);
    CK(hipEventRecord(a));
    const int reps = 4;
    for (int r = 0; r < reps; ++r) k<<<ntiles, NT, ldsb>>>(in, out, N, inner, tiles_per_outer, 0);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return 2.0 * nouter * N * inner / (ms / reps * 1e-3) / 1e12;
}

template <int SEG>
void sweep(const char* in, char* out, size_t total, int N, size_t inner) {
    printf("  seg %4d B:", SEG);
    constexpr int U1024 = (SEG == 64) ? 2 : 4;   // N * SEG/16 / 1024 must be a multiple of U
    printf("  s->s %.2f", run<SEG, 1024, 0, U1024>(in, out, total, N, inner, 72 * 1024));
    printf("  s->c %.2f", run<SEG, 1024, 1, U1024>(in, out, total, N, inner, 72 * 1024));
    printf("  c->s %.2f", run<SEG, 1024, 2, U1024>(in, out, total, N, inner, 72 * 1024));
    printf("  s->s(U1) %.2f", run<SEG, 1024, 0, 1>(in, out, total, N, inner, 72 * 1024));
    printf("  s->s(1WG/CU) %.2f", run<SEG, 1024, 0, U1024>(in, out, total, N, inner, 130 * 1024));
    printf("  s->s(256thr,4/CU,U4) %.2f\n", run<SEG, 256, 0, 4>(in, out, total, N, inner, 36 * 1024));
}

int main() {
    const size_t total = (size_t)8 << 30;
    char *in, *out; CK(hipMalloc(&in, total)); CK(hipMalloc(&out, total));
    CK(hipMemset(in, 1, total)); CK(hipMemset(out, 0, total));
    struct Cfg { int N; size_t inner; const char* name; } cfgs[] = {
        {1024, 8192, "N=1024 rows, stride 8 KiB (TB/s r+w)"}, {2048, 16384, "N=2048 rows, stride 16 KiB"},
        {512, (size_t)16 << 20, "N=512 rows, stride 16 MiB"}, {2048, (size_t)4 << 20, "N=2048 rows, stride 4 MiB"}};
    for (auto& c : cfgs) {
        printf("%s\n", c.name);
        sweep<64>(in, out, total, c.N, c.inner);
        sweep<128>(in, out, total, c.N, c.inner);
        sweep<256>(in, out, total, c.N, c.inner);
        sweep<512>(in, out, total, c.N, c.inner);
    }
    return 0;
}
