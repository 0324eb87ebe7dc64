// Sweep for the rocFFT plan-state defect (tools/hipfft_two_plans.cpp): many hipfftPlan3d real plans of random 7-smooth
// shapes kept alive in one process, round trip checked at creation and again after all exist.  Prints every failing shape
// and the failure count split by "all three extents are powers of two" / "not".
//   hipcc -O2 --offload-arch=gfx950 tools/hipfft_plan3d_sweep.cpp -o /tmp/sweep -lhipfft && /tmp/sweep [seed] [count]
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
struct P { hipfftHandle r2c, c2r; int Z, Y, X; };
static bool pow2(int n) { return (n & (n - 1)) == 0; }
static bool smooth(int n) { for (int p : {2, 3, 5, 7}) while (n % p == 0) n /= p; return n == 1; }
static double rt(const P& p, float* a, hipfftComplex* s, float* c) {
    const size_t V = (size_t)p.Z * p.Y * p.X;
    std::vector<float> h(V), b(V);
    for (size_t i = 0; i < V; ++i) h[i] = (float)(rand() % 1000) / 10.0f;
    (void)hipMemcpy(a, h.data(), V * 4, hipMemcpyHostToDevice);
    hipfftExecR2C(p.r2c, a, s);
    hipfftExecC2R(p.c2r, s, c);
    (void)hipMemcpy(b.data(), c, V * 4, hipMemcpyDeviceToHost);
    double e = 0, m = 0;
    for (size_t i = 0; i < V; ++i) { e = fmax(e, fabs(b[i] / (double)V - h[i])); m = fmax(m, fabs(h[i])); }
    return e / m;
}
int main(int argc, char** argv) {
    const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1u;
    const int count = argc > 2 ? atoi(argv[2]) : 150;
    srand(seed);
    std::vector<int> sm, p2;
    for (int n = 2; n <= 512; ++n) { if (smooth(n)) sm.push_back(n); if (pow2(n)) p2.push_back(n); }
    float *a, *c; hipfftComplex* s;
    (void)hipMalloc(&a, 1 << 26); (void)hipMalloc(&c, 1 << 26); (void)hipMalloc(&s, 1 << 27);
    std::vector<P> plans;
    int bad[2] = {0, 0}, tot[2] = {0, 0};
    auto check = [&](const P& p, const char* when) {
        const double e = rt(p, a, s, c);
        const int cls = pow2(p.Z) && pow2(p.Y) && pow2(p.X);
        if (!(e < 1e-4)) { ++bad[cls]; printf("%s FAIL (%d,%d,%d) %s: %.2e\n", when, p.Z, p.Y, p.X, cls ? "pow2" : "mixed", e); }
    };
    while ((int)plans.size() < count) {
        const bool allp2 = rand() % 4 == 0;
        const std::vector<int>& src = allp2 ? p2 : sm;
        P p{0, 0, src[rand() % src.size()], src[rand() % src.size()], src[rand() % src.size()]};
        if ((size_t)p.Z * p.Y * p.X > (1u << 22)) continue;
        if (hipfftPlan3d(&p.r2c, p.Z, p.Y, p.X, HIPFFT_R2C) != HIPFFT_SUCCESS || hipfftPlan3d(&p.c2r, p.Z, p.Y, p.X, HIPFFT_C2R) != HIPFFT_SUCCESS) {
            printf("plan failed (%d,%d,%d)\n", p.Z, p.Y, p.X); continue;
        }
        ++tot[pow2(p.Z) && pow2(p.Y) && pow2(p.X)];
        plans.push_back(p);
        check(p, "create");
    }
    for (const P& p : plans) check(p, "rerun ");
    printf("seed %u: %d mixed-radix plans, %d failures (create+rerun); %d power-of-two plans, %d failures\n", seed, tot[0], bad[0], tot[1], bad[1]);
    return 0;
}
