// Reproducer of a rocFFT (ROCm 7.2, gfx950) plan-state defect, independent of libbhcore: planning the 3-D real transform
// (8,128,64) and then (4,32,256) in one process leaves the second with round-trip errors of order 1 (every other pair in
// the list below is fine).  Both are small powers of two, which libbhcore never sends to hipFFT (its own engine handles
// them) unless BH_FFT_BACKEND=hipfft forces it; see DESIGN.md.
//   hipcc -O2 --offload-arch=gfx950 tools/hipfft_two_plans.cpp -o /tmp/two_plans -lhipfft && /tmp/two_plans
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
struct P { hipfftHandle r2c, c2r; int Z, Y, X; };
static P mk(int Z, int Y, int X) { P p{0, 0, Z, Y, X}; hipfftPlan3d(&p.r2c, Z, Y, X, HIPFFT_R2C); hipfftPlan3d(&p.c2r, Z, Y, X, HIPFFT_C2R); return p; }
static double rt(const P& p, float* dreal, hipfftComplex* dspec, float* dback) {
    const size_t V = (size_t)p.Z * p.Y * p.X;
    std::vector<float> h(V), b(V);
    for (size_t i = 0; i < V; ++i) h[i] = (float)(rand() % 1000) / 10.0f;
    hipMemcpy(dreal, h.data(), V * 4, hipMemcpyHostToDevice);
    hipfftExecR2C(p.r2c, dreal, dspec);
    hipfftExecC2R(p.c2r, dspec, dback);
    hipMemcpy(b.data(), dback, V * 4, hipMemcpyDeviceToHost);
    double e = 0, m = 0;
    for (size_t i = 0; i < V; ++i) { e = fmax(e, fabs(b[i] / (double)V - h[i])); m = fmax(m, fabs(h[i])); }
    return e / m;
}
int main(int argc, char** argv) {
    float *a, *c; hipfftComplex* s;
    hipMalloc(&a, 1 << 28); hipMalloc(&c, 1 << 28); hipMalloc(&s, 1 << 29);
    const int shapes[][3] = {{8,128,64},{4,32,256},{15,21,25},{16,20,24},{37,53,71},{4,32,256},{32,48,64},{8,16,32},{19,40,134},{4,128,256},
                             {16,64,1024},{8,128,64},{33,17,12},{20,14,9},{64,48,40},{86,256,380},{2,16,32},{16,16,16},{8,12,10},{9,7,11}};
    const int n = sizeof(shapes) / sizeof(shapes[0]);
    std::vector<P> plans;
    for (int i = 0; i < n; ++i) {
        plans.push_back(mk(shapes[i][0], shapes[i][1], shapes[i][2]));
        printf("create+run (%d,%d,%d): %.2e\n", shapes[i][0], shapes[i][1], shapes[i][2], rt(plans.back(), a, s, c));
    }
    for (int i = 0; i < n; ++i) printf("rerun      (%d,%d,%d): %.2e\n", shapes[i][0], shapes[i][1], shapes[i][2], rt(plans[i], a, s, c));
    return 0;
}
