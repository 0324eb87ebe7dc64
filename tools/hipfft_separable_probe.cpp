// Probe: the same 3-D real round trip built from a batched 1-D real plan along x and a strided batched 2-D complex plan over
// (z, y) (hipfftPlanMany) instead of hipfftPlan3d,
// on the shape sequence that breaks the 3-D plans (tools/hipfft_two_plans.cpp).
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
struct P { hipfftHandle xr, xi, y, z; int Z, Y, X, Xh; };
static P mk(int Z, int Y, int X) {
    P p; p.Z = Z; p.Y = Y; p.X = X; p.Xh = X / 2 + 1;
    int nx[1] = {X}, ny[1] = {Y}, nz[1] = {Z};
    // x: Z*Y contiguous rows, real X -> complex Xh
    hipfftPlanMany(&p.xr, 1, nx, nullptr, 1, X, nullptr, 1, p.Xh, HIPFFT_R2C, Z * Y);
    hipfftPlanMany(&p.xi, 1, nx, nullptr, 1, p.Xh, nullptr, 1, X, HIPFFT_C2R, Z * Y);
    // (z, y) together: a 2-D C2C over the two slow axes, element stride Xh, one batch entry per x (distance 1)
    int nzy[2] = {Z, Y};
    hipfftPlanMany(&p.y, 2, nzy, nzy, p.Xh, 1, nzy, p.Xh, 1, HIPFFT_C2C, p.Xh);
    p.z = 0; (void)ny; (void)nz;
    return p;
}
static void fwd(const P& p, float* in, hipfftComplex* s) {
    hipfftExecR2C(p.xr, in, s);
    hipfftExecC2C(p.y, s, s, HIPFFT_FORWARD);
}
static void inv(const P& p, hipfftComplex* s, float* out) {
    hipfftExecC2C(p.y, s, s, HIPFFT_BACKWARD);
    hipfftExecC2R(p.xi, s, out);
}
static double rt(const P& p, float* a, hipfftComplex* s, float* c) {
    const size_t V = (size_t)p.Z * p.Y * p.X;
    std::vector<float> h(V), b(V);
    for (size_t i = 0; i < V; ++i) h[i] = (float)(rand() % 1000) / 10.0f;
    (void)hipMemcpy(a, h.data(), V * 4, hipMemcpyHostToDevice);
    fwd(p, a, s); inv(p, s, c);
    (void)hipMemcpy(b.data(), c, V * 4, hipMemcpyDeviceToHost);
    double e = 0, m = 0;
    for (size_t i = 0; i < V; ++i) { e = fmax(e, fabs(b[i] / (double)V - h[i])); m = fmax(m, fabs(h[i])); }
    return e / m;
}
int main() {
    float *a, *c; hipfftComplex* s;
    (void)hipMalloc(&a, 1 << 28); (void)hipMalloc(&c, 1 << 28); (void)hipMalloc(&s, 1 << 29);
    const int shapes[][3] = {{16,32,64},{8,64,128},{8,128,64},{4,32,256},{64,64,64},{15,21,25},{37,53,71},{4,32,256},{32,48,64},{64,64,64},{16,64,1024},{8,128,64}};
    const int n = sizeof(shapes) / sizeof(shapes[0]);
    std::vector<P> plans;
    for (int i = 0; i < n; ++i) { plans.push_back(mk(shapes[i][0], shapes[i][1], shapes[i][2])); printf("create+run (%d,%d,%d): %.2e\n", shapes[i][0], shapes[i][1], shapes[i][2], rt(plans.back(), a, s, c)); }
    for (int i = 0; i < n; ++i) printf("rerun      (%d,%d,%d): %.2e\n", shapes[i][0], shapes[i][1], shapes[i][2], rt(plans[i], a, s, c));
    return 0;
}
