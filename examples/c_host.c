/* A plain-C host of libbhcore.so: no Python, no torch — the drop-in boundary is the C-ABI of include/bhcore.h.
 *
 *   gcc -std=c99 -Iinclude examples/c_host.c -Lbiahub_amd -lbhcore -Wl,-rpath,$PWD/biahub_amd -lm -o c_host
 *   ./c_host            geometry query only (no GPU needed)
 *   ./c_host gpu        deskew a small uint16 stack on device 0 and check two invariants on the host
 *
 * The same calls are what a cgo / JNI / ctypes binding would make.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bhcore.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int s_ = (call);                                                         \
        if (s_ != BH_OK) {                                                       \
            fprintf(stderr, "%s -> %d: %s\n", #call, s_, bh_last_error());       \
            return 1;                                                            \
        }                                                                        \
    } while (0)

int main(int argc, char** argv) {
    if (bh_abi_version() != BH_ABI_VERSION) {
        fprintf(stderr, "ABI mismatch: library %d, header %d\n", bh_abi_version(), BH_ABI_VERSION);
        return 1;
    }
    /* settings/example_deskew_settings.yml of the reference on BASELINE config 1 */
    const int64_t Z = 64, Y = 256, X = 256;
    int64_t out_shape[3];
    double voxel[3];
    CHECK(bh_deskew_shape(Z, Y, X, 36.17, 0.371, 1, 3, 0.116, out_shape, voxel));
    printf("deskewed shape (%lld, %lld, %lld), voxel (%.4f, %.4f, %.4f) um\n", (long long)out_shape[0],
           (long long)out_shape[1], (long long)out_shape[2], voxel[0], voxel[1], voxel[2]);
    if (out_shape[0] != 86 || out_shape[1] != 256 || out_shape[2] != 380) return 2;
    /* the reference's ValueError case (tests/test_cli/test_deskew_cli.py:191-197) surfaces as BH_ERR_INVALID */
    if (bh_deskew_shape(10, 500, 100, 30.0, 0.1, 0, 1, 1.0, out_shape, voxel) != BH_ERR_INVALID) return 3;
    printf("invalid geometry refused: %s\n", bh_last_error());
    CHECK(bh_deskew_shape(Z, Y, X, 36.17, 0.371, 1, 3, 0.116, out_shape, voxel));
    if (argc < 2 || strcmp(argv[1], "gpu") != 0) return 0;

    int ndev = 0;
    CHECK(bh_device_count(&ndev));
    if (ndev < 1) {
        fprintf(stderr, "no GPU visible\n");
        return 4;
    }
    bh_ctx* ctx = NULL;
    CHECK(bh_ctx_create(0, NULL, &ctx));
    const size_t nin = (size_t)(Z * Y * X), nout = (size_t)(out_shape[0] * out_shape[1] * out_shape[2]);
    unsigned short* h_in = (unsigned short*)malloc(nin * sizeof(unsigned short));
    float* h_out = (float*)malloc(nout * sizeof(float));
    for (size_t i = 0; i < nin; ++i) h_in[i] = 500; /* a constant stack */
    void *d_in = NULL, *d_out = NULL;
    CHECK(bh_malloc(&d_in, nin * sizeof(unsigned short)));
    CHECK(bh_malloc(&d_out, nout * sizeof(float)));
    CHECK(bh_memcpy_h2d(ctx, d_in, h_in, nin * sizeof(unsigned short)));
    float fill = 0.0f;
    CHECK(bh_deskew(ctx, d_in, BH_DT_U16, Z, Y, X, 36.17, 0.371, 1, 3, BH_FILL_MEAN, 0.0f, (float*)d_out, &fill));
    CHECK(bh_memcpy_d2h(ctx, h_out, d_out, nout * sizeof(float)));
    /* a constant stack deskews to that constant inside the sheared support and the overhang is filled with the mean of the
     * kept voxels; only the few blended voxels the 3-voxel mask dilation does not reach differ (reference semantics) */
    size_t bad = 0;
    for (size_t i = 0; i < nout; ++i)
        if (fabsf(h_out[i] - 500.0f) > 0.01f) ++bad;
    printf("fill value %.3f, voxels off the constant: %zu of %zu\n", fill, bad, nout);
    CHECK(bh_free(d_in));
    CHECK(bh_free(d_out));
    CHECK(bh_ctx_destroy(ctx));
    free(h_in);
    free(h_out);
    return (fabsf(fill - 500.0f) < 0.1f && bad * 1000 < nout) ? 0 : 5;
}
