// Shared host-side plumbing for libbhcore (context, errors, workspace, timing).
#pragma once

#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "bhcore.h"

namespace bh {

void set_error(const char* fmt, ...);

#define BH_CHECK_HIP(expr)                                                                   \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            bh::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,   \
                          __LINE__);                                                         \
            return _e == hipErrorOutOfMemory ? BH_ERR_NOMEM : BH_ERR_HIP;                    \
        }                                                                                    \
    } while (0)

#define BH_CHECK_FFT(expr)                                                                   \
    do {                                                                                     \
        hipfftResult _r = (expr);                                                            \
        if (_r != HIPFFT_SUCCESS) {                                                          \
            bh::set_error("%s failed: hipfftResult %d (%s:%d)", #expr, (int)_r, __FILE__,    \
                          __LINE__);                                                         \
            return _r == HIPFFT_ALLOC_FAILED ? BH_ERR_NOMEM : BH_ERR_HIP;                    \
        }                                                                                    \
    } while (0)

#define BH_REQUIRE(cond, ...)              \
    do {                                   \
        if (!(cond)) {                     \
            bh::set_error(__VA_ARGS__);    \
            return BH_ERR_INVALID;         \
        }                                  \
    } while (0)

#define BH_TRY(expr)                 \
    do {                             \
        int _s = (expr);             \
        if (_s != BH_OK) return _s;  \
    } while (0)

enum TimerSlot { T_DESKEW = 0, T_FILL, T_RL_TOTAL, T_TIKHONOV, T_AFFINE, T_CROPFLIP, T_RL_ITER, T_TF, T_FLATFIELD, T_COUNT };

// Library plans of the 3-D real transform of a (Z, Y, X) volume.  Shapes with a non-power-of-two extent (what reaches the
// library in production: the fused engine in fftconv.hip takes the power-of-two ones) use hipfftPlan3d.  All-power-of-two
// shapes (only here when BH_FFT_BACKEND=hipfft forces it, or beyond the engine's size limits) are decomposed into batched
// 1-D real transforms along x and one strided batched 2-D complex transform over (z, y): rocFFT 1.0.36 (ROCm 7.2) returns
// wrong 3-D real transforms for some power-of-two shapes once 3-D plans of certain other power-of-two shapes exist in the
// process (tools/hipfft_two_plans.cpp; tools/hipfft_plan3d_sweep.cpp: 8 of 131 power-of-two plans wrong, 0 of 319
// mixed-radix ones); the decomposition is immune (tools/hipfft_separable_probe.cpp) but ~2.5x slower at large sizes.
struct FftPlans {
    bool separable = false;
    hipfftHandle r2c = 0, c2r = 0;  // 3-D plans (separable == false)
    hipfftHandle xr = 0;            // x: Z*Y rows, real X -> complex X/2+1
    hipfftHandle xi = 0;            // x: complex X/2+1 -> real X
    hipfftHandle zy = 0;            // (z, y): complex, element stride X/2+1, one batch entry per x
    void* work = nullptr;
    size_t work_bytes = 0;
};

// Named, grow-only device scratch buffers owned by the context.
struct Scratch {
    void* ptr = nullptr;
    size_t bytes = 0;
};

}  // namespace bh

struct bh_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool timing = false;
    hipEvent_t ev[2 * bh::T_COUNT] = {};
    bool ev_valid[bh::T_COUNT] = {};
    float ms_override[bh::T_COUNT] = {};
    std::map<std::tuple<int64_t, int64_t, int64_t>, bh::FftPlans> plans;
    std::map<std::string, bh::Scratch> scratch;
    int num_cus = 256;
    int deskew_path = 0;     // how the last bh_deskew filled the overhang: 0 mask pipeline (or no fill), 1 one-pass (deskew_rows.inc)
    int plans_replaced = 0;  // 3-D library plans that failed their self-check and were rebuilt decomposed (context.hip)
    // Richardson-Lucy OTF cache: the OTF in "fc_otf" belongs to the PSF kept in "rl_psf_kept" (compared byte for byte on every
    // call; the hash is informational) / these shapes / this spectrum layout
    bool otf_valid = false;
    unsigned long long otf_hash = 0;
    int otf_tag = 0;  // spectrum layout of the plan that built it (fftconv_plan_tag)
    void* spec_tuned = nullptr;  // the "fc_spec" allocation fftconv_tune_spectrum chose (or accepted) for this context
    int64_t otf_dims[6] = {0, 0, 0, 0, 0, 0};
};

namespace bh {

// returns a device buffer of at least `bytes`, cached under `name`
int get_scratch(bh_ctx* ctx, const char* name, size_t bytes, void** out);
int free_scratch(bh_ctx* ctx, const char* name);
// the workspace's allocator (hipMalloc, or the virtual-memory API under BH_ALLOC_VMM_MB: context.hip)
hipError_t dev_alloc(int device, size_t bytes, void** out);
hipError_t dev_free(void* p);
bool dev_alloc_is_shuffled();  // the default layout is in force (not switched off by BH_ALLOC_VMM_MB=0)
bool dev_block_is_vmm(const void* p);  // this block was built from mapped chunks (large enough for the layout)
int get_plans(bh_ctx* ctx, int64_t Z, int64_t Y, int64_t X, FftPlans** out);
// real (Z,Y,X) -> half spectrum (Z,Y,X/2+1), unnormalised
int fft_forward(const FftPlans* pl, const float* real, float2* spec);
// half spectrum -> real, unnormalised; the spectrum buffer is overwritten
int fft_inverse(const FftPlans* pl, float2* spec, float* real);

struct ScopedTimer {
    bh_ctx* ctx;
    int slot;
    ScopedTimer(bh_ctx* c, int s) : ctx(c), slot(s) {
        if (ctx->timing) {
            ctx->ev_valid[slot] = false;
            (void)hipEventRecord(ctx->ev[2 * slot], ctx->stream);
        }
    }
    ~ScopedTimer() {
        if (ctx->timing) {
            (void)hipEventRecord(ctx->ev[2 * slot + 1], ctx->stream);
            ctx->ev_valid[slot] = true;
        }
    }
};

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Gigabyte blocks that prepared handles own (staged inverse filters, Richardson-Lucy transfer functions): released blocks are
// kept per (device, size) for the next handle of that size, bh_inverse_filter_trim() returns them to the driver (invtf.hip).
void* filter_pool_take(int device, size_t bytes);
void filter_pool_give(int device, size_t bytes, void* p);

// Device-resident result of the overhang fill's reductions (fill.hip; the one-pass deskew of deskew.hip writes `fill` before
// its resampling kernel starts and raises `fallback` when it meets an exact zero that geometry does not explain).
struct FillStats {
    double sum_all;               // sum of every voxel (zeros contribute nothing)
    double sum_shell;             // sum over dilated & ~zero
    unsigned long long n_masked;  // voxels in the dilated mask
    float fill;                   // value written
    int fallback;                 // one-pass deskew: 1 = a data-dependent zero was seen, the mask pipeline re-runs the volume
};

// |value| and flat index of a running argmax (np.argmax semantics: the FIRST occurrence of the maximum wins)
struct ArgMax {
    float v;
    long long i;
};

}  // namespace bh
