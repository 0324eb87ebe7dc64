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

struct FftPlans {
    hipfftHandle r2c = 0;
    hipfftHandle c2r = 0;
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
    // Richardson-Lucy OTF cache: the OTF in "fc_otf" belongs to the PSF with this content hash / these shapes
    bool otf_valid = false;
    unsigned long long otf_hash = 0;
    int64_t otf_dims[6] = {0, 0, 0, 0, 0, 0};
};

namespace bh {

// returns a device buffer of at least `bytes`, cached under `name`
int get_scratch(bh_ctx* ctx, const char* name, size_t bytes, void** out);
int get_plans(bh_ctx* ctx, int64_t Z, int64_t Y, int64_t X, FftPlans** out);

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

}  // namespace bh
