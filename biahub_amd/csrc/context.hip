// Context, error string, workspace and hipFFT plan cache of libbhcore.
#include "common.hpp"

#include <cstdio>
#include <cstdlib>

#include <atomic>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace bh {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Device memory of the workspace ("Root cause").  Blocks of 2 GiB and more are assembled with the HIP
// virtual-memory API: one reserved address range, physical chunks of 2 MiB created one by one and mapped in a shuffled order —
// physically contiguous gigabytes make the strided streams of the transform passes collide in the HBM channel / bank hash
// (R-L iteration 34.7 -> 31.5-32.7 ms), chunks below 2 MiB lose the page-table fragment (121-202 ms).  Smaller blocks, and
// everything when BH_ALLOC_VMM_MB=0 or when the driver refuses the API, come from hipMalloc.  The knobs of the experiment
// stay: BH_ALLOC_VMM_MB / _KB (chunk size), BH_ALLOC_VMM_SHUFFLE=0 (chunks in order), BH_ALLOC_VMM_SEED=n (one permutation
// for all blocks), BH_ALLOC_VMM_MIN_MB (smallest block built this way), BH_ALLOC_POISON=1, BH_ALLOC_VMM_FREE_VA=1 (dev_free).
struct VmmBlock {
    size_t size = 0, chunk = 0;
    int device = 0;  // the device that owns the physical chunks: synchronised (and made current) around the unmap
    std::vector<hipMemGenericAllocationHandle_t> handles;
};
// Address ranges of released blocks are RETAINED (dev_free explains why): the totals are reported by bh_alloc_layout, and once
// more than BH_ALLOC_VMM_VA_CAP_GB (default 16384 = 16 TiB of the 128 TiB a process has) is held back, further gigabyte blocks
// come from hipMalloc — slower passes, never a wrong result, and the address space cannot run out under a long-lived service
// that rebuilds its workspace per plate.
static std::atomic<unsigned long long> g_va_retained_bytes{0}, g_va_retained_ranges{0};
static void retain_range(size_t size) {
    g_va_retained_bytes.fetch_add(size);
    g_va_retained_ranges.fetch_add(1);
}
static bool va_budget_left() {
    static const unsigned long long cap = (unsigned long long)(getenv("BH_ALLOC_VMM_VA_CAP_GB") ? atoll(getenv("BH_ALLOC_VMM_VA_CAP_GB")) : 16384) << 30;
    return g_va_retained_bytes.load() < cap;
}
struct DeviceGuard {  // makes `device` current for the scope and restores what was current before
    int prev = -1;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) (void)hipSetDevice(device);
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};
// every chunk is unmapped by a call of its own, with exactly the range its hipMemMap call had (an unmap spanning several
// mappings is not something the API promises to take apart)
static hipError_t vmm_unmap_all(void* va, const VmmBlock& blk, size_t mapped_chunks) {
    hipError_t first = hipSuccess;
    for (size_t i = 0; i < mapped_chunks; ++i) {
        const hipError_t e = hipMemUnmap(static_cast<char*>(va) + i * blk.chunk, blk.chunk);
        if (e != hipSuccess && first == hipSuccess) first = e;
    }
    return first;
}
static std::mutex g_vmm_mu;
static std::map<void*, VmmBlock> g_vmm;

static hipError_t vmm_alloc(int device, size_t bytes, size_t chunk, bool shuffle, void** out) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0, gran_rec = 0;
    hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
    if (e != hipSuccess) return e;
    (void)hipMemGetAllocationGranularity(&gran_rec, &prop, hipMemAllocationGranularityRecommended);
    static bool said = false;
    if (!said && getenv("BH_DEBUG_SCRATCH")) {
        said = true;
        fprintf(stderr, "[bh vmm] allocation granularity: minimum %zu, recommended %zu bytes; chunk %zu\n", gran, gran_rec, chunk);
    }
    chunk = std::max(chunk, gran) / gran * gran;
    const size_t n = (bytes + chunk - 1) / chunk, size = n * chunk;
    void* va = nullptr;
    if ((e = hipMemAddressReserve(&va, size, chunk, nullptr, 0)) != hipSuccess) return e;
    VmmBlock blk;
    blk.size = size;
    blk.chunk = chunk;
    blk.device = device;
    std::vector<size_t> slot(n);
    for (size_t i = 0; i < n; ++i) slot[i] = i;
    if (shuffle) {
        // a different permutation for every allocation: buffers of one size that are walked in lockstep (estimate, data,
        // result) must not share their physical pattern either (BH_ALLOC_VMM_SEED=fixed: one permutation for all, the A/B)
        static std::atomic<unsigned long long> counter{0};
        static const char* seed_env = getenv("BH_ALLOC_VMM_SEED");  // an integer: that permutation for every allocation
        unsigned long long st = 0x9E3779B97F4A7C15ull +
                                0xD1B54A32D192ED03ull * (seed_env ? strtoull(seed_env, nullptr, 10) : counter.fetch_add(1) + 1);
        for (size_t i = n - 1; i > 0; --i) {
            st = st * 6364136223846793005ull + 1442695040888963407ull;
            std::swap(slot[i], slot[(size_t)((st >> 33) % (i + 1))]);
        }
    }
    for (size_t i = 0; i < n && e == hipSuccess; ++i) {
        hipMemGenericAllocationHandle_t h;
        if ((e = hipMemCreate(&h, chunk, &prop, 0)) != hipSuccess) break;
        blk.handles.push_back(h);
        e = hipMemMap(static_cast<char*>(va) + slot[i] * chunk, chunk, 0, h, 0);
    }
    if (e == hipSuccess) {
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(va, size, &acc, 1);
    }
    if (e != hipSuccess) {
        // slots are not mapped in address order: unmap whatever is there, chunk by chunk (errors for unmapped slots are expected)
        (void)hipDeviceSynchronize();
        (void)vmm_unmap_all(va, blk, n);
        (void)hipGetLastError();
        for (auto h : blk.handles) (void)hipMemRelease(h);
        // the partly mapped range is retained like a released block's (dev_free): a retry or the hipMalloc fallback must not be
        // handed this range again while translations of the aborted mapping may still be cached
        static const bool free_va = getenv("BH_ALLOC_VMM_FREE_VA") && atoi(getenv("BH_ALLOC_VMM_FREE_VA")) != 0;
        if (free_va) (void)hipMemAddressFree(va, size);
        else retain_range(size);
        return e;
    }
    std::lock_guard<std::mutex> lk(g_vmm_mu);
    g_vmm[va] = std::move(blk);
    *out = va;
    return hipSuccess;
}

static long vmm_chunk_kb() {
    // default: 2-MiB chunks (BH_ALLOC_VMM_MB=0: plain hipMalloc; BH_ALLOC_VMM_KB takes precedence: chunks below 1 MiB)
    static const long kb = getenv("BH_ALLOC_VMM_KB") ? atol(getenv("BH_ALLOC_VMM_KB"))
                           : (getenv("BH_ALLOC_VMM_MB") ? atol(getenv("BH_ALLOC_VMM_MB")) * 1024 : 2048);
    return kb;
}
static bool vmm_shuffle() {
    static const bool on = !(getenv("BH_ALLOC_VMM_SHUFFLE") && atoi(getenv("BH_ALLOC_VMM_SHUFFLE")) == 0);
    return on;
}
static std::atomic<bool> g_vmm_failed{false};  // the driver refused the virtual-memory API once: hipMalloc from then on
bool dev_alloc_is_shuffled() { return vmm_chunk_kb() > 0 && vmm_shuffle() && !g_vmm_failed.load(); }
bool dev_block_is_vmm(const void* p) {
    std::lock_guard<std::mutex> lk(g_vmm_mu);
    return g_vmm.find(const_cast<void*>(p)) != g_vmm.end();
}

static hipError_t dev_alloc_raw(int device, size_t bytes, void** out);
hipError_t dev_alloc(int device, size_t bytes, void** out) {
    // BH_ALLOC_POISON=1 (debugging): every new block is filled with 0xFF bytes (float NaNs, huge integers), so that a kernel
    // that reads memory nobody wrote shows up deterministically instead of depending on what the pages held before
    static const bool poison = getenv("BH_ALLOC_POISON") && atoi(getenv("BH_ALLOC_POISON")) != 0;
    const hipError_t e = dev_alloc_raw(device, bytes, out);
    if (e == hipSuccess && poison) {
        (void)hipMemset(*out, 0xFF, bytes);
        (void)hipDeviceSynchronize();
    }
    return e;
}
static hipError_t dev_alloc_raw(int device, size_t bytes, void** out) {
    const long chunk_kb = g_vmm_failed.load() ? 0 : vmm_chunk_kb();
    const bool shuffle = vmm_shuffle();
    // 2 GiB: below that the shuffled layout costs instead of paying (R-L x10 on (256,1024,1024): 48.7 against 42.8 ms, on
    // (128,512,512) 6.4 against 5.8; on (256,2048,2048), 4.4-GB buffers, 175-181 against 179; on (512,2048,2048), 8.7 GB, 31.5-32
    // against 34.7 per iteration: tools/time_rl_sizes.py) — 2-MiB fragments shorten the reach of the address translation, and
    // only the giant blocks suffer enough from the contiguous layout to make up for it; a threshold sweep on one box (R-L x10,
    // thresholds 3 GiB / 2000 MB / 1000 MB / hipMalloc): (256,1024,1024) 43.0 / 43.0 / 46.8 / 43.1 ms, the deskewed config-4
    // volume (2.4-GB buffers) 102.6 / 101.2 / 99.2 / 102.7, (256,2048,2048) (4.4 GB) 172.4 / 167.9 / 166.3 / 172.9
    static const size_t min_bytes = (size_t)(getenv("BH_ALLOC_VMM_MIN_MB") ? atol(getenv("BH_ALLOC_VMM_MIN_MB")) : 2048) << 20;
    if (chunk_kb > 0 && bytes >= min_bytes && !va_budget_left()) {
        static std::atomic<bool> said{false};
        if (!said.exchange(true))
            fprintf(stderr, "[bhcore] %llu GiB of address space retained by released blocks (BH_ALLOC_VMM_VA_CAP_GB): gigabyte buffers "
                            "come from hipMalloc from here on\n", g_va_retained_bytes.load() >> 30);
    } else if (chunk_kb > 0 && bytes >= min_bytes) {
        const hipError_t e = vmm_alloc(device, bytes, (size_t)chunk_kb << 10, shuffle, out);
        if (e == hipSuccess || e == hipErrorOutOfMemory) return e;
        fprintf(stderr, "[bhcore] virtual-memory allocation of %zu bytes failed (%s): gigabyte buffers come from hipMalloc from here on\n",
                bytes, hipGetErrorString(e));
        (void)hipGetLastError();  // a driver without the virtual-memory API: hipMalloc from here on
        g_vmm_failed.store(true);
    }
    return hipMalloc(out, bytes);
}
hipError_t dev_free(void* p) {
    if (!p) return hipSuccess;
    VmmBlock blk;
    {
        std::lock_guard<std::mutex> lk(g_vmm_mu);
        auto it = g_vmm.find(p);
        if (it == g_vmm.end()) return hipFree(p);
        blk = std::move(it->second);
        g_vmm.erase(it);
    }
    DeviceGuard guard(blk.device);  // the block's own device, whatever the calling thread has current
    (void)hipDeviceSynchronize();
    hipError_t e = vmm_unmap_all(p, blk, blk.size / blk.chunk);
    for (auto h : blk.handles) {
        const hipError_t er = hipMemRelease(h);
        if (er != hipSuccess && e == hipSuccess) e = er;
    }
    // The address range is NOT given back (BH_ALLOC_VMM_FREE_VA=1 does): a range that is reserved again right after its release
    // and mapped onto other physical chunks was seen to deliver the old pages' contents now and then (tools/alloc_stress.py: a
    // wrong Richardson-Lucy result once in ~100 rebuilds of the workspace, never with this) — address translations of the
    // old mapping outliving it.  Address space is 2^47 bytes; a freed block costs none of it that matters.
    static const bool free_va = getenv("BH_ALLOC_VMM_FREE_VA") && atoi(getenv("BH_ALLOC_VMM_FREE_VA")) != 0;
    if (free_va) {
        const hipError_t ef = hipMemAddressFree(p, blk.size);
        if (ef != hipSuccess && e == hipSuccess) e = ef;
    } else {
        retain_range(blk.size);
    }
    (void)hipDeviceSynchronize();
    return e;
}

int get_scratch(bh_ctx* ctx, const char* name, size_t bytes, void** out) {
    Scratch& s = ctx->scratch[name];
    if (s.bytes < bytes) {
        if (s.ptr) {
            BH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
            BH_CHECK_HIP(dev_free(s.ptr));
            s.ptr = nullptr;
            s.bytes = 0;
        }
        BH_CHECK_HIP(dev_alloc(ctx->device, bytes, &s.ptr));
        s.bytes = bytes;
        if (std::strcmp(name, "fc_spec") == 0) ctx->spec_tuned = nullptr;  // new pages: fftconv_tune_spectrum auditions them
        if (getenv("BH_DEBUG_SCRATCH")) fprintf(stderr, "[bh scratch] %-14s %p  %zu bytes\n", name, s.ptr, bytes);
    }
    *out = s.ptr;
    return BH_OK;
}

// gives a named scratch buffer back to the driver (transients of a one-off set-up: hundreds of milliseconds for gigabytes)
int free_scratch(bh_ctx* ctx, const char* name) {
    auto it = ctx->scratch.find(name);
    if (it == ctx->scratch.end() || !it->second.ptr) return BH_OK;
    BH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    BH_CHECK_HIP(dev_free(it->second.ptr));
    ctx->scratch.erase(it);
    return BH_OK;
}

// ---- plan self-check ---------------------------------------------------------------------------------------------
// rocFFT 1.0.36 (ROCm 7.2) can hand back a 3-D real plan that computes garbage, depending on which other plans the process
// created before (tools/hipfft_two_plans.cpp, tools/hipfft_plan3d_sweep.cpp).  Such a plan is wrong from its
// first execution on and stays wrong; a plan that passes stays right.  So every library plan runs one round trip on
// pseudo-random data when it is created; a 3-D plan that fails is replaced by the decomposed one, which is checked too.
__device__ __forceinline__ float check_value(int64_t i) {
    unsigned h = (unsigned)i * 2654435761u + (unsigned)(i >> 32) * 40503u;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}
__global__ void check_fill_kernel(float* x, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        x[i] = check_value(i);
}
__global__ void check_compare_kernel(const float* x, int64_t n, float inv_n, unsigned* worst) {
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float e = fabsf(x[i] * inv_n - check_value(i));
        m = fmaxf(m, e == e ? e : 1e30f);  // a NaN counts as a failure
    }
    atomicMax(worst, __float_as_uint(m));  // non-negative floats order like their bit patterns
}

static void destroy_handles(FftPlans& p) {
    for (hipfftHandle* h : {&p.r2c, &p.c2r, &p.xr, &p.xi, &p.zy})
        if (*h) {
            hipfftDestroy(*h);
            *h = 0;
        }
    if (p.work) (void)dev_free(p.work);
    p.work = nullptr;
    p.work_bytes = 0;
}

static int make_handles(bh_ctx* ctx, FftPlans& p, int64_t Z, int64_t Y, int64_t X) {
    hipfftHandle* hs[3] = {&p.r2c, &p.c2r, nullptr};
    size_t ws[3] = {0, 0, 0};
    int nh = 2;
    if (p.separable) {
        BH_REQUIRE(Z * Y < (1ll << 31), "FFT batch %lld too large", (long long)(Z * Y));
        hs[0] = &p.xr, hs[1] = &p.xi, hs[2] = &p.zy;
        nh = 3;
    }
    for (int i = 0; i < nh; ++i) {
        BH_CHECK_FFT(hipfftCreate(hs[i]));
        BH_CHECK_FFT(hipfftSetAutoAllocation(*hs[i], 0));
    }
    if (p.separable) {
        const int Xh = (int)(X / 2 + 1);
        int nx[1] = {(int)X}, nzy[2] = {(int)Z, (int)Y};
        BH_CHECK_FFT(hipfftMakePlanMany(p.xr, 1, nx, nullptr, 1, (int)X, nullptr, 1, Xh, HIPFFT_R2C, (int)(Z * Y), &ws[0]));
        BH_CHECK_FFT(hipfftMakePlanMany(p.xi, 1, nx, nullptr, 1, Xh, nullptr, 1, (int)X, HIPFFT_C2R, (int)(Z * Y), &ws[1]));
        BH_CHECK_FFT(hipfftMakePlanMany(p.zy, 2, nzy, nzy, Xh, 1, nzy, Xh, 1, HIPFFT_C2C, Xh, &ws[2]));
    } else {
        BH_CHECK_FFT(hipfftMakePlan3d(p.r2c, (int)Z, (int)Y, (int)X, HIPFFT_R2C, &ws[0]));
        BH_CHECK_FFT(hipfftMakePlan3d(p.c2r, (int)Z, (int)Y, (int)X, HIPFFT_C2R, &ws[1]));
    }
    for (int i = 0; i < nh; ++i)
        if (ws[i] > p.work_bytes) p.work_bytes = ws[i];
    // one work area shared by all plans of the shape (they never run concurrently on one stream)
    if (p.work_bytes) BH_CHECK_HIP(dev_alloc(ctx->device, p.work_bytes, &p.work));
    for (int i = 0; i < nh; ++i) {
        if (p.work_bytes) BH_CHECK_FFT(hipfftSetWorkArea(*hs[i], p.work));
        BH_CHECK_FFT(hipfftSetStream(*hs[i], ctx->stream));
    }
    return BH_OK;
}

// round trip on the context's own scratch (the buffers the caller is about to use at this size anyway)
static int plan_round_trip_error(bh_ctx* ctx, const FftPlans& p, int64_t Z, int64_t Y, int64_t X, float* err) {
    const int64_t V = Z * Y * X, NS = Z * Y * (X / 2 + 1);
    float* real;
    float2* spec;
    unsigned* worst;
    BH_TRY(get_scratch(ctx, "fft_real", V * sizeof(float), (void**)&real));
    BH_TRY(get_scratch(ctx, "fft_spec", NS * sizeof(float2), (void**)&spec));
    BH_TRY(get_scratch(ctx, "plan_check", 64, (void**)&worst));
    hipStream_t s = ctx->stream;
    const int grid = (int)std::min<int64_t>(ceil_div(V, 256), (int64_t)ctx->num_cus * 16);
    BH_CHECK_HIP(hipMemsetAsync(worst, 0, sizeof(unsigned), s));
    hipLaunchKernelGGL(check_fill_kernel, dim3(grid), dim3(256), 0, s, real, V);
    BH_TRY(fft_forward(&p, real, spec));
    BH_TRY(fft_inverse(&p, spec, real));
    hipLaunchKernelGGL(check_compare_kernel, dim3(grid), dim3(256), 0, s, (const float*)real, V, (float)(1.0 / (double)V), worst);
    BH_CHECK_HIP(hipGetLastError());
    unsigned bits = 0;
    BH_CHECK_HIP(hipMemcpyAsync(&bits, worst, sizeof(bits), hipMemcpyDeviceToHost, s));
    BH_CHECK_HIP(hipStreamSynchronize(s));
    memcpy(err, &bits, sizeof(float));
    return BH_OK;
}

int get_plans(bh_ctx* ctx, int64_t Z, int64_t Y, int64_t X, FftPlans** out) {
    auto key = std::make_tuple(Z, Y, X);
    auto it = ctx->plans.find(key);
    if (it != ctx->plans.end()) {
        *out = &it->second;
        return BH_OK;
    }
    BH_REQUIRE(Z > 0 && Y > 0 && X > 0 && Z < (1ll << 31) && Y < (1ll << 31) && X < (1ll << 31),
               "invalid FFT shape (%lld,%lld,%lld)", (long long)Z, (long long)Y, (long long)X);
    FftPlans p;
    auto pow2 = [](int64_t n) { return (n & (n - 1)) == 0; };
    p.separable = getenv("BH_FFT_SEPARABLE") != nullptr || (pow2(Z) && pow2(Y) && pow2(X));
    const bool check = getenv("BH_FFT_NOCHECK") == nullptr;
    const float tol = 1e-3f;  // a sound float32 round trip of values in [0, 1) errs by ~1e-6; a bad plan by ~1
    float err = 0.0f;
    BH_TRY(make_handles(ctx, p, Z, Y, X));
    if (check) {
        BH_TRY(plan_round_trip_error(ctx, p, Z, Y, X, &err));
        if (!(err < tol) && !p.separable) {
            destroy_handles(p);
            p.separable = true;
            BH_TRY(make_handles(ctx, p, Z, Y, X));
            BH_TRY(plan_round_trip_error(ctx, p, Z, Y, X, &err));
            ++ctx->plans_replaced;
        }
        if (!(err < tol)) {
            destroy_handles(p);
            set_error("hipFFT plan for (%lld,%lld,%lld) fails its round-trip self-check (error %g): the FFT library is "
                      "returning wrong transforms", (long long)Z, (long long)Y, (long long)X, (double)err);
            return BH_ERR_HIP;
        }
    }
    auto ins = ctx->plans.emplace(key, p);
    *out = &ins.first->second;
    return BH_OK;
}

int fft_forward(const FftPlans* pl, const float* real, float2* spec) {
    if (!pl->separable) {
        BH_CHECK_FFT(hipfftExecR2C(pl->r2c, const_cast<float*>(real), (hipfftComplex*)spec));
        return BH_OK;
    }
    BH_CHECK_FFT(hipfftExecR2C(pl->xr, const_cast<float*>(real), (hipfftComplex*)spec));
    BH_CHECK_FFT(hipfftExecC2C(pl->zy, (hipfftComplex*)spec, (hipfftComplex*)spec, HIPFFT_FORWARD));
    return BH_OK;
}

int fft_inverse(const FftPlans* pl, float2* spec, float* real) {
    if (!pl->separable) {
        BH_CHECK_FFT(hipfftExecC2R(pl->c2r, (hipfftComplex*)spec, real));
        return BH_OK;
    }
    BH_CHECK_FFT(hipfftExecC2C(pl->zy, (hipfftComplex*)spec, (hipfftComplex*)spec, HIPFFT_BACKWARD));
    BH_CHECK_FFT(hipfftExecC2R(pl->xi, (hipfftComplex*)spec, real));
    return BH_OK;
}

}  // namespace bh

extern "C" {

int bh_abi_version(void) { return BH_ABI_VERSION; }

// How gigabyte buffers are laid out right now: chunk size in KiB (0 = hipMalloc), chunks shuffled or not, and how many
// blocks / bytes are currently built that way (diagnostics: bench.py prints it).
int bh_alloc_layout(int* chunk_kib, int* shuffled, uint64_t* live_blocks, uint64_t* live_bytes) {
    if (chunk_kib) *chunk_kib = bh::g_vmm_failed.load() ? 0 : (int)bh::vmm_chunk_kb();
    if (shuffled) *shuffled = bh::dev_alloc_is_shuffled() ? 1 : 0;
    std::lock_guard<std::mutex> lk(bh::g_vmm_mu);
    if (live_blocks) *live_blocks = bh::g_vmm.size();
    if (live_bytes) {
        uint64_t n = 0;
        for (auto& kv : bh::g_vmm) n += kv.second.size;
        *live_bytes = n;
    }
    return BH_OK;
}

int bh_alloc_retained(uint64_t* ranges, uint64_t* bytes) {
    if (ranges) *ranges = bh::g_va_retained_ranges.load();
    if (bytes) *bytes = bh::g_va_retained_bytes.load();
    return BH_OK;
}

// torch.cuda.memory.CUDAPluggableAllocator entry points: the allocator of a torch MemPool whose large blocks are laid out like
// the library's own workspace (biahub_amd/device.py: volume_pool)
void* bh_torch_alloc(size_t size, int device, void* stream) {
    (void)stream;
    void* p = nullptr;
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (cur != device) (void)hipSetDevice(device);
    const hipError_t e = bh::dev_alloc(device, size ? size : 1, &p);
    if (cur != device) (void)hipSetDevice(cur);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}
void bh_torch_free(void* ptr, size_t size, int device, void* stream) {
    (void)size;
    (void)stream;
    bh::DeviceGuard guard(device);  // hipFree / the unmap synchronise the block's device, not whatever is current
    (void)bh::dev_free(ptr);
}

const char* bh_last_error(void) { return bh::g_err; }

int bh_device_count(int* count) {
    BH_REQUIRE(count != nullptr, "count is NULL");
    BH_CHECK_HIP(hipGetDeviceCount(count));
    return BH_OK;
}

int bh_ctx_create(int device, void* hip_stream, bh_ctx** out) {
    BH_REQUIRE(out != nullptr, "out is NULL");
    int n = 0;
    BH_CHECK_HIP(hipGetDeviceCount(&n));
    BH_REQUIRE(device >= 0 && device < n, "device %d out of range (have %d)", device, n);
    BH_CHECK_HIP(hipSetDevice(device));
    bh_ctx* c = new bh_ctx();
    c->device = device;
    c->stream = (hipStream_t)hip_stream;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cus = prop.multiProcessorCount;
    for (int i = 0; i < 2 * bh::T_COUNT; ++i) BH_CHECK_HIP(hipEventCreate(&c->ev[i]));
    *out = c;
    return BH_OK;
}

int bh_ctx_release_workspace(bh_ctx* ctx) {
    BH_REQUIRE(ctx != nullptr, "ctx is NULL");
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    BH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    for (auto& kv : ctx->plans) {
        for (hipfftHandle h : {kv.second.r2c, kv.second.c2r, kv.second.xr, kv.second.xi, kv.second.zy})
            if (h) hipfftDestroy(h);
        if (kv.second.work) (void)bh::dev_free(kv.second.work);
    }
    ctx->plans.clear();
    for (auto& kv : ctx->scratch)
        if (kv.second.ptr) (void)bh::dev_free(kv.second.ptr);
    ctx->scratch.clear();
    ctx->otf_valid = false;
    ctx->spec_tuned = nullptr;
    return BH_OK;
}

int bh_ctx_workspace_bytes(bh_ctx* ctx, uint64_t* bytes) {
    BH_REQUIRE(ctx != nullptr && bytes != nullptr, "NULL argument");
    uint64_t b = 0;
    for (auto& kv : ctx->plans) b += kv.second.work_bytes;
    for (auto& kv : ctx->scratch) b += kv.second.bytes;
    *bytes = b;
    return BH_OK;
}

int bh_ctx_fft_plans_replaced(bh_ctx* ctx, int* count) {
    BH_REQUIRE(ctx != nullptr && count != nullptr, "NULL argument");
    *count = ctx->plans_replaced;
    return BH_OK;
}

int bh_ctx_destroy(bh_ctx* ctx) {
    if (!ctx) return BH_OK;
    bh_ctx_release_workspace(ctx);
    for (int i = 0; i < 2 * bh::T_COUNT; ++i)
        if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    delete ctx;
    return BH_OK;
}

int bh_ctx_set_stream(bh_ctx* ctx, void* hip_stream) {
    BH_REQUIRE(ctx != nullptr, "ctx is NULL");
    ctx->stream = (hipStream_t)hip_stream;
    for (auto& kv : ctx->plans) {
        for (hipfftHandle h : {kv.second.r2c, kv.second.c2r, kv.second.xr, kv.second.xi, kv.second.zy})
            if (h) BH_CHECK_FFT(hipfftSetStream(h, ctx->stream));
    }
    return BH_OK;
}

int bh_ctx_synchronize(bh_ctx* ctx) {
    BH_REQUIRE(ctx != nullptr, "ctx is NULL");
    BH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return BH_OK;
}

int bh_ctx_set_timing(bh_ctx* ctx, int enabled) {
    BH_REQUIRE(ctx != nullptr, "ctx is NULL");
    ctx->timing = enabled != 0;
    return BH_OK;
}

int bh_last_elapsed_ms(bh_ctx* ctx, int what, float* ms) {
    BH_REQUIRE(ctx != nullptr && ms != nullptr, "NULL argument");
    BH_REQUIRE(what >= 0 && what < bh::T_COUNT, "unknown timer slot %d", what);
    if (what == bh::T_RL_ITER) {
        *ms = ctx->ms_override[what];
        return BH_OK;
    }
    BH_REQUIRE(ctx->ev_valid[what], "no timed call recorded for slot %d (enable bh_ctx_set_timing)", what);
    BH_CHECK_HIP(hipEventSynchronize(ctx->ev[2 * what + 1]));
    BH_CHECK_HIP(hipEventElapsedTime(ms, ctx->ev[2 * what], ctx->ev[2 * what + 1]));
    return BH_OK;
}

int bh_malloc(void** dptr, uint64_t bytes) {
    BH_REQUIRE(dptr != nullptr, "dptr is NULL");
    int device = 0;
    BH_CHECK_HIP(hipGetDevice(&device));  // torch-less hosts: their volumes get the library's page layout too
    BH_CHECK_HIP(bh::dev_alloc(device, bytes ? bytes : 1, dptr));
    return BH_OK;
}

int bh_free(void* dptr) {
    if (dptr) BH_CHECK_HIP(bh::dev_free(dptr));
    return BH_OK;
}

int bh_memcpy_h2d(bh_ctx* ctx, void* dst, const void* src, uint64_t bytes) {
    BH_REQUIRE(ctx != nullptr, "ctx is NULL");
    BH_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    BH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return BH_OK;
}

int bh_memcpy_d2h(bh_ctx* ctx, void* dst, const void* src, uint64_t bytes) {
    BH_REQUIRE(ctx != nullptr, "ctx is NULL");
    BH_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    BH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return BH_OK;
}

}  // extern "C"
