// Byte permutations of the Blosc-1 chunk container on gfx950: byte shuffle and bit shuffle, both directions.
//   c-blosc 1.21 shuffle.c / shuffle-generic.h / bitshuffle-generic.c (the library numcodecs wraps; reference
//   uv.lock:3160-3161), restated in biahub_amd/codecs.py (shuffle, unshuffle, bitshuffle, bitunshuffle, unfilter).
// A chunk of an iohub-written store is `nblocks` independently permuted blocks of `blocksize` bytes (the last one
// shorter).  Inside a block of n = bytes / typesize elements:
//   byte shuffle: byte j of element i sits at j n + i; the bytes % typesize tail is unpermuted;
//   bit shuffle : bit k of byte j of element i sits at bit i % 8 of byte (8 j + k) n / 8 + i / 8 — only when n % 8 == 0,
//                 otherwise c-blosc 1.x stores the block unpermuted.
// The entropy decoder (zstd) runs on host cores; its output is uploaded still permuted and un-permuted here, so the
// strided byte gather happens at HBM rate instead of on a core.  Pure byte movement: bit-exact by construction, checked
// against the NumPy restatement and against streams of the real library (tests/test_gpu_parity.py).
//
// Work item = G consecutive elements of one block; a 256-thread workgroup takes 256 consecutive items, so every plane is
// read (un-filter) or written (filter) in contiguous runs of 256 * G / 8 (bit) or 256 * G (byte) bytes.
#include "common.hpp"

#include <vector>

namespace bh {

struct FilterParams {
    uint64_t nbytes;
    uint32_t blocksize, typesize;
    uint32_t items_per_block;  // ceil(ceil(blocksize / typesize) / G)
    uint64_t items;            // nblocks * items_per_block
};

// 8x8 bit-matrix transpose of a 64-bit word seen as 8 rows (bytes, row r = byte r) of 8 columns (bit c = column c):
// out row c, column r = in row r, column c.
__device__ __forceinline__ uint64_t transpose8x8(uint64_t x) {
    uint64_t t;
    t = (x ^ (x >> 7)) & 0x00AA00AA00AA00AAull;
    x = x ^ t ^ (t << 7);
    t = (x ^ (x >> 14)) & 0x0000CCCC0000CCCCull;
    x = x ^ t ^ (t << 14);
    t = (x ^ (x >> 28)) & 0x00000000F0F0F0F0ull;
    x = x ^ t ^ (t << 28);
    return x;
}

template <typename V>
__device__ __forceinline__ V load_unaligned(const uint8_t* p) {
    V v;
    __builtin_memcpy(&v, p, sizeof(V));
    return v;
}
template <typename V>
__device__ __forceinline__ void store_unaligned(uint8_t* p, V v) {
    __builtin_memcpy(p, &v, sizeof(V));
}

// ---- byte shuffle ----------------------------------------------------------------------------------------------
// G = 16 elements per item: 16 bytes per plane, 16 TS bytes of elements.
template <int TS, bool FORWARD>
__global__ __launch_bounds__(256) void byte_shuffle_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, FilterParams p) {
    constexpr int G = 16;
    const uint32_t ts = TS > 0 ? TS : p.typesize;
    for (uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x; w < p.items; w += (uint64_t)gridDim.x * 256) {
        const uint64_t b = w / p.items_per_block;
        const uint32_t g = (uint32_t)(w - b * p.items_per_block);
        const uint64_t o0 = b * p.blocksize;
        const uint32_t bs = (uint32_t)min((uint64_t)p.blocksize, p.nbytes - o0);
        const uint32_t n = bs / ts;
        const uint32_t i0 = g * G;
        const uint8_t* s = src + o0;
        uint8_t* d = dst + o0;
        if (g == 0)
            for (uint32_t k = n * ts; k < bs; ++k) d[k] = s[k];  // tail shorter than one element
        if (i0 >= n) continue;
        if (TS > 0 && i0 + G <= n) {
            uint8_t e[G * (TS > 0 ? TS : 1)];
            if (FORWARD) {
#pragma unroll
                for (int q = 0; q < TS; ++q) store_unaligned<uint4>(e + 16 * q, load_unaligned<uint4>(s + (size_t)i0 * TS + 16 * q));
#pragma unroll
                for (int j = 0; j < TS; ++j) {
                    uint8_t pl[G];
#pragma unroll
                    for (int i = 0; i < G; ++i) pl[i] = e[i * TS + j];
                    store_unaligned<uint4>(d + (size_t)j * n + i0, load_unaligned<uint4>(pl));
                }
            } else {
#pragma unroll
                for (int j = 0; j < TS; ++j) {
                    uint8_t pl[G];
                    store_unaligned<uint4>(pl, load_unaligned<uint4>(s + (size_t)j * n + i0));
#pragma unroll
                    for (int i = 0; i < G; ++i) e[i * TS + j] = pl[i];
                }
#pragma unroll
                for (int q = 0; q < TS; ++q) store_unaligned<uint4>(d + (size_t)i0 * TS + 16 * q, load_unaligned<uint4>(e + 16 * q));
            }
        } else {  // ragged end of a block, or a type size without a specialisation
            const uint32_t i1 = min(n, i0 + G);
            for (uint32_t i = i0; i < i1; ++i)
                for (uint32_t j = 0; j < ts; ++j) {
                    if (FORWARD) d[(size_t)j * n + i] = s[(size_t)i * ts + j];
                    else d[(size_t)i * ts + j] = s[(size_t)j * n + i];
                }
        }
    }
}

// ---- bit shuffle -----------------------------------------------------------------------------------------------
// G = 32 elements per item: 4 bytes of each of the 8 TS bit planes, 32 TS bytes of elements.
template <int TS, bool FORWARD>
__global__ __launch_bounds__(256) void bit_shuffle_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, FilterParams p) {
    constexpr int G = 32;
    const uint32_t ts = TS > 0 ? TS : p.typesize;
    for (uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x; w < p.items; w += (uint64_t)gridDim.x * 256) {
        const uint64_t b = w / p.items_per_block;
        const uint32_t g = (uint32_t)(w - b * p.items_per_block);
        const uint64_t o0 = b * p.blocksize;
        const uint32_t bs = (uint32_t)min((uint64_t)p.blocksize, p.nbytes - o0);
        const uint32_t n = bs / ts;
        const uint8_t* s = src + o0;
        uint8_t* d = dst + o0;
        const uint32_t i0 = g * G;
        if (n % 8 != 0 || bs < ts) {  // c-blosc 1.x leaves such a block unpermuted: plain copy, G ts bytes per item
            const uint64_t k0 = (uint64_t)i0 * ts, k1 = min((uint64_t)bs, k0 + (uint64_t)G * ts);
            for (uint64_t k = k0; k < k1; ++k) d[k] = s[k];
            continue;
        }
        if (g == 0)
            for (uint32_t k = n * ts; k < bs; ++k) d[k] = s[k];
        if (i0 >= n) continue;
        const uint32_t pl = n / 8;  // bytes per bit plane
        const uint32_t c0 = i0 / 8; // first plane byte of this item
        if (TS > 0 && i0 + G <= n) {
            // r[j][q]: the 8x8 bit matrix of byte j of elements 8 q .. 8 q + 7 (row = element, column = bit), q < 4
            uint64_t m[(TS > 0 ? TS : 1)][4];
            if (FORWARD) {
                uint8_t e[G * (TS > 0 ? TS : 1)];
#pragma unroll
                for (int q = 0; q < 2 * TS; ++q) store_unaligned<uint4>(e + 16 * q, load_unaligned<uint4>(s + (size_t)i0 * TS + 16 * q));
#pragma unroll
                for (int j = 0; j < TS; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        uint64_t x = 0;
#pragma unroll
                        for (int r = 0; r < 8; ++r) x |= (uint64_t)e[(8 * q + r) * TS + j] << (8 * r);
                        m[j][q] = transpose8x8(x);  // row k = bit plane k, column = element
                    }
#pragma unroll
                for (int j = 0; j < TS; ++j)
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        uint32_t v = 0;
#pragma unroll
                        for (int q = 0; q < 4; ++q) v |= (uint32_t)((m[j][q] >> (8 * k)) & 0xFF) << (8 * q);
                        store_unaligned<uint32_t>(d + (size_t)(8 * j + k) * pl + c0, v);
                    }
            } else {
#pragma unroll
                for (int j = 0; j < TS; ++j) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) m[j][q] = 0;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const uint32_t v = load_unaligned<uint32_t>(s + (size_t)(8 * j + k) * pl + c0);
#pragma unroll
                        for (int q = 0; q < 4; ++q) m[j][q] |= (uint64_t)((v >> (8 * q)) & 0xFF) << (8 * k);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) m[j][q] = transpose8x8(m[j][q]);  // row = element, column = bit
                }
                uint8_t e[G * (TS > 0 ? TS : 1)];
#pragma unroll
                for (int j = 0; j < TS; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int r = 0; r < 8; ++r) e[(8 * q + r) * TS + j] = (uint8_t)(m[j][q] >> (8 * r));
#pragma unroll
                for (int q = 0; q < 2 * TS; ++q) store_unaligned<uint4>(d + (size_t)i0 * TS + 16 * q, load_unaligned<uint4>(e + 16 * q));
            }
        } else {  // groups of 8 elements, one byte per plane
            const uint32_t i1 = min(n, i0 + G);
            for (uint32_t i = i0; i < i1; i += 8)
                for (uint32_t j = 0; j < ts; ++j) {
                    uint64_t x = 0;
                    if (FORWARD) {
                        for (int r = 0; r < 8; ++r) x |= (uint64_t)s[(size_t)(i + r) * ts + j] << (8 * r);
                        x = transpose8x8(x);
                        for (int k = 0; k < 8; ++k) d[(size_t)(8 * j + k) * pl + i / 8] = (uint8_t)(x >> (8 * k));
                    } else {
                        for (int k = 0; k < 8; ++k) x |= (uint64_t)s[(size_t)(8 * j + k) * pl + i / 8] << (8 * k);
                        x = transpose8x8(x);
                        for (int r = 0; r < 8; ++r) d[(size_t)(i + r) * ts + j] = (uint8_t)(x >> (8 * r));
                    }
                }
        }
    }
}

template <bool FORWARD>
static int run_filter(bh_ctx* ctx, const void* src, void* dst, uint64_t nbytes, uint32_t blocksize, uint32_t typesize, int mode) {
    BH_REQUIRE(ctx && (nbytes == 0 || (src && dst)), "NULL argument");
    BH_REQUIRE(mode >= 0 && mode <= 2, "unknown blosc shuffle mode %d", mode);
    BH_REQUIRE(typesize >= 1 && typesize <= 255, "typesize %u outside 1..255", typesize);
    BH_REQUIRE(nbytes == 0 || blocksize >= 1, "blocksize must be positive");
    BH_REQUIRE(src != dst || nbytes == 0, "in-place (un)filtering is not supported");
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    if (nbytes == 0) return BH_OK;
    if (mode == 0 || (mode == 1 && typesize == 1)) {
        BH_CHECK_HIP(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToDevice, ctx->stream));
        return BH_OK;
    }
    FilterParams p;
    p.nbytes = nbytes;
    p.blocksize = (uint32_t)std::min<uint64_t>(blocksize, nbytes);
    p.typesize = typesize;
    const uint32_t G = mode == 1 ? 16 : 32;
    // items cover ceil(blocksize / typesize) elements so that an unpermuted (copied) block is covered to its last byte
    p.items_per_block = (uint32_t)ceil_div(ceil_div((int64_t)p.blocksize, (int64_t)typesize), (int64_t)G);
    p.items = (uint64_t)ceil_div((int64_t)nbytes, (int64_t)p.blocksize) * p.items_per_block;
    const dim3 grid((unsigned)std::min<uint64_t>((uint64_t)ceil_div((int64_t)p.items, 256), (uint64_t)ctx->num_cus * 32));
    const uint8_t* s = (const uint8_t*)src;
    uint8_t* d = (uint8_t*)dst;
#define BH_LAUNCH(KERNEL, TS) hipLaunchKernelGGL((KERNEL<TS, FORWARD>), grid, dim3(256), 0, ctx->stream, s, d, p)
    if (mode == 1) {
        switch (typesize) {
            case 2: BH_LAUNCH(byte_shuffle_kernel, 2); break;
            case 4: BH_LAUNCH(byte_shuffle_kernel, 4); break;
            case 8: BH_LAUNCH(byte_shuffle_kernel, 8); break;
            default: BH_LAUNCH(byte_shuffle_kernel, 0); break;
        }
    } else {
        switch (typesize) {
            case 1: BH_LAUNCH(bit_shuffle_kernel, 1); break;
            case 2: BH_LAUNCH(bit_shuffle_kernel, 2); break;
            case 4: BH_LAUNCH(bit_shuffle_kernel, 4); break;
            default: BH_LAUNCH(bit_shuffle_kernel, 0); break;
        }
    }
#undef BH_LAUNCH
    BH_CHECK_HIP(hipGetLastError());
    return BH_OK;
}

// ---- the same permutations on a host core (volumes that stay on the host; no GPU or context involved) -----------
static inline uint64_t transpose8x8_host(uint64_t x) {
    uint64_t t;
    t = (x ^ (x >> 7)) & 0x00AA00AA00AA00AAull;
    x = x ^ t ^ (t << 7);
    t = (x ^ (x >> 14)) & 0x0000CCCC0000CCCCull;
    x = x ^ t ^ (t << 14);
    t = (x ^ (x >> 28)) & 0x00000000F0F0F0F0ull;
    x = x ^ t ^ (t << 28);
    return x;
}

// byte plane j of n elements of TS bytes (a strided read, a contiguous write)
template <int TS>
static inline void byte_plane_host(const uint8_t* __restrict__ s, uint8_t* __restrict__ plane, size_t n, size_t tsr, size_t j) {
    const size_t ts = TS > 0 ? (size_t)TS : tsr;
    for (size_t i = 0; i < n; ++i) plane[i] = s[i * ts + j];
}

// one block; TS > 0: compile-time element size (lets the compiler interleave / vectorise), TS == 0: run-time `tsr`
template <bool FORWARD, int TS>
static void filter_block_host(const uint8_t* __restrict__ s, uint8_t* __restrict__ d, size_t bs, size_t tsr, int mode) {
    const size_t ts = TS > 0 ? (size_t)TS : tsr;
    const size_t n = bs / ts;
    if (mode == 0 || (mode == 1 && ts == 1) || (mode == 2 && (n % 8 != 0 || bs < ts))) {
        __builtin_memcpy(d, s, bs);
        return;
    }
    __builtin_memcpy(d + n * ts, s + n * ts, bs - n * ts);
    if (mode == 1) {
        if (FORWARD) {
            for (size_t j = 0; j < ts; ++j) byte_plane_host<TS>(s, d + j * n, n, ts, j);
        } else {
            for (size_t i = 0; i < n; ++i)
                for (size_t j = 0; j < ts; ++j) d[i * ts + j] = s[j * n + i];
        }
        return;
    }
    const size_t pl = n / 8;
    if (FORWARD) {  // byte planes first (contiguous 8-byte reads afterwards), then the 8x8 bit transposes
        static thread_local std::vector<uint8_t> tmp;
        if (tmp.size() < n * ts) tmp.resize(n * ts);
        uint8_t* __restrict__ t = tmp.data();
        for (size_t j = 0; j < ts; ++j) byte_plane_host<TS>(s, t + j * n, n, ts, j);
        for (size_t j = 0; j < ts; ++j)
            for (size_t c = 0; c < pl; ++c) {
                uint64_t x;
                __builtin_memcpy(&x, t + j * n + 8 * c, 8);
                x = transpose8x8_host(x);
                for (int k = 0; k < 8; ++k) d[(8 * j + k) * pl + c] = (uint8_t)(x >> (8 * k));
            }
        return;
    }
    for (size_t c = 0; c < pl; ++c)
        for (size_t j = 0; j < ts; ++j) {
            uint64_t x = 0;
            if (FORWARD) {
            } else {
                for (int k = 0; k < 8; ++k) x |= (uint64_t)s[(8 * j + k) * pl + c] << (8 * k);
                x = transpose8x8_host(x);
                for (int r = 0; r < 8; ++r) d[(8 * c + r) * ts + j] = (uint8_t)(x >> (8 * r));
            }
        }
}

template <bool FORWARD>
static int run_filter_host(const uint8_t* src, uint8_t* dst, uint64_t nbytes, uint32_t blocksize, uint32_t typesize, int mode) {
    BH_REQUIRE(nbytes == 0 || (src && dst), "NULL argument");
    BH_REQUIRE(mode >= 0 && mode <= 2, "unknown blosc shuffle mode %d", mode);
    BH_REQUIRE(typesize >= 1 && typesize <= 255, "typesize %u outside 1..255", typesize);
    BH_REQUIRE(nbytes == 0 || blocksize >= 1, "blocksize must be positive");
    BH_REQUIRE(src != dst || nbytes == 0, "in-place (un)filtering is not supported");
    for (uint64_t o0 = 0; o0 < nbytes; o0 += blocksize) {
        const size_t bs = (size_t)std::min<uint64_t>(blocksize, nbytes - o0);
        switch (typesize) {
            case 1: filter_block_host<FORWARD, 1>(src + o0, dst + o0, bs, 1, mode); break;
            case 2: filter_block_host<FORWARD, 2>(src + o0, dst + o0, bs, 2, mode); break;
            case 4: filter_block_host<FORWARD, 4>(src + o0, dst + o0, bs, 4, mode); break;
            case 8: filter_block_host<FORWARD, 8>(src + o0, dst + o0, bs, 8, mode); break;
            default: filter_block_host<FORWARD, 0>(src + o0, dst + o0, bs, typesize, mode); break;
        }
    }
    return BH_OK;
}

}  // namespace bh

extern "C" {

int bh_host_blosc_unfilter(const void* src, void* dst, uint64_t nbytes, uint32_t blocksize, uint32_t typesize, int mode) {
    return bh::run_filter_host<false>((const uint8_t*)src, (uint8_t*)dst, nbytes, blocksize, typesize, mode);
}

int bh_host_blosc_filter(const void* src, void* dst, uint64_t nbytes, uint32_t blocksize, uint32_t typesize, int mode) {
    return bh::run_filter_host<true>((const uint8_t*)src, (uint8_t*)dst, nbytes, blocksize, typesize, mode);
}

int bh_blosc_unfilter(bh_ctx* ctx, const void* src, void* dst, uint64_t nbytes, uint32_t blocksize, uint32_t typesize, int mode) {
    return bh::run_filter<false>(ctx, src, dst, nbytes, blocksize, typesize, mode);
}

int bh_blosc_filter(bh_ctx* ctx, const void* src, void* dst, uint64_t nbytes, uint32_t blocksize, uint32_t typesize, int mode) {
    return bh::run_filter<true>(ctx, src, dst, nbytes, blocksize, typesize, mode);
}

}  // extern "C"
