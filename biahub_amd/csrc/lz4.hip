// LZ4 block compression / decompression of Blosc-1 blocks on gfx950, and the assembly of Blosc frames on the device.
//
// The stores either side of the path are iohub's: zarr chunks holding Blosc-1 frames (numcodecs.Blosc; reference
// biahub/deskew.py:608-640, settings.py:43,460).  Round 1-3 ran the byte permutations of the container on the GPU
// (csrc/codec.hip) and left the entropy coder to host threads, so a compressed store moved every RAW byte across PCIe and cost
// ~0.55 s of zstd per 2-GB volume.  Here the block codec itself runs on the device — LZ4, which the Blosc container carries as
// one of its inner codecs (format 1; numcodecs / c-blosc read it like any other) — so only compressed bytes cross PCIe:
//   compress:  one WAVEFRONT per Blosc block (256 KiB of already permuted bytes).  Per step the 64 lanes hash the four bytes at
//              64 consecutive positions into a wave-private LDS table (4096 x 4 B), test the candidates they find, the earliest
//              match is extended 256 bytes per step by dword compares and ballots, literals and match go out cooperatively.
//              A greedy matcher without lazy evaluation or back-extension: bit-shuffled float / uint16 volumes are long runs in
//              the high bit planes and noise in the low ones — the runs are what there is to win.
//   frames:    per chunk, an exclusive scan of the block sizes gives the block start table; one kernel writes header, table and
//              payloads of every frame into one packed buffer (frames at 16-byte aligned offsets), so ONE download carries it.
//   decompress: one wavefront per block walks the token stream (lane 0's bytes broadcast), literals and matches are copied 64
//              lanes wide (a match that overlaps its own output is periodic in its offset: dst[i] = src[i mod offset]).
// The format is the LZ4 block format (token, literal length extension, literals, 2-byte offset, match length extension; last
// five bytes literals, last match starts >= 12 bytes before the end), so c-blosc decodes what this writes and this decodes
// what c-blosc writes (tests/test_codecs.py against the real library's streams).
#include "common.hpp"

namespace bh {

namespace lz4 {

constexpr int HASH_LOG = 12, HASH_SIZE = 1 << HASH_LOG;
// A table entry is the low 16 bits of a position: a match reaches at most 65535 bytes back, so the candidate is the position
// with those low bits in the 64 KiB behind the current one (an entry never written reads as "exactly 65536 back": rejected).
// 8 KiB per table: 8 wavefronts per workgroup (the 64-KiB static limit), 16 per CU — the matcher is a chain of dependent
// accesses (table, candidate, compare), and with 16-KiB tables the CU held 8 wavefronts to hide them behind.
constexpr int WAVES = 8;            // wavefronts per workgroup, one block each
constexpr int MFLIMIT = 12, LASTLITERALS = 5, MINMATCH = 4;

__device__ __forceinline__ unsigned ld32(const uint8_t* p) {
    unsigned v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ void st32(uint8_t* p, unsigned v) { __builtin_memcpy(p, &v, 4); }

// copy n bytes (any alignment, non-overlapping), the wavefront together: 256 bytes per step as dwords, the tail by bytes
__device__ __forceinline__ void wave_copy(uint8_t* dst, const uint8_t* src, unsigned n, int lane) {
    unsigned i = 0;
    for (; i + 256 <= n; i += 256) st32(dst + i + 4 * lane, ld32(src + i + 4 * lane));
    for (unsigned j = i + lane; j < n; j += 64) dst[j] = src[j];
}

// a length >= 15 continues behind its nibble as bytes that add up to (len - 15): 255, 255, ..., rest
__device__ __forceinline__ unsigned ext_bytes(unsigned len) { return len >= 15 ? (len - 15) / 255 + 1 : 0; }
__device__ __forceinline__ void put_ext(uint8_t* dst, unsigned len, int lane) {  // len >= 15
    const unsigned r = len - 15, n255 = r / 255;
    for (unsigned j = lane; j < n255; j += 64) dst[j] = 255;
    if (lane == 0) dst[n255] = (uint8_t)(r - 255 * n255);
}

// sizes[b] = compressed bytes of block b written at dst + b * slot (== the block's length when it does not shrink: the block
// is then copied raw, which is how the Blosc container marks a stored block)
// block b = frame b / nb, block b % nb of that frame: cbytes bytes per frame, the last block of a frame may be short
__global__ __launch_bounds__(64 * WAVES) void compress_kernel(const uint8_t* __restrict__ src, uint32_t cbytes, uint32_t nb, uint32_t blocksize,
                                                             uint8_t* __restrict__ dst, uint64_t slot, uint32_t* __restrict__ sizes,
                                                             uint32_t nblocks) {
    __shared__ unsigned short table_all[WAVES][HASH_SIZE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned short* table = table_all[wave];
    for (uint32_t b = blockIdx.x * WAVES + wave; b < nblocks; b += gridDim.x * WAVES) {
        const uint32_t fr = b / nb, jb = b - fr * nb;
        const uint8_t* in = src + (uint64_t)fr * cbytes + (uint64_t)jb * blocksize;
        const unsigned n = min(blocksize, cbytes - jb * blocksize);  // (jb * blocksize < cbytes < 2^32)
        uint8_t* out = dst + (uint64_t)b * slot;
        // (no reset: a stale entry of the previous block is a candidate like any other — it is verified against the bytes)
        unsigned ip = 0, anchor = 0, op = 0;
        bool fits = true;
        const unsigned mflimit = n > MFLIMIT ? n - MFLIMIT : 0;   // a match starts before this position
        const unsigned matchlimit = n > LASTLITERALS ? n - LASTLITERALS : 0;  // and ends at or before this one
        // the next window's dwords are requested a step ahead (they do not depend on the matching; after a match that ends
        // beyond the window they are loaded again)
        unsigned vnext = ip + 64 <= mflimit ? ld32(in + ip + lane) : 0u, pnext = ip;
        while (fits && ip + 64 <= mflimit) {
            const unsigned p = ip + lane;
            const unsigned v = pnext == ip ? vnext : ld32(in + p);
            pnext = ip + 64;
            vnext = pnext + 64 <= mflimit ? ld32(in + pnext + lane) : 0u;
            const unsigned h = (v * 2654435761u) >> (32 - HASH_LOG);
            const unsigned e = table[h];
            table[h] = (unsigned short)p;  // reads of the step come before its writes (one wavefront's LDS operations execute in order)
            // the position with these low 16 bits in (p - 65536, p]: distance 0 (the lane finds ITSELF: after a short match the
            // window starts again inside positions the table already holds) and distances beyond the start of the block fail
            const unsigned dist = (p - e) & 0xffffu;
            const unsigned cand = p - dist;
            bool ok = dist != 0u && dist <= p;
            ok = ok && ld32(in + (ok ? cand : 0)) == v;
            // every match of the window is taken in turn: the ballot of this step stays valid for the lanes behind a match's end
            unsigned long long m = __ballot(ok);
            const unsigned wend = ip + 64;
            unsigned next_ip = wend;
            while (m != 0ull) {
                const int first = __builtin_ctzll(m);
                const unsigned mp = ip + first;
                const unsigned mc = (unsigned)__builtin_amdgcn_readlane((int)cand, first);
                // extend: dwords 1.. of the match, 256 bytes per step
                unsigned len = MINMATCH;
                for (;;) {
                    const unsigned q = mp + len + 4 * lane;
                    const bool in_range = q + 4 <= matchlimit;
                    const unsigned a = in_range ? ld32(in + q) : 0u, c = in_range ? ld32(in + (mc + len + 4 * lane)) : 1u;
                    const unsigned x = a ^ c;
                    const unsigned long long neq = __ballot(x != 0u || !in_range);
                    if (neq == 0ull) {
                        len += 256;
                        continue;
                    }
                    const int fl = __builtin_ctzll(neq);
                    len += 4 * fl;
                    // the lane that stopped the run: a differing byte inside its dword, or the end of the comparable range
                    const unsigned qf = mp + len;
                    if (qf + 4 <= matchlimit) {
                        const unsigned xf = (unsigned)__builtin_amdgcn_readlane((int)x, fl);
                        len += (unsigned)__builtin_ctz(xf) >> 3;
                    } else {
                        while (mp + len < matchlimit && in[mp + len] == in[mc + len]) ++len;  // at most three bytes (uniform loop)
                    }
                    break;
                }
                // sequence: token, literal length bytes, literals, offset, match length bytes
                const unsigned lit = mp - anchor, ml = len - MINMATCH;
                const unsigned need = 1 + ext_bytes(lit) + lit + 2 + ext_bytes(ml);
                if (op + need + (n - (mp + len)) / 255 + 16 >= n) {  // would not shrink: stored instead
                    fits = false;
                    break;
                }
                if (lane == 0) out[op] = (uint8_t)((min(lit, 15u) << 4) | min(ml, 15u));
                ++op;
                if (lit >= 15) {
                    put_ext(out + op, lit, lane);
                    op += ext_bytes(lit);
                }
                wave_copy(out + op, in + anchor, lit, lane);
                op += lit;
                if (lane == 0) {
                    const unsigned off = mp - mc;
                    out[op] = (uint8_t)(off & 255u);
                    out[op + 1] = (uint8_t)(off >> 8);
                }
                op += 2;
                if (ml >= 15) {
                    put_ext(out + op, ml, lane);
                    op += ext_bytes(ml);
                }
                const unsigned end = mp + len;
                anchor = end;
                if (end >= wend) {
                    next_ip = end;
                    break;
                }
                m &= ~0ull << (end - ip);  // the lanes behind this match
            }
            ip = next_ip;
        }
        unsigned csize = n;
        if (fits) {
            const unsigned lit = n - anchor;
            const unsigned need = 1 + ext_bytes(lit) + lit;
            if (op + need < n) {
                if (lane == 0) out[op] = (uint8_t)(min(lit, 15u) << 4);
                ++op;
                if (lit >= 15) {
                    put_ext(out + op, lit, lane);
                    op += ext_bytes(lit);
                }
                wave_copy(out + op, in + anchor, lit, lane);
                op += lit;
                csize = op;
            } else {
                fits = false;
            }
        }
        if (!fits) wave_copy(out, in, n, lane);  // stored block
        if (lane == 0) sizes[b] = csize;
    }
}

// Blosc-1 frames of `nframes` chunks of `cbytes` permuted bytes each (the last block of a chunk may be short), blocks never
// split (flag 0x10): frame f = 16-byte header, nb block starts, then per block an int32 size and the payload.
//   scan_kernel : one workgroup per frame — block starts (exclusive scan of 4 + size) and the frame's total
//   offsets     : frame f starts at foff[f] (16-byte aligned) in the packed output; foff[nframes] = total bytes
__global__ __launch_bounds__(256) void frame_scan_kernel(const uint32_t* __restrict__ sizes, uint32_t nb, uint32_t* __restrict__ bstarts,
                                                         uint32_t* __restrict__ fbytes) {
    __shared__ uint32_t part[256];
    const uint32_t f = blockIdx.x;
    const uint32_t* s = sizes + (uint64_t)f * nb;
    uint32_t* bs = bstarts + (uint64_t)f * nb;
    const uint32_t per = (nb + 255) / 256, b0 = threadIdx.x * per, b1 = min(nb, b0 + per);
    uint32_t acc = 0;
    for (uint32_t b = b0; b < b1; ++b) acc += 4 + s[b];
    part[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 16 + 4 * nb;
        for (int i = 0; i < 256; ++i) {
            const uint32_t t = part[i];
            part[i] = run;
            run += t;
        }
        fbytes[f] = run;
    }
    __syncthreads();
    uint32_t pos = part[threadIdx.x];
    for (uint32_t b = b0; b < b1; ++b) {
        bs[b] = pos;
        pos += 4 + s[b];
    }
}
__global__ void frame_offsets_kernel(const uint32_t* __restrict__ fbytes, uint32_t nframes, uint64_t* __restrict__ foff) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        uint64_t run = 0;
        for (uint32_t f = 0; f < nframes; ++f) {
            foff[f] = run;
            run += ((uint64_t)fbytes[f] + 15) & ~(uint64_t)15;
        }
        foff[nframes] = run;
    }
}
// one wavefront per block copies its payload into place; the first wavefront of a frame also writes header and table
__global__ __launch_bounds__(256) void frame_pack_kernel(const uint8_t* __restrict__ slots, uint64_t slot, const uint32_t* __restrict__ sizes,
                                                         const uint32_t* __restrict__ bstarts, const uint32_t* __restrict__ fbytes,
                                                         const uint64_t* __restrict__ foff, uint32_t nb, uint32_t nframes, uint32_t cbytes,
                                                         uint32_t blocksize, uint32_t typesize, uint32_t flags, uint8_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const uint64_t w = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= (uint64_t)nframes * nb) return;
    const uint32_t f = (uint32_t)(w / nb), b = (uint32_t)(w - (uint64_t)f * nb);
    uint8_t* fr = out + foff[f];
    if (b == 0) {
        if (lane == 0) {
            fr[0] = 2;  // blosc format version
            fr[1] = 1;  // inner codec format version
            fr[2] = (uint8_t)flags;
            fr[3] = (uint8_t)typesize;
            st32(fr + 4, cbytes);
            st32(fr + 8, min(blocksize, cbytes));  // c-blosc refuses a block size beyond the buffer (a chunk smaller than one block)
            st32(fr + 12, fbytes[f]);
        }
        for (uint32_t j = lane; j < nb; j += 64) st32(fr + 16 + 4 * j, bstarts[(uint64_t)f * nb + j]);
    }
    const uint32_t cs = sizes[w], pos = bstarts[w];
    if (lane == 0) st32(fr + pos, cs);
    wave_copy(fr + pos + 4, slots + w * slot, cs, lane);
}

// ---- decompression: one wavefront per LZ4 stream --------------------------------------------------------------------------
// stream i: csize[i] bytes at src + soff[i] -> dlen[i] bytes at dst + doff[i]; csize == dlen marks a stored stream (Blosc keeps what
// does not shrink raw).  A Blosc block is one stream, or `typesize` streams when the writer split it.  status[0] is raised on a
// corrupt stream.
//
// A token stream is a chain of dependent reads — token, length bytes, literals, offset, match — and the first decoder took every
// link from global memory: two or three memory round trips (plus a wait for its own stores before each match) per sequence,
// 68 ms for a 537-MB volume.  Here a wavefront keeps both ends in LDS: the compressed bytes are staged 4 KiB at a time, and the
// output lives in a 16-KiB ring — literals go stage -> ring, matches ring -> ring (64 lanes wide, periodic in the offset when a
// match overlaps its own output), and every completed 4 KiB of the ring is written to global memory with 16-B stores.  Only a
// match reaching further back than the ring reads global memory (the part of the output already flushed; waited for, read past
// the vector cache).  A link of the chain is then an LDS access.
constexpr int D_RING = 16384, D_STAGE = 4096, D_FLUSH = 4096, D_WAVES = 4;
constexpr unsigned D_RMASK = D_RING - 1;

__global__ __launch_bounds__(64 * D_WAVES) void decompress_kernel(const uint8_t* __restrict__ src, const uint64_t* __restrict__ soff,
                                                                 const uint32_t* __restrict__ csize, const uint64_t* __restrict__ doff,
                                                                 const uint32_t* __restrict__ dlen, uint32_t nstreams, uint8_t* __restrict__ dst,
                                                                 int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) uint8_t dlds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint8_t* ring = dlds + (size_t)wave * (D_RING + D_STAGE);
    uint8_t* stage = ring + D_RING;
    for (uint32_t b = blockIdx.x * D_WAVES + wave; b < nstreams; b += gridDim.x * D_WAVES) {
        const uint8_t* in = src + soff[b];
        const unsigned cs = csize[b];
        uint8_t* out = dst + doff[b];
        const unsigned n = dlen[b];
        if (cs == n) {
            wave_copy(out, in, n, lane);
            continue;
        }
        unsigned ip = 0, op = 0, flushed = 0;
        unsigned sbase = 0, sfill = 0;  // the stage holds in[sbase, sbase + sfill)
        bool bad = false;
        const bool out16 = (reinterpret_cast<uintptr_t>(out) & 15) == 0;
        auto refill = [&](unsigned from) {  // (uniform) stage in[from, from + D_STAGE)
            sbase = from;
            sfill = min((unsigned)D_STAGE, cs - from);
            const uint8_t* g = in + from;
            const unsigned whole = sfill & ~3u;
            unsigned v[D_STAGE / 256];  // every load in flight before the first store
#pragma unroll
            for (int k = 0; k < D_STAGE / 256; ++k) {
                const unsigned i = 256 * k + 4 * lane;
                v[k] = i < whole ? ld32(g + i) : 0u;
            }
#pragma unroll
            for (int k = 0; k < D_STAGE / 256; ++k) {
                const unsigned i = 256 * k + 4 * lane;
                if (i < whole) st32(stage + i, v[k]);
            }
            if (whole + lane < sfill) stage[whole + lane] = g[whole + lane];
        };
        auto byte_at = [&](unsigned pos) -> unsigned {  // (uniform) one byte of the compressed stream, pos < cs
            if (pos - sbase >= sfill) refill(pos);
            return stage[pos - sbase];
        };
        // ring [from, to) -> global; from is a multiple of D_FLUSH, to - from <= D_FLUSH
        auto flush = [&](unsigned from, unsigned to) {
            const unsigned len = to - from;
            const uint8_t* r = ring + (from & D_RMASK);  // (a D_FLUSH-aligned piece never wraps)
            if (out16) {
                const unsigned whole = len & ~15u;
                for (unsigned i = 16 * lane; i < whole; i += 1024)
                    *reinterpret_cast<uint4*>(out + from + i) = *reinterpret_cast<const uint4*>(r + i);
                if (whole + lane < len) out[from + whole + lane] = r[whole + lane];
            } else {
                for (unsigned i = lane; i < len; i += 64) out[from + i] = r[i];
            }
        };
        auto advance = [&](unsigned m) {  // op += m (m <= D_FLUSH), completed pieces of the ring leave for global memory
            op += m;
            while (op - flushed >= (unsigned)D_FLUSH) {
                flush(flushed, flushed + D_FLUSH);
                flushed += D_FLUSH;
            }
        };
        while (ip < cs) {
            const unsigned tok = byte_at(ip++);
            unsigned lit = tok >> 4;
            if (lit == 15) {
                unsigned e;
                do {
                    if (ip >= cs) { bad = true; break; }
                    e = byte_at(ip++);
                    lit += e;
                } while (e == 255);
            }
            if (bad || ip + lit > cs || op + lit > n) { bad = true; break; }
            while (lit > 0) {  // literals: stage -> ring, at most what the stage holds and one flush piece at a time
                if (ip - sbase >= sfill) refill(ip);
                const unsigned m = min(min(lit, sbase + sfill - ip), (unsigned)D_FLUSH);
                const uint8_t* g = stage + (ip - sbase);
                for (unsigned j = lane; j < m; j += 64) ring[(op + j) & D_RMASK] = g[j];
                ip += m;
                lit -= m;
                advance(m);
            }
            if (ip >= cs) break;  // the last sequence has no match
            if (ip + 2 > cs) { bad = true; break; }
            unsigned off = byte_at(ip);
            off |= byte_at(ip + 1) << 8;
            ip += 2;
            unsigned ml = tok & 15u;
            if (ml == 15) {
                unsigned e;
                do {
                    if (ip >= cs) { bad = true; break; }
                    e = byte_at(ip++);
                    ml += e;
                } while (e == 255);
            }
            ml += MINMATCH;
            if (bad || off == 0 || off > op || op + ml > n) { bad = true; break; }
            while (ml > 0) {
                // one flush piece at a time.  A match that overlaps its own output (offset < length: runs) is periodic in the
                // offset — out[q] = out[q - off] all along — so every byte of the piece comes from the `off` bytes before it
                unsigned m = min(ml, (unsigned)D_FLUSH);
                if (off <= (unsigned)(D_RING - D_FLUSH)) {  // source inside the ring (which holds (op + m - D_RING, op + m])
                    const unsigned from = op - off;
                    if (off >= m) {
                        for (unsigned j = lane; j < m; j += 64) ring[(op + j) & D_RMASK] = ring[(from + j) & D_RMASK];
                    } else {
                        unsigned r = (unsigned)lane % off;         // (j mod off), advanced by 64 mod off per step
                        const unsigned step = 64u % off;
                        for (unsigned j = lane; j < m; j += 64) {
                            ring[(op + j) & D_RMASK] = ring[(from + r) & D_RMASK];
                            r += step;
                            r -= r >= off ? off : 0u;
                        }
                    }
                } else {
                    // beyond the ring: from the output already in global memory (its start is: off > D_RING - D_FLUSH > op - flushed)
                    m = min(min(m, off), flushed - (op - off));  // (no overlap inside the piece; its source already flushed)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's flushes have landed
                    const uint8_t* g = out + (op - off);
                    for (unsigned j = lane; j < m; j += 64)
                        ring[(op + j) & D_RMASK] = __hip_atomic_load(g + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                ml -= m;
                advance(m);
            }
        }
        if (!bad && op == n) {
            if (op > flushed) flush(flushed, op);
        } else if (lane == 0) {
            atomicExch(status, 1);
        }
    }
}

}  // namespace lz4

}  // namespace bh

extern "C" {

// src: nframes * cbytes permuted bytes (chunks back to back); out: packed Blosc frames; foff (device, nframes + 1 uint64): the
// frames' offsets in `out` and the total.  out must hold bh_blosc_lz4_bound(nframes, cbytes, blocksize) bytes.
uint64_t bh_blosc_lz4_bound(uint32_t nframes, uint32_t cbytes, uint32_t blocksize) {
    const uint64_t nb = (cbytes + (uint64_t)blocksize - 1) / blocksize;
    return (uint64_t)nframes * ((16 + 8 * nb + cbytes + 15) & ~(uint64_t)15);
}

int bh_blosc_lz4_compress(bh_ctx* ctx, const void* src, uint32_t nframes, uint32_t cbytes, uint32_t blocksize, uint32_t typesize,
                          int shuffle_mode, void* out, uint64_t* foff) {
    using namespace bh;
    BH_REQUIRE(ctx && src && out && foff, "NULL argument");
    BH_REQUIRE(nframes > 0 && cbytes >= 128 && blocksize >= 128 && blocksize <= (1u << 30), "invalid frame geometry");
    BH_REQUIRE(typesize >= 1 && typesize <= 255, "invalid typesize %u", typesize);
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    const uint32_t nb = (uint32_t)(((uint64_t)cbytes + blocksize - 1) / blocksize);
    BH_REQUIRE((uint64_t)nframes * nb < (1ull << 31), "too many blocks");
    const uint32_t nblocks = nframes * nb;
    const uint64_t slot = ((uint64_t)blocksize + 15) & ~(uint64_t)15;
    uint8_t* slots;
    uint32_t *sizes, *bstarts, *fbytes;
    BH_TRY(get_scratch(ctx, "lz4_slots", (uint64_t)nblocks * slot, (void**)&slots));
    BH_TRY(get_scratch(ctx, "lz4_sizes", (uint64_t)nblocks * 4, (void**)&sizes));
    BH_TRY(get_scratch(ctx, "lz4_bstarts", (uint64_t)nblocks * 4, (void**)&bstarts));
    BH_TRY(get_scratch(ctx, "lz4_fbytes", (uint64_t)nframes * 4, (void**)&fbytes));
    hipStream_t s = ctx->stream;
    const int grid = (int)std::min<uint64_t>(((uint64_t)nblocks + lz4::WAVES - 1) / lz4::WAVES, (uint64_t)ctx->num_cus * 2);
    // every block of every frame in one launch (frame by frame, a store with 60-MB chunks kept 30 CUs busy per launch: 35
    // launches of 5.7 ms for a 2-GB volume)
    hipLaunchKernelGGL(lz4::compress_kernel, dim3(grid), dim3(64 * lz4::WAVES), 0, s, (const uint8_t*)src, cbytes, (uint32_t)nb, blocksize, slots,
                       slot, sizes, nblocks);
    hipLaunchKernelGGL(lz4::frame_scan_kernel, dim3(nframes), dim3(256), 0, s, sizes, nb, bstarts, fbytes);
    hipLaunchKernelGGL(lz4::frame_offsets_kernel, dim3(1), dim3(64), 0, s, fbytes, nframes, foff);
    const uint32_t flags = 0x10u | (1u << 5) | (shuffle_mode == BH_BLOSC_SHUFFLE ? 0x1u : (shuffle_mode == BH_BLOSC_BITSHUFFLE ? 0x4u : 0u));
    hipLaunchKernelGGL(lz4::frame_pack_kernel, dim3((unsigned)(((uint64_t)nblocks + 3) / 4)), dim3(256), 0, s, slots, slot, sizes, bstarts, fbytes,
                       foff, nb, nframes, cbytes, blocksize, typesize, flags, (uint8_t*)out);
    BH_CHECK_HIP(hipGetLastError());
    return BH_OK;
}

// LZ4 streams back to bytes on the device: stream i = csize[i] bytes at src + soff[i] -> dlen[i] bytes at dst + doff[i] (four
// device arrays of nstreams entries; csize == dlen marks a stored stream).  The caller parses the Blosc frame's block table on
// the host (a block is one stream, or typesize streams when split).  Synchronises (the status word is read back):
// BH_ERR_INVALID on a corrupt stream.
int bh_lz4_decompress_streams(bh_ctx* ctx, const void* src, const uint64_t* soff, const uint32_t* csize, const uint64_t* doff,
                              const uint32_t* dlen, uint32_t nstreams, void* dst) {
    using namespace bh;
    BH_REQUIRE(ctx && src && soff && csize && doff && dlen && dst, "NULL argument");
    BH_REQUIRE(nstreams > 0, "no streams");
    BH_CHECK_HIP(hipSetDevice(ctx->device));
    int* status;
    BH_TRY(get_scratch(ctx, "lz4_status", sizeof(int), (void**)&status));
    hipStream_t s = ctx->stream;
    BH_CHECK_HIP(hipMemsetAsync(status, 0, sizeof(int), s));
    const size_t lds = (size_t)lz4::D_WAVES * (lz4::D_RING + lz4::D_STAGE);  // 80 KiB: two workgroups per CU
    BH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lz4::decompress_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = (int)std::min<uint64_t>(((uint64_t)nstreams + lz4::D_WAVES - 1) / lz4::D_WAVES, (uint64_t)ctx->num_cus * 2);
    hipLaunchKernelGGL(lz4::decompress_kernel, dim3(grid), dim3(64 * lz4::D_WAVES), lds, s, (const uint8_t*)src, soff, csize, doff, dlen,
                       nstreams, (uint8_t*)dst, status);
    BH_CHECK_HIP(hipGetLastError());
    int h = 0;
    BH_CHECK_HIP(hipMemcpyAsync(&h, status, sizeof(int), hipMemcpyDeviceToHost, s));
    BH_CHECK_HIP(hipStreamSynchronize(s));
    BH_REQUIRE(h == 0, "corrupt LZ4 stream");
    return BH_OK;
}

}  // extern "C"
