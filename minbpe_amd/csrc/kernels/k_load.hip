// k_load.hip -- K0: bytes -> id stream, chunk starts, chunk weights.
// Part of bpe_kernels.hip, which includes the parts in order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../bpe_device.h"
#include "k_common.hip"

namespace bpe {

// ---------------------------------------------------------------------------
// K0: list(text_bytes)  (basic.py:25-26, regex.py:44)
// 16 B read -> 64 B written per lane; HBM-bound, 5 B of traffic per id.

__global__ void __launch_bounds__(256)
k_widen(const uint8_t *__restrict__ src, uint32_t *__restrict__ dst, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * 16;
    for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; i < n; i += stride) {
        if (i + 16 <= n) {
            const uint4 v = *reinterpret_cast<const uint4 *>(src + i);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint4 o;
                o.x = w[k] & 0xffu;
                o.y = (w[k] >> 8) & 0xffu;
                o.z = (w[k] >> 16) & 0xffu;
                o.w = w[k] >> 24;
                *reinterpret_cast<uint4 *>(dst + i + 4 * k) = o;
            }
        } else {
            for (uint64_t j = i; j < n; j++) dst[j] = src[j];
        }
    }
}

// ---------------------------------------------------------------------------
// K0 + K1 in ONE pass over the BYTES: list(text_bytes), the chunk starts of regex.py:44 and the first get_stats
// (base.py:13-22) of a train() -- what k_widen + k_mark_starts + k_pair_count_bytes do in three passes over 4-byte
// words (5 + 8/chunk + 4 + 4 bytes of HBM traffic per id) costs one read of the byte, one read of each chunk offset and
// one write of the word.  One 1024-thread workgroup per CU shares the whole 256 x 256 table as 16-bit counters in
// 128 KiB of LDS; nothing else is shared, so the pass has NO barrier:
//  * a WAVE owns a segment of LC_SEG consecutive positions at a time.  Lane l of sub-step s takes the four bytes at
//    256 s + 4 l (one aligned word per lane; the four id words it writes make every store a full 1 KiB wave access); the
//    byte after them comes from the next lane (DPP), not from memory;
//  * the chunk offsets that fall into the segment (k_seg_lb's table: lb[g] = first chunk starting at or after segment
//    g) set bits in the wave's own 132-word bitmap -- LDS operations of one wave execute in order, no fence needed
//    beyond the compiler's;
//  * every pair adds one to its counter with a RETURNING ds_add (measured, tools/lds_atomic_bench: 3.4 ns per wave
//    instruction per CU against 3.2 without the return -- 5,000 G lane-atomics/s over the chip on uniform addresses,
//    2,900 G on a skewed distribution: the histogram is nowhere near LDS-bound); the lane that takes a 16-bit counter
//    across 2^15 moves that 2^15 to the table in memory at once (one device atomic per 32,768 occurrences of a pair),
//    so there is no periodic flush -- the flush of k_pair_count_bytes, 16 K rounds x ~10^3 device atomics at the
//    chip's ~10 G/s, is what bounds that kernel, not its LDS atomics;
//  * a position that is not a pair (a chunk starts at its right neighbour, or the stream ends) adds to a per-lane
//    dummy word: under a GPT-style split every sixth position is such a non-pair.
constexpr uint32_t LC_SEG = 4096;                          // positions of one wave segment: 16 sub-steps x 64 lanes x 4
constexpr uint32_t LC_SUB = LC_SEG / 256;
constexpr uint32_t LC_BM_WORDS = LC_SEG / 32 + 4;          // chunk-start bits of [s0, s0 + LC_SEG], padded
constexpr int LC_WAVES = PC_THREADS / 64;
constexpr int LC_LDS_BYTES = 131072 + (LC_WAVES * (int)LC_BM_WORDS + 64) * 4;

__global__ void k_seg_lb(const uint64_t *__restrict__ off, uint64_t n_chunks, uint64_t segs, uint64_t *__restrict__ lb) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g > segs) return;
    const uint64_t target = g * LC_SEG;
    uint64_t lo = 0, hi = n_chunks;
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if (off[mid] < target) lo = mid + 1;
        else hi = mid;
    }
    lb[g] = lo;
}

__global__ void __launch_bounds__(PC_THREADS)
k_load_count(const uint8_t *__restrict__ bytes, const uint64_t *__restrict__ off, const uint64_t *__restrict__ lb,
             uint64_t n_chunks, uint64_t n, uint32_t *__restrict__ ids, uint32_t *__restrict__ mat, uint32_t stride) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_lc[];  // 32768 counter words | 16 bitmaps | 64 dummy words
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint32_t *s_bm = s_lc + 32768 + wv * LC_BM_WORDS;
    const uint32_t dummy = 32768u + LC_WAVES * LC_BM_WORDS + lane;
    for (uint32_t i = threadIdx.x; i < 32768u + LC_WAVES * LC_BM_WORDS + 64u; i += PC_THREADS) s_lc[i] = 0;
    __syncthreads();
    const uint64_t segs = (n + LC_SEG - 1) / LC_SEG;
    for (uint64_t g = (uint64_t)blockIdx.x * LC_WAVES + wv; g < segs; g += (uint64_t)gridDim.x * LC_WAVES) {
        const uint64_t s0 = g * LC_SEG;
        // (1) the bytes: every load of the segment in flight before anything waits
        uint32_t v[LC_SUB];
#pragma unroll
        for (uint32_t s = 0; s < LC_SUB; s++) {
            const uint64_t p = s0 + s * 256u + lane * 4u;
            v[s] = (p < n) ? *reinterpret_cast<const uint32_t *>(bytes + p) : 0u;  // (the buffer is padded by 16 bytes)
        }
        const uint32_t after = (s0 + LC_SEG < n) ? (uint32_t)bytes[s0 + LC_SEG] : 0u;  // (uniform)
        // (2) the chunks that start in [s0, s0 + LC_SEG] (bit LC_SEG: does the next segment begin with one)
        s_bm[lane] = 0;
        s_bm[64 + lane] = 0;
        if (lane < LC_BM_WORDS - 128) s_bm[128 + lane] = 0;
        if (off) {
            const uint64_t c0 = lb[g], c1 = lb[g + 1];
            for (uint64_t i = c0 + lane; i < c1; i += 64) {
                const uint64_t rel = off[i] - s0;  // (offsets are sorted -- bpe_load_bytes checks; the test keeps LDS safe regardless)
                if (rel <= LC_SEG) atomicOr(&s_bm[rel >> 5], 1u << (rel & 31u));
            }
            if (lane == 0 && c1 < n_chunks && off[c1] == s0 + LC_SEG) atomicOr(&s_bm[LC_SEG >> 5], 1u);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // (3) words out, pairs counted
#pragma unroll
        for (uint32_t s = 0; s < LC_SUB; s++) {
            const uint32_t rel = s * 256u + lane * 4u;
            const uint64_t p = s0 + rel;
            // the byte after my four: the next lane's first byte; lane 63: the next sub-step's first (the next segment's)
            const uint32_t up = (s + 1 < LC_SUB) ? lane_first(v[(s + 1) % LC_SUB]) & 0xFFu : after;
            const uint32_t nxb = lane_next(v[s] & 0xFFu, up);
            const unsigned long long bm = ((unsigned long long)s_bm[(rel >> 5) + 1] << 32) | s_bm[rel >> 5];
            const uint32_t f = (uint32_t)(bm >> (rel & 31u)) & 31u;  // chunk starts at p .. p + 4
            if (p >= n) continue;
            const uint32_t x[5] = {v[s] & 0xFFu, (v[s] >> 8) & 0xFFu, (v[s] >> 16) & 0xFFu, v[s] >> 24, nxb};
            uint4 o;
            o.x = x[0] | ((f & 1u) << 31);
            o.y = x[1] | ((f & 2u) << 30);
            o.z = x[2] | ((f & 4u) << 29);
            o.w = x[3] | ((f & 8u) << 28);
            if (p + 4 <= n) {
                *reinterpret_cast<uint4 *>(ids + p) = o;
            } else {
                if (p + 0 < n) ids[p + 0] = o.x;
                if (p + 1 < n) ids[p + 1] = o.y;
                if (p + 2 < n) ids[p + 2] = o.z;
            }
            uint32_t old[4], inc[4], idx[4];
            bool ok[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                ok[k] = (p + k + 1 < n) & !((f >> (k + 1)) & 1u);
                idx[k] = (x[k] << 8) | x[k + 1];
                inc[k] = (idx[k] & 1u) ? 0x10000u : 1u;
                old[k] = atomicAdd(&s_lc[ok[k] ? (idx[k] >> 1) : dummy], inc[k]);
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                // my add took the counter's top bit from 0 to 1: move that 2^15 to memory (the bit is clear again
                // long before the other half of the word could be reached)
                if (ok[k] && ((old[k] + inc[k]) & ~old[k] & (inc[k] << 15))) {
                    atomicSub(&s_lc[idx[k] >> 1], inc[k] << 15);
                    atomicAdd(&mat[(size_t)(idx[k] >> 8) * stride + (idx[k] & 0xFFu)], 0x8000u);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();  // (the bitmap is cleared by the next round: after its last reader)
    }
    __syncthreads();
    // what is left in the counters (two per word)
    for (uint32_t i = threadIdx.x * 4u; i < 32768u; i += PC_THREADS * 4u) {
        const uint4 q = *reinterpret_cast<const uint4 *>(&s_lc[i]);
        if ((q.x | q.y | q.z | q.w) == 0) continue;
        const uint32_t w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t w = w4[j];
            if (!w) continue;
            const uint32_t i0 = 2u * (i + (uint32_t)j);  // idx = a << 8 | b
            if (w & 0xFFFFu) atomicAdd(&mat[(size_t)(i0 >> 8) * stride + (i0 & 0xFFu)], w & 0xFFFFu);
            if (w >> 16) atomicAdd(&mat[(size_t)((i0 + 1) >> 8) * stride + ((i0 + 1) & 0xFFu)], w >> 16);
        }
    }
}

// int32 ids from the host (module-level get_stats/merge drop-ins): strip sign.
__global__ void k_mark_starts(uint32_t *ids, const uint64_t *__restrict__ off, uint64_t n_chunks,
                              uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks; c += stride) {
        const uint64_t o = off[c];
        if (o < n) atomicOr(&ids[o], FLAG);  // duplicate offsets (empty chunks) are idempotent
    }
}

// weighted chunks (N1): every word of chunk c carries the chunk's weight exponent
__global__ void k_mark_weights(uint32_t *ids, const uint64_t *__restrict__ off, const uint8_t *__restrict__ wexp,
                               uint64_t n_chunks, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks; c += stride) {
        const uint32_t e = (uint32_t)(wexp[c] & 31u) << WSHIFT;
        if (!e) continue;
        const uint64_t p0 = off[c], p1 = (c + 1 < n_chunks) ? off[c + 1] : n;
        for (uint64_t p = p0; p < p1 && p < n; p++) ids[p] |= e;
    }
}

}  // namespace bpe
