// k_stats.hip -- K1: get_stats (pair histograms) and the row maxima.
// Part of bpe_kernels.hip, which includes the parts in order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../bpe_device.h"
#include "k_common.hip"
#include "k_table.hip"

namespace bpe {

// ---------------------------------------------------------------------------
// K1: get_stats  (base.py:13-22; shared dict over chunks regex.py:51-54)
//
// k_pair_count_simple: one global atomic per position. Used when the first
// position of every pair is wanted too (bpe_get_stats: dict insertion order).
template <bool FIRST>
__global__ void __launch_bounds__(256)
k_pair_count_simple(const uint32_t *__restrict__ ids, const DevState *__restrict__ st, int par,
                    uint32_t *__restrict__ mat, uint32_t stride, uint32_t *__restrict__ first) {
    const uint64_t n = st->n[par];
    const uint64_t total = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g * 4 < n; g += total) {
        const uint64_t p = g * 4;
        const uint4 v = *reinterpret_cast<const uint4 *>(ids + p);  // buffers are tile-padded
        uint32_t x[5] = {v.x, v.y, v.z, v.w, ids[p + 4]};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (p + k + 1 < n && !(x[k + 1] & FLAG)) {
                const size_t idx = (size_t)(x[k] & IDMASK) * stride + (x[k + 1] & IDMASK);
                atomicAdd(&mat[idx], word_weight(x[k]));
                if (FIRST) atomicMin(&first[idx], (uint32_t)(p + k));
            }
        }
    }
}

// k_pair_count_lds: the general histogram.  Each workgroup (1024 threads, one
// per CU) owns a contiguous span of the stream and a 16 Ki-slot LDS cache
// {key = table index, count} in all 128 KiB of dynamic LDS.  A position costs
// one ds_read + one ds_add on a hit; a key that finds its slot and the next
// three taken goes straight to an L2 atomic.  The cache is flushed once per
// workgroup (one global atomic per resident key), which turns the Zipf-hot
// pairs -- the ones that would serialise at one L2 channel -- into ~#CU atomics.
__device__ __forceinline__ void cache_add(uint32_t *keys, uint32_t *vals, uint32_t *__restrict__ g,
                                          uint32_t idx, uint32_t v) {
    uint32_t h = (idx * 0x9E3779B1u) >> (32 - PC_BITS);
#pragma unroll
    for (int probe = 0; probe < 4; probe++) {
        uint32_t k = __atomic_load_n(&keys[h], __ATOMIC_RELAXED);
        if (k == EMPTY_KEY) {
            const uint32_t old = atomicCAS(&keys[h], EMPTY_KEY, idx);
            k = (old == EMPTY_KEY) ? idx : old;
        }
        if (k == idx) {
            atomicAdd(&vals[h], v);
            return;
        }
        h = (h + 1) & ((1u << PC_BITS) - 1);
    }
    atomicAdd(&g[idx], v);
}

__global__ void __launch_bounds__(PC_THREADS)
k_pair_count_lds(const uint32_t *__restrict__ ids, const DevState *__restrict__ st, int par,
                 uint32_t *__restrict__ mat, uint32_t stride) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_pc[];
    uint32_t *s_keys = s_pc, *s_vals = s_pc + (1 << PC_BITS);
    for (int i = threadIdx.x; i < (1 << PC_BITS); i += PC_THREADS) {
        s_keys[i] = EMPTY_KEY;
        s_vals[i] = 0;
    }
    __syncthreads();
    const uint64_t n = st->n[par];
    // contiguous span per workgroup, rounded to whole 4-id groups per thread
    const uint64_t groups = (n + 3) / 4;
    const uint64_t per_wg = (groups + gridDim.x - 1) / gridDim.x;
    const uint64_t g0 = per_wg * blockIdx.x;
    const uint64_t g1 = min(g0 + per_wg, groups);
    for (uint64_t g = g0 + threadIdx.x; g < g1; g += PC_THREADS) {
        const uint64_t p = g * 4;
        const uint4 v = *reinterpret_cast<const uint4 *>(ids + p);
        const uint32_t x[5] = {v.x, v.y, v.z, v.w, ids[p + 4]};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (p + k + 1 < n && !(x[k + 1] & FLAG))
                cache_add(s_keys, s_vals, mat, (x[k] & IDMASK) * stride + (x[k + 1] & IDMASK), word_weight(x[k]));
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (1 << PC_BITS); i += PC_THREADS) {
        const uint32_t c = s_vals[i];
        if (c) atomicAdd(&mat[s_keys[i]], c);
    }
}

// k_pair_count_h32: the general histogram for unweighted streams (every pair counts 1).  Same
// plan as k_pair_count_lds -- one 1024-thread workgroup per CU over a contiguous span, all of its
// 128 KiB of LDS a cache, one flush -- with twice the slots: 32 Ki direct-mapped 4-byte slots,
//     slot word = tag (17 bits) << 15 | count (15 bits),
// where key = a << 16 | b is scrambled by an odd multiplier (a bijection of 32-bit words): the top
// 15 bits choose the slot, the low 17 are the tag, so slot + tag give the key back exactly at the
// flush.  A key that finds its slot taken by another goes straight to an L2 atomic.  A count that
// crosses 2^14 is drained by the one thread whose add crossed it (adds to one address are totally
// ordered), so 15 bits never overflow: at most 4096 adds are in flight in a workgroup.
constexpr uint32_t H32_MUL = 0x9E3779B1u, H32_INV = 0x0E8B2F51u;  // MUL * INV == 1 (mod 2^32)
static_assert((uint32_t)(H32_MUL * 0x0E8B2F51u) == 1u, "H32_INV is not the inverse of H32_MUL");
__global__ void __launch_bounds__(PC_THREADS)
k_pair_count_h32(const uint32_t *__restrict__ ids, const DevState *__restrict__ st, int par,
                 uint32_t *__restrict__ mat, uint32_t stride) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_pc[];  // 32768 slots
    for (int i = threadIdx.x; i < 32768; i += PC_THREADS) s_pc[i] = 0;
    __syncthreads();
    const uint64_t n = st->n[par];
    const uint64_t groups = (n + 3) / 4;
    const uint64_t per_wg = (groups + gridDim.x - 1) / gridDim.x;
    const uint64_t g0 = per_wg * blockIdx.x;
    const uint64_t g1 = min(g0 + per_wg, groups);
    constexpr int U = 2;  // 4-id groups per thread in flight
    for (uint64_t gb = g0; gb < g1; gb += (uint64_t)U * PC_THREADS) {
        uint4 v[U];
        uint32_t nx[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t g = gb + (uint64_t)u * PC_THREADS + threadIdx.x;
            if (g < g1) {
                v[u] = *reinterpret_cast<const uint4 *>(ids + g * 4);
                nx[u] = ids[g * 4 + 4];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t g = gb + (uint64_t)u * PC_THREADS + threadIdx.x;
            if (g >= g1) continue;
            const uint64_t p = g * 4;
            const uint32_t x[5] = {v[u].x, v[u].y, v[u].z, v[u].w, nx[u]};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (p + k + 1 >= n || (x[k + 1] & FLAG)) continue;
                const uint32_t a = x[k] & IDMASK, b = x[k + 1] & IDMASK;
                const uint32_t kk = ((a << 16) | b) * H32_MUL;
                const uint32_t slot = kk >> 17, tag = kk & 0x1FFFFu;
                uint32_t cur = __atomic_load_n(&s_pc[slot], __ATOMIC_RELAXED);
                if (cur == 0) {
                    const uint32_t old = atomicCAS(&s_pc[slot], 0u, (tag << 15) | 1u);
                    if (old == 0) continue;
                    cur = old;
                }
                if ((cur >> 15) == tag) {
                    const uint32_t old = atomicAdd(&s_pc[slot], 1u);
                    if (!(old & 0x4000u) && ((old + 1u) & 0x4000u)) {  // my add crossed 2^14: drain it
                        atomicSub(&s_pc[slot], 0x4000u);
                        atomicAdd(&mat[(size_t)a * stride + b], 0x4000u);
                    }
                } else {
                    atomicAdd(&mat[(size_t)a * stride + b], 1u);
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 32768; i += PC_THREADS) {
        const uint32_t w = s_pc[i];
        const uint32_t c = w & 0x7FFFu;
        if (c) {
            const uint32_t key = (((uint32_t)i << 17) | (w >> 15)) * H32_INV;
            atomicAdd(&mat[(size_t)(key >> 16) * stride + (key & 0xFFFFu)], c);
        }
    }
}

// k_pair_count_bytes: get_stats of a freshly widened stream (every id < 256) --
// the one full histogram a delta-mode train() runs.  The whole 256 x 256 table
// fits in LDS as 16-bit counters (two per word, 128 KiB): one ds_add per
// position, no keys, no probing.  A workgroup flushes every PCB_ROUND positions
// (< 65536), so a half-word can never carry into its neighbour.
__global__ void __launch_bounds__(PC_THREADS)
k_pair_count_bytes(const uint32_t *__restrict__ ids, const DevState *__restrict__ st, int par,
                   uint32_t *__restrict__ mat, uint32_t stride) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_pc[];  // 32768 words
    for (int i = threadIdx.x; i < 32768; i += PC_THREADS) s_pc[i] = 0;
    __syncthreads();
    const uint64_t n = st->n[par];
    const uint64_t groups = (n + 3) / 4;
    const uint64_t per_wg = (groups + gridDim.x - 1) / gridDim.x;
    const uint64_t g0 = per_wg * blockIdx.x;
    const uint64_t g1 = min(g0 + per_wg, groups);
    constexpr uint64_t ROUND_GROUPS = PCB_ROUND / 4;
    constexpr int U = 4;  // 4-id groups per thread in flight: the kernel is latency-bound otherwise
    for (uint64_t r0 = g0; r0 < g1; r0 += ROUND_GROUPS) {
        const uint64_t r1 = min(r0 + ROUND_GROUPS, g1);
        for (uint64_t gb = r0; gb < r1; gb += (uint64_t)U * PC_THREADS) {
            uint4 v[U];
            uint32_t nx[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint64_t g = gb + (uint64_t)u * PC_THREADS + threadIdx.x;
                if (g < r1) {
                    v[u] = *reinterpret_cast<const uint4 *>(ids + g * 4);
                    nx[u] = ids[g * 4 + 4];
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint64_t g = gb + (uint64_t)u * PC_THREADS + threadIdx.x;
                if (g >= r1) continue;
                const uint64_t p = g * 4;
                const uint32_t x[5] = {v[u].x, v[u].y, v[u].z, v[u].w, nx[u]};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    // branch-free: a position that is not a pair adds 0 (one ds_add per
                    // position either way; no exec-mask juggling around every atomic)
                    const bool ok = (p + k + 1 < n) & !(x[k + 1] & FLAG);
                    const uint32_t idx = ((x[k] & 0xFFu) << 8) | (x[k + 1] & 0xFFu);
                    atomicAdd(&s_pc[idx >> 1], ok ? ((idx & 1u) ? 0x10000u : 1u) : 0u);
                }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 32768; i += PC_THREADS) {
            const uint32_t w = s_pc[i];
            if (w) {
                s_pc[i] = 0;
                const uint32_t i0 = 2u * (uint32_t)i;  // idx = a<<8 | b
                if (w & 0xFFFFu) atomicAdd(&mat[(size_t)(i0 >> 8) * stride + (i0 & 0xFFu)], w & 0xFFFFu);
                if (w >> 16) atomicAdd(&mat[(size_t)((i0 + 1) >> 8) * stride + ((i0 + 1) & 0xFFu)], w >> 16);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// K2: pair = max(stats, key=stats.get)  (basic.py:35, regex.py:56)

// one workgroup per row: rowmax[x] = max_y count[x][y], rowarg[x] = the column attaining it
// (row_scan, k_table.hip; the two are interleaved: rowmax[2x] = maximum, rowmax[2x + 1] = column)
__global__ void __launch_bounds__(256)
k_rowmax_all(uint32_t *__restrict__ mat, uint32_t stride, uint32_t vcur, uint32_t *__restrict__ rowmax) {
    __shared__ unsigned long long s_red[8];
    const uint32_t x = blockIdx.x;
    uint32_t m = 0, arg = 0;
    row_scan(mat + (size_t)x * stride, vcur, -1, s_red, m, arg);
    if (threadIdx.x == 0) {
        reinterpret_cast<uint2 *>(rowmax)[x] = make_uint2(m, arg);
    }
}

}  // namespace bpe
