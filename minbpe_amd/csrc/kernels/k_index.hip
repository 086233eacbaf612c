// k_index.hip -- the inverted slot index of the second slotted form (k_slots2.hip): hashing,
// updates, and the candidate list of a sparse merge pass (built inside k_select).
// Part of bpe_kernels.hip, which includes the parts in order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../bpe_device.h"
#include "k_common.hip"

namespace bpe {

// Inverted slot index (sparse passes): for every group of 32 slots, a Bloom filter of the PAIRS
// its slots hold -- IDX_H buckets of 32 bits (bit s = slot 32*g + s), three hash functions.  A pair
// belongs to the slot of its LEFT word (the boundary pair to the slot that ends with it), which
// is also the slot that has to run a merge of that pair.
constexpr uint32_t IDX_H = 32768;
__device__ __forceinline__ void pair_hash(uint32_t x, uint32_t y, uint32_t &h1, uint32_t &h2, uint32_t &h3) {
    h1 = ((x * 0x9E3779B1u) ^ (y * 0x85EBCA77u)) >> 17;
    h2 = ((x * 0xC2B2AE3Du) + (y * 0x27D4EB2Fu) + 0x165667B1u) >> 17;
    h3 = (((x + 0x7F4A7C15u) * 0xD6E8FEB9u) ^ ((y + 0x51ED270Bu) * 0xA24BAED5u)) >> 17;
}
__device__ __forceinline__ void index_add(uint32_t *__restrict__ idx, uint32_t owner, uint32_t x, uint32_t y) {
    uint32_t h1, h2, h3;
    pair_hash(x, y, h1, h2, h3);
    uint32_t *row = idx + (size_t)(owner >> 5) * IDX_H;
    const uint32_t bit = 1u << (owner & 31);
    atomicOr(&row[h1], bit);
    atomicOr(&row[h2], bit);
    atomicOr(&row[h3], bit);
}

// The candidate list of a sparse pass, made by ONE 1024-thread block (the block of k_select
// that makes the pair final): a mask per group of 32 slots from the filter rows of the pair's
// three hashes, compacted with a block-wide scan -- no atomics, order = slot order.
struct CandArgs {
    const uint32_t *idx, *dirty;
    uint32_t *cand;
    uint32_t T;
    uint32_t enable;  // 0: this iteration's a != b pass is a dense one
};
__device__ __forceinline__ void build_cand_list(const CandArgs &C, DevState *st, uint32_t a, uint32_t b) {
    __shared__ uint32_t s_wtot[16];
    __shared__ uint32_t s_base;
    const uint32_t nwords = (C.T + 31) / 32;
    uint32_t h1, h2, h3;
    pair_hash(a, b, h1, h2, h3);
    const bool all = st->gap != 0;  // short slots about: adjacency in slot numbers means nothing, visit everything
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nwords; base += 1024) {
        const uint32_t w = base + threadIdx.x;
        uint32_t m = 0;
        if (w < nwords) {
            if (all) {
                m = 0xFFFFFFFFu;
            } else {
                const uint32_t *row = C.idx + (size_t)w * IDX_H;
                m = (row[h1] & row[h2] & row[h3]) | C.dirty[w];
            }
            const uint32_t left = C.T - w * 32;
            if (left < 32) m &= (1u << left) - 1u;
        }
        const uint32_t c = (uint32_t)__popc(m);
        const uint32_t inc = wave_iscan_add(c);
        if (lane_id() == 63) s_wtot[wave_id()] = inc;
        __syncthreads();
        uint32_t off = s_base + inc - c, tot = 0;
        for (int v = 0; v < 16; v++) {
            const uint32_t x = s_wtot[v];
            if (v < wave_id()) off += x;
            tot += x;
        }
        while (m) {
            C.cand[off++] = w * 32 + (uint32_t)__ffs((int)m) - 1u;
            m &= m - 1u;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_base += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) st->ncand = s_base;
}

}  // namespace bpe
