// k_index.hip -- the inverted slot index of the second slotted form (k_slots2.hip): hashing,
// updates, and the candidate list of a sparse merge pass (built inside k_select).
// Part of bpe_kernels.hip, which includes the parts in order.
// (no include guard: bpe_kernels.hip includes this part once per geometry, namespace BPE_G)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../bpe_device.h"

namespace bpe {
namespace BPE_G {

// Inverted slot index (sparse passes): for every group of 32 slots, a Bloom filter of the PAIRS
// its slots hold -- IDX_H buckets of 32 bits (bit s = slot 32*g + s), three hash functions.  A pair
// belongs to the slot of its LEFT word; a boundary pair is known to both slots it touches (the one
// that owns its site and the one that drops the site's second word).
// Layout: bucket-major, idx[h * stride + g].  A query reads the three bucket rows of one pair --
// contiguous, so that the single block that lists a pass's candidates (below) or breaks a tie
// (k_select) streams a few KB instead of touching one cache line per group.
// (32 Ki buckets for the ~1000 pairs of a 1024-id slot: about 9 % of a slot's bits set with three
// hashes, false positives around 0.07 %.  With 16 Ki buckets they were 0.46 %: ~800 of the 195 k slots of a
// late 1 GB stream per pair, more than the ~650 slots that do hold a late pair -- and a chain step's pass
// (k_chain.hip) visits the candidates of all its pairs.)
// (IDX_H, IDX_SHIFT: the geometry's, bpe_device.h -- 32 Ki buckets for 1024-id slots, 8 Ki for 256-id slots: the same density)
static_assert((1u << (32 - IDX_SHIFT)) == IDX_H, "hash width");
__device__ __forceinline__ void pair_hash(uint32_t x, uint32_t y, uint32_t &h1, uint32_t &h2, uint32_t &h3) {
    h1 = ((x * 0x9E3779B1u) ^ (y * 0x85EBCA77u)) >> IDX_SHIFT;
    h2 = ((x * 0xC2B2AE3Du) + (y * 0x27D4EB2Fu) + 0x165667B1u) >> IDX_SHIFT;
    h3 = (((x + 0x7F4A7C15u) * 0xD6E8FEB9u) ^ ((y + 0x51ED270Bu) * 0xA24BAED5u)) >> IDX_SHIFT;
}
__device__ __forceinline__ void index_add(uint32_t *__restrict__ idx, uint32_t stride, uint32_t owner, uint32_t x,
                                          uint32_t y) {
    uint32_t h1, h2, h3;
    pair_hash(x, y, h1, h2, h3);
    const uint32_t g = owner >> 5, bit = 1u << (owner & 31);
    atomicOr(&idx[(size_t)h1 * stride + g], bit);
    atomicOr(&idx[(size_t)h2 * stride + g], bit);
    atomicOr(&idx[(size_t)h3 * stride + g], bit);
#ifdef BPE_DUMMY_ATOMICS  // (experiment: what do the index's atomics cost a pass?  three more that change nothing)
    atomicOr(&idx[(size_t)(h1 ^ 1u) * stride + g], 0u);
    atomicOr(&idx[(size_t)(h2 ^ 1u) * stride + g], 0u);
    atomicOr(&idx[(size_t)(h3 ^ 1u) * stride + g], 0u);
#endif
}

// The candidate list of a sparse pass, made by ONE 1024-thread block (the block of k_select
// that makes the pair final): a mask per group of 32 slots from the filter rows of the pair's
// three hashes, compacted with a block-wide scan -- no atomics, order = slot order.
// a == b: the pass charges every pair to its LEFT element, so the slot before a candidate owes table
// updates too (the pair that ends at the candidate's first word): the list holds both.
__device__ __forceinline__ void build_cand_list(const CandArgs &C, DevState *st, uint32_t a, uint32_t b) {
    __shared__ uint32_t s_wtot[16];
    __shared__ uint32_t s_base;
    const uint32_t nwords = (C.T + 31) / 32;
    uint32_t h1, h2, h3;
    pair_hash(a, b, h1, h2, h3);
    const bool all = st->gap != 0;  // short slots about: adjacency in slot numbers means nothing, visit everything
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    // four groups per thread per round (all their loads in flight together): 131072 slots a round
    constexpr int GP = 4;
    for (uint32_t base = 0; base < nwords; base += GP * 1024) {
        uint32_t m[GP] = {0, 0, 0, 0};
        uint32_t c = 0;
        const uint32_t w4 = base + GP * threadIdx.x;
        uint32_t mnext = 0;  // (a == b) the mask word after my four
        if (a == b && w4 + GP < nwords && !all)
            mnext = (C.idx[(size_t)h1 * C.stride + w4 + GP] & C.idx[(size_t)h2 * C.stride + w4 + GP] &
                     C.idx[(size_t)h3 * C.stride + w4 + GP]) | C.dirty[w4 + GP];
        if (w4 < nwords && !all) {
            // (rows are 16-byte aligned and padded: the stride is a multiple of 4 groups)
            const uint4 r1 = *reinterpret_cast<const uint4 *>(C.idx + (size_t)h1 * C.stride + w4);
            const uint4 r2 = *reinterpret_cast<const uint4 *>(C.idx + (size_t)h2 * C.stride + w4);
            const uint4 r3 = *reinterpret_cast<const uint4 *>(C.idx + (size_t)h3 * C.stride + w4);
            const uint4 d = *reinterpret_cast<const uint4 *>(C.dirty + w4);
            m[0] = (r1.x & r2.x & r3.x) | d.x;
            m[1] = (r1.y & r2.y & r3.y) | d.y;
            m[2] = (r1.z & r2.z & r3.z) | d.z;
            m[3] = (r1.w & r2.w & r3.w) | d.w;
            if (a == b) {  // slot s is visited if s or s + 1 is a candidate
                m[0] |= (m[0] >> 1) | (m[1] << 31);
                m[1] |= (m[1] >> 1) | (m[2] << 31);
                m[2] |= (m[2] >> 1) | (m[3] << 31);
                m[3] |= (m[3] >> 1) | (mnext << 31);
            }
        }
#pragma unroll
        for (int u = 0; u < GP; u++) {
            const uint32_t w = w4 + u;
            if (w < nwords) {
                if (all) m[u] = 0xFFFFFFFFu;
                const uint32_t left = C.T - w * 32;
                if (left < 32) m[u] &= (1u << left) - 1u;
            } else {
                m[u] = 0;
            }
            c += (uint32_t)__popc(m[u]);
        }
        const uint32_t inc = wave_iscan_add(c);
        if (lane_id() == 63) s_wtot[wave_id()] = inc;
        __syncthreads();
        uint32_t off = s_base + inc - c, tot = 0;
        for (int v = 0; v < 16; v++) {
            const uint32_t x = s_wtot[v];
            if (v < wave_id()) off += x;
            tot += x;
        }
#pragma unroll
        for (int u = 0; u < GP; u++) {
            const uint32_t w = base + GP * threadIdx.x + u;
            uint32_t mm = m[u];
            while (mm) {
                C.cand[off++] = w * 32 + (uint32_t)__ffs((int)mm) - 1u;
                mm &= mm - 1u;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) s_base += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) st->ncand = s_base;
}

}  // namespace BPE_G
}  // namespace bpe
