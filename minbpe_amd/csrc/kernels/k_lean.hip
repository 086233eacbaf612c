// k_lean.hip -- the training iteration once a merged pair has few sites ("lean" iterations: the
// last ~29,000 of the 31,744 merges of a 1 GB / vocab-32000 run, where a pass rewrites a few
// hundred slots and nothing is bound by bytes any more, only by the number of launches and of
// dependent memory round trips inside them).  Four launches instead of five, shorter ones:
//   k_select            (k_select.hip)  the pair -- and no candidate list
//   k_merge_ab_lean     every wave finds its own candidate slots in the inverted index (the three
//                       filter rows of the pair, 1024 slots per step: no list, no single block
//                       building one) and rewrites them (merge_ab_wave, k_slots2.hip; delta format B)
//   k_apply2            (k_table.hip)   folds the delta into the pair table, staged headers, length, record
//   k_rowmax_lean       rows a, b, Z and the queued rows, one workgroup per row, the whole row in
//                       flight at once (k_rowmax_list walks a row 16 KB at a time)
// A pair with a == b is not merged here: the pass is DEFERRED -- the iteration reports ST_DEFER,
// everything enqueued behind it is a no-op that only carries the stream length forward, and the
// host re-runs the iteration through the general path (k_merge_aa).  75 of 31,744 merges; the
// general path pays a k_merge_aa launch in every iteration for them.
// Part of bpe_kernels.hip, which includes the parts in order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../bpe_device.h"
#include "k_index.hip"
#include "k_slots2.hip"
#include "k_table.hip"

namespace bpe {

// use_index == 0: no index (small streams) -- every live slot is visited
template <bool INDEXED>
__global__ void __launch_bounds__(MT, 4)
k_merge_ab_lean(AbArgs A, const uint32_t *__restrict__ idx_dirty, uint32_t use_index) {
    __shared__ __attribute__((aligned(16))) uint32_t s_out[MT / 64][TILE2];
    DevState *st = A.st;
    if (blockIdx.x == 0 && threadIdx.x == 0) *A.dirty_n = 0;  // (for the table update that follows)
    if (st->status || st->defer) return;
    if (!st->found) {
        if (blockIdx.x == 0 && threadIdx.x == 0) st->status = ST_INTERNAL;
        return;
    }
    const uint32_t a = (uint32_t)st->a, b = (uint32_t)st->b;
    if (a == b) {  // the general path's pass (k_merge_aa): the host re-runs this iteration there
        if (blockIdx.x == 0 && threadIdx.x == 0) st->defer = 1;
        return;
    }
    const uint32_t Tl = min(A.T, st->tlive);
    const uint32_t gw = blockIdx.x * (MT / 64) + wave_id(), nw = gridDim.x * (MT / 64);
    // short slots about: adjacency in slot numbers means nothing, visit everything (k_index.hip)
    if (!use_index || st->gap != 0) {
        for (uint32_t t = gw; t < Tl; t += nw) merge_ab_wave<true, INDEXED, false>(s_out[wave_id()], nullptr, t, A, a, b);
        return;
    }
    // One 32-slot word of the candidate mask per wave and step: the three filter words of the pair
    // AND-ed, plus the slots an a == b pass rewrote since the index was built.  The addresses are
    // wave-uniform (scalar loads); with a few hundred candidates among ~200 k slots almost every
    // wave finds nothing, and the ones that do are spread over the whole grid.
    const uint32_t nwords = (Tl + 31) / 32;
    uint32_t h1, h2, h3;
    pair_hash(a, b, h1, h2, h3);
    for (uint32_t w = gw; w < nwords; w += nw) {
        uint32_t mk = (A.idx[(size_t)h1 * A.istride + w] & A.idx[(size_t)h2 * A.istride + w] &
                       A.idx[(size_t)h3 * A.istride + w]) | idx_dirty[w];
        const uint32_t left = Tl - w * 32;
        if (left < 32) mk &= (1u << left) - 1u;
        mk = (uint32_t)__builtin_amdgcn_readfirstlane((int)mk);
        while (mk) {
            const uint32_t t = w * 32 + (uint32_t)__ffs((int)mk) - 1u;
            mk &= mk - 1u;
            merge_ab_wave<true, INDEXED, false>(s_out[wave_id()], nullptr, t, A, a, b);
        }
    }
}

// One row of the table by a 1024-thread workgroup, every load of the row in flight at once
// (vocab 32000: 8 x 16 bytes per thread); otherwise row_scan (k_table.hip).
__device__ __forceinline__ void row_scan_wide(uint32_t *__restrict__ row, uint32_t ncols, int zero_col,
                                              unsigned long long *s_red, uint32_t &m_out, uint32_t &arg_out) {
    unsigned long long kf = 0, kl = 0;  // count << 32 | ~column  and  count << 32 | column
    const uint32_t n4 = (ncols + 3) & ~3u;
    constexpr int U = 8;
    for (uint32_t base = 0; base < n4; base += U * 4096) {
        uint4 q[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t y = base + ((uint32_t)u * 1024u + threadIdx.x) * 4u;
            q[u] = (y < n4) ? *reinterpret_cast<const uint4 *>(row + y) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t y = base + ((uint32_t)u * 1024u + threadIdx.x) * 4u;
            uint32_t v[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
            if (zero_col >= 0 && (uint32_t)zero_col - y < 4u && y < n4) {
                v[(uint32_t)zero_col - y] = 0;
                row[zero_col] = 0;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (v[k]) {
                    const unsigned long long hi = (unsigned long long)v[k] << 32;
                    const unsigned long long f = hi | (0xFFFFFFFFu - (y + k)), l = hi | (y + k);
                    kf = f > kf ? f : kf;
                    kl = l > kl ? l : kl;
                }
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long of = __shfl_xor(kf, d), ol = __shfl_xor(kl, d);
        kf = of > kf ? of : kf;
        kl = ol > kl ? ol : kl;
    }
    __syncthreads();  // s_red may still be read by the previous row's thread 0
    if (lane_id() == 0) {
        s_red[2 * wave_id()] = kf;
        s_red[2 * wave_id() + 1] = kl;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; w++) {
            kf = s_red[2 * w] > kf ? s_red[2 * w] : kf;
            kl = s_red[2 * w + 1] > kl ? s_red[2 * w + 1] : kl;
        }
        m_out = (uint32_t)(kf >> 32);
        const uint32_t cf = 0xFFFFFFFFu - (uint32_t)kf, cl = (uint32_t)kl;
        arg_out = (m_out == 0) ? 0u : (cf == cl ? cf : ROWARG_MULTI);
    }
}

// Row maxima after k_apply2: workgroups 0, 1, 2 take rows a, b, Z (always re-scanned; (a,b) is
// retired on the way: no (a,b) survives the merge, F2), the others the queued rows -- k_apply2 queues
// a, b and Z too, those entries are skipped.  Everything a workgroup needs before its row is read
// in one round trip (the pair, the queue length, its queue entry).
__global__ void __launch_bounds__(1024)
k_rowmax_lean(uint32_t *__restrict__ mat, uint32_t stride, uint32_t *__restrict__ rowmax, const DevState *__restrict__ st,
              uint32_t Z, const uint32_t *__restrict__ dirty_list, const uint32_t *__restrict__ dirty_n) {
    __shared__ unsigned long long s_red[32];
    const uint32_t status = st->status, defer = st->defer;
    const uint32_t a = (uint32_t)st->fin_a, b = (uint32_t)st->fin_b;
    const uint32_t nd = *dirty_n;
    uint32_t q = blockIdx.x >= 3 ? dirty_list[blockIdx.x - 3] : 0u;  // (the list has room for every row: always readable)
    if (status || defer) return;
    for (uint32_t i = blockIdx.x; i < 3 + nd; i += gridDim.x) {
        uint32_t x;
        if (i < 3) {
            x = i == 0 ? a : (i == 1 ? b : Z);
            if (i == 1 && b == a) continue;
        } else {
            x = (i == blockIdx.x) ? q : dirty_list[i - 3];
            if (x == a || x == b || x == Z) continue;
        }
        uint32_t m = 0, arg = 0;
        row_scan_wide(mat + (size_t)x * stride, Z + 1, x == a ? (int)b : -1, s_red, m, arg);
        if (threadIdx.x == 0) reinterpret_cast<uint2 *>(rowmax)[x] = make_uint2(m, arg);
    }
}

// host: a deferred iteration is about to be re-run through the general path
__global__ void k_clear_defer(DevState *st) { st->defer = 0; }

}  // namespace bpe
