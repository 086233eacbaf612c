// k_table.hip -- delta mode: fold the delta vectors into the pair table, keep rowmax current.
// Part of bpe_kernels.hip, which includes the parts in order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../bpe_device.h"
#include "k_common.hip"

namespace bpe {

// Apply the four delta vectors to the dense table and keep rowmax[] current.
// Thread t owns token t: column a, row b, the new column Z and the new row Z.
// Rows whose maximum may have dropped are queued for k_rowmax_list; for every
// other row the only entry that grew is the brand-new column Z.
template <bool FOLDED>
__device__ __forceinline__ void apply_body(uint32_t *__restrict__ mat, uint32_t stride,
                                           uint32_t *__restrict__ delta, uint32_t vcap,
                                           uint32_t *__restrict__ rowmax, DevState *st, uint32_t Z,
                                           uint32_t *__restrict__ dirty_list,
                                           uint32_t *__restrict__ dirty_n, int par, IterRec *rec, int iter,
                                           int slot_finish) {
    if (slot_finish && blockIdx.x == 0 && threadIdx.x == 0) {
        // slotted pass: new stream length and this iteration's record
        const unsigned long long n = st->n[par];
        unsigned long long nn = n;
        if (st->status == 0) {
            nn = n - st->removed;
            st->n[par ^ 1] = nn;
        }
        st->removed = 0;
        if (rec) {
            rec[iter].a = st->status == 0 ? st->fin_a : st->a;
            rec[iter].b = st->status == 0 ? st->fin_b : st->b;
            rec[iter].count = st->count;
            rec[iter].status = st->status;
            rec[iter].new_len = nn;
            __threadfence_system();
            rec[iter].seq = (unsigned long long)iter + 1;
        }
    }
    if (st->status) return;
    // 8 lanes per token: each folds a quarter of the replicas (all its loads in flight at
    // once), then a 3-step shuffle sum.  The kernel is latency-bound, so width, not work, counts.
    const uint32_t g = threadIdx.x & 7u;
    const uint32_t t = blockIdx.x * (blockDim.x / 8) + (threadIdx.x >> 3);
    const bool live = t <= Z;
    const uint32_t a = (uint32_t)st->fin_a, b = (uint32_t)st->fin_b;
    uint32_t acc4[4] = {0, 0, 0, 0};
    const uint32_t nrep = 1u << (vcap >> 24);
    vcap &= 0xFFFFFFu;
    if (FOLDED) {
        if (live && g == 0) {
#pragma unroll
            for (int v = 0; v < 4; v++) acc4[v] = delta[(size_t)v * vcap + t];
        }
    } else if (live) {
        uint32_t x[4][4];
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const uint32_t r = g + 8u * k;
                x[k][v] = (r < nrep) ? delta[((size_t)r * 4 + v) * vcap + t] : 0u;
            }
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int v = 0; v < 4; v++) {
                if (x[k][v]) delta[((size_t)(g + 8u * k) * 4 + v) * vcap + t] = 0;
                acc4[v] += x[k][v];
            }
    }
#pragma unroll
    for (int v = 0; v < 4; v++) {
        acc4[v] += (uint32_t)__shfl_xor((int)acc4[v], 1);
        acc4[v] += (uint32_t)__shfl_xor((int)acc4[v], 2);
        acc4[v] += (uint32_t)__shfl_xor((int)acc4[v], 4);
    }
    if (!live || g != 0) return;
    const uint32_t dl = acc4[0], dr = acc4[1], il = acc4[2], ir = acc4[3];
    bool dirty = (t == a) | (t == b) | (t == Z);  // always recomputed
    if (dl) {
        const uint32_t old = atomicSub(&mat[(size_t)t * stride + a], dl);
        if (t != Z && old == rowmax[t]) dirty = true;
    }
    if (dr) atomicSub(&mat[(size_t)b * stride + t], dr);
    if (il) atomicAdd(&mat[(size_t)t * stride + Z], il);
    if (ir) atomicAdd(&mat[(size_t)Z * stride + t], ir);
    if (dirty) {
        dirty_list[atomicAdd(dirty_n, 1u)] = t;
    } else if (il > rowmax[t]) {
        rowmax[t] = il;  // column Z was empty before this iteration
    }
}

// Recompute rowmax for the queued rows; also retires the merged pair: after the
// merge no (a,b) remains (F2), whatever the a == b bookkeeping left there.
__device__ __forceinline__ void rowmax_body(uint32_t *__restrict__ mat, uint32_t stride, uint32_t vnew,
                                            uint32_t *__restrict__ rowmax, const DevState *st,
                                            const uint32_t *__restrict__ dirty_list,
                                            const uint32_t *__restrict__ dirty_n, uint32_t first,
                                            uint32_t step) {
    __shared__ uint32_t s_red[4];
    if (st->status) return;
    const uint32_t a = (uint32_t)st->fin_a, b = (uint32_t)st->fin_b;
    const uint32_t nd = *dirty_n;
    for (uint32_t i = first; i < nd; i += step) {
        const uint32_t x = dirty_list[i];
        uint32_t *row = mat + (size_t)x * stride;
        uint32_t m = 0;
        for (uint32_t y = threadIdx.x; y < vnew; y += 256) {
            uint32_t v = row[y];
            if (x == a && y == b) {
                v = 0;
                row[y] = 0;
            }
            m = max(m, v);
        }
        m = wave_max_u32(m);
        __syncthreads();
        if (lane_id() == 0) s_red[wave_id()] = m;
        __syncthreads();
        if (threadIdx.x == 0) rowmax[x] = max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3]));
    }
}
__global__ void __launch_bounds__(256)
k_rowmax_list(uint32_t *__restrict__ mat, uint32_t stride, uint32_t vnew,
              uint32_t *__restrict__ rowmax, const DevState *__restrict__ st,
              const uint32_t *__restrict__ dirty_list, const uint32_t *__restrict__ dirty_n) {
    rowmax_body(mat, stride, vnew, rowmax, st, dirty_list, dirty_n, blockIdx.x, gridDim.x);
}

// Table update in one launch: blocks [0, na) apply the delta vectors, blocks
// [na, gridDim) wait until all of them are done (a monotonic counter, agent-scope
// release/acquire) and recompute the queued row maxima.  The apply blocks never
// wait and come first in dispatch order, so the wait always ends.
template <bool FOLDED>
__global__ void __launch_bounds__(256)
k_apply_delta(uint32_t *__restrict__ mat, uint32_t stride, uint32_t *__restrict__ delta,
              uint32_t vcap, uint32_t *__restrict__ rowmax, DevState *st, uint32_t Z,
              uint32_t *__restrict__ dirty_list, uint32_t *__restrict__ dirty_n, int par, IterRec *rec,
              int iter, int slot_finish, uint32_t na, unsigned long long target) {
    if (blockIdx.x < na) {
        apply_body<FOLDED>(mat, stride, delta, vcap, rowmax, st, Z, dirty_list, dirty_n, par, rec, iter,
                           slot_finish);
        if (target == 0) return;  // row maxima run as their own launch (the default)
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(&st->apply_done, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    __shared__ uint32_t s_ok;
    if (threadIdx.x == 0) {
        bool ok = false;
        for (uint32_t spins = 0; spins < LOOKBACK_SPINS; spins++) {
            if (__hip_atomic_load(&st->apply_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) {
                ok = true;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (!ok) atomicExch(&st->status, ST_LOOKBACK);
        s_ok = ok;
    }
    __syncthreads();
    if (!s_ok) return;
    rowmax_body(mat, stride, Z + 1, rowmax, st, dirty_list, dirty_n, blockIdx.x - na, gridDim.x - na);
}

}  // namespace bpe
