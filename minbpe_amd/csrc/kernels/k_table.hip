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
// FMTB: the a != b merge passes of the second slotted form (k_slots2.hip) write delta format B
// -- vector 0 = SL (pairs (L,a) -> (L,Z)), vector 1 = SR (pairs (b,R) -> (Z,R)), st->adj = pairs
// (b,a) -> (Z,Z) -- which expands to the four vectors here; a == b passes write format A.
template <bool FOLDED, bool FMTB = false>
__device__ __forceinline__ void apply_body(uint32_t *__restrict__ mat, uint32_t stride,
                                           uint32_t *__restrict__ delta, uint32_t vcap,
                                           uint32_t *__restrict__ rowmax, DevState *st, uint32_t Z,
                                           uint32_t *__restrict__ dirty_list,
                                           uint32_t *__restrict__ dirty_n, int par, IterRec *rec, int iter,
                                           int slot_finish, const uint32_t *adj_ptr = nullptr) {
    if (slot_finish && blockIdx.x == 0 && threadIdx.x == 0) {
        // slotted pass: new stream length and this iteration's record
        const unsigned long long n = st->n[par];
        unsigned long long nn = n;
        if (st->status == 0) {
            // (a deferred lean iteration, k_lean.hip, removed nothing: the length is carried forward,
            // so that the host's ping-pong parity and a re-packing enqueued behind it stay right)
            nn = n - st->removed;
            st->n[par ^ 1] = nn;
        }
        st->removed = 0;
        if (rec) {
            rec[iter].a = st->status == 0 ? st->fin_a : st->a;
            rec[iter].b = st->status == 0 ? st->fin_b : st->b;
            rec[iter].count = st->count;
            rec[iter].status = (st->status == 0 && st->defer) ? ST_DEFER : st->status;
            rec[iter].new_len = nn;
            __threadfence_system();
            rec[iter].seq = (unsigned long long)iter + 1;
        }
    }
    if (st->status || st->defer) return;
    // 8 lanes per token: each folds a quarter of the replicas (all its loads in flight at
    // once), then a 3-step shuffle sum.  The kernel is latency-bound, so width, not work, counts.
    const uint32_t g = threadIdx.x & 7u;
    const uint32_t t = blockIdx.x * (blockDim.x / 8) + (threadIdx.x >> 3);
    const bool live = t <= Z;
    const uint32_t a = (uint32_t)st->fin_a, b = (uint32_t)st->fin_b;
    uint32_t acc4[4] = {0, 0, 0, 0};
    const uint32_t nrep = 1u << (vcap >> 24);
    vcap &= 0xFFFFFFu;
    const bool fmtb = FMTB && a != b;
    const int nv = fmtb ? 2 : 4;
    if (FOLDED) {
        if (live && g == 0) {
#pragma unroll
            for (int v = 0; v < 4; v++) acc4[v] = delta[(size_t)v * vcap + t];
        }
    } else if (live) {
        // lane g folds replicas g, g+8, ...: four of them (all their loads) in flight at a time
        for (uint32_t r0 = g; r0 < nrep; r0 += 32) {
            uint32_t x[4][4];
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const uint32_t r = r0 + 8u * k;
                    x[k][v] = (r < nrep && v < nv) ? delta[delta_rep_off(r, vcap) + (size_t)v * vcap + t] : 0u;
                }
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    if (x[k][v]) delta[delta_rep_off(r0 + 8u * k, vcap) + (size_t)v * vcap + t] = 0;
                    acc4[v] += x[k][v];
                }
        }
    }
#pragma unroll
    for (int v = 0; v < 4; v++) {
        acc4[v] += (uint32_t)__shfl_xor((int)acc4[v], 1);
        acc4[v] += (uint32_t)__shfl_xor((int)acc4[v], 2);
        acc4[v] += (uint32_t)__shfl_xor((int)acc4[v], 4);
    }
    if (fmtb) {
        const uint32_t adj = adj_ptr ? *adj_ptr : st->adj;  // (st->adj is reset by the next k_select; sharded: the global sum)
        acc4[2] = acc4[0];
        acc4[3] = acc4[1] + (t == Z ? adj : 0u);
        acc4[1] += (t == a ? adj : 0u);
    }
    if (!live || g != 0) return;
    const uint32_t dl = acc4[0], dr = acc4[1], il = acc4[2], ir = acc4[3];
    bool dirty = (t == a) | (t == b) | (t == Z);  // always recomputed
    if (dl) {
        const uint32_t old = atomicSub(&mat[(size_t)t * stride + a], dl);
        if (t != Z && old == rowmax[2 * t]) dirty = true;
    }
    if (dr) atomicSub(&mat[(size_t)b * stride + t], dr);
    if (il) atomicAdd(&mat[(size_t)t * stride + Z], il);
    if (ir) atomicAdd(&mat[(size_t)Z * stride + t], ir);
    if (dirty) {
        dirty_list[atomicAdd(dirty_n, 1u)] = t;
    } else if (il) {
        // column Z was empty before this iteration: it is the only entry of this row that grew
        const uint32_t m = rowmax[2 * t];
        if (il > m) {
            reinterpret_cast<uint2 *>(rowmax)[t] = make_uint2(il, Z);
        } else if (il == m) {
            rowmax[2 * t + 1] = ROWARG_MULTI;  // a second column attains the row maximum
        }
    }
}

// One workgroup scans one row of the table: its maximum, and WHICH column attains it
// (rowarg = that column, or ROWARG_MULTI when several do) -- k_select then reads the tied pairs
// straight from rowarg instead of re-scanning every row at the maximum.  16-byte loads (rows are
// 256-byte aligned and padded with zero columns up to the stride).  zero_col >= 0: that entry
// is retired (set to 0) on the way.  Returns the result to thread 0.
__device__ __forceinline__ void row_scan(uint32_t *__restrict__ row, uint32_t ncols, int zero_col,
                                         unsigned long long *s_red, uint32_t &m_out, uint32_t &arg_out) {
    unsigned long long kf = 0, kl = 0;  // count << 32 | ~column  and  count << 32 | column
    const uint32_t n4 = (ncols + 3) & ~3u;
    for (uint32_t y = threadIdx.x * 4; y < n4; y += blockDim.x * 4) {
        uint4 q = *reinterpret_cast<const uint4 *>(row + y);
        if (zero_col >= 0 && (uint32_t)zero_col - y < 4u) {
            const uint32_t k = (uint32_t)zero_col - y;
            if (k == 0) q.x = 0; else if (k == 1) q.y = 0; else if (k == 2) q.z = 0; else q.w = 0;
            row[zero_col] = 0;
        }
        const uint32_t v[4] = {q.x, q.y, q.z, q.w};
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
        const int nw = blockDim.x >> 6;
        for (int w = 1; w < nw; w++) {
            kf = s_red[2 * w] > kf ? s_red[2 * w] : kf;
            kl = s_red[2 * w + 1] > kl ? s_red[2 * w + 1] : kl;
        }
        m_out = (uint32_t)(kf >> 32);
        const uint32_t cf = 0xFFFFFFFFu - (uint32_t)kf, cl = (uint32_t)kl;
        arg_out = (m_out == 0) ? 0u : (cf == cl ? cf : ROWARG_MULTI);
    }
}

// Recompute rowmax / rowarg for the queued rows; also retires the merged pair: after the
// merge no (a,b) remains (F2), whatever the a == b bookkeeping left there.
__device__ __forceinline__ void rowmax_body(uint32_t *__restrict__ mat, uint32_t stride, uint32_t vnew,
                                            uint32_t *__restrict__ rowmax, const DevState *st,
                                            const uint32_t *__restrict__ dirty_list,
                                            const uint32_t *__restrict__ dirty_n, uint32_t first,
                                            uint32_t step) {
    __shared__ unsigned long long s_red[32];
    if (st->status) return;
    const uint32_t a = (uint32_t)st->fin_a, b = (uint32_t)st->fin_b;
    const uint32_t nd = *dirty_n;
    for (uint32_t i = first; i < nd; i += step) {
        const uint32_t x = dirty_list[i];
        uint32_t m = 0, arg = 0;
        row_scan(mat + (size_t)x * stride, vnew, x == a ? (int)b : -1, s_red, m, arg);
        if (threadIdx.x == 0) {
            reinterpret_cast<uint2 *>(rowmax)[x] = make_uint2(m, arg);
        }
    }
}
__global__ void __launch_bounds__(1024)
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

// Table update of the second slotted form: blocks [0, na) apply the delta vectors (format B for
// a != b), blocks [na, grid) commit the headers a sparse merge pass staged (k_slots2.hip).
template <bool FOLDED>
__global__ void __launch_bounds__(256)
k_apply2(uint32_t *__restrict__ mat, uint32_t stride, uint32_t *__restrict__ delta, uint32_t vcap,
         uint32_t *__restrict__ rowmax, DevState *st, uint32_t Z, uint32_t *__restrict__ dirty_list,
         uint32_t *__restrict__ dirty_n, int par, IterRec *rec, int iter, uint32_t na,
         SlotHdr *__restrict__ hdr_cur, const StageRec *__restrict__ stage, uint32_t *__restrict__ removed,
         uint32_t *__restrict__ smask, uint32_t nwords) {
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        // ids removed by the merge pass: 256 counters, one per 256-byte line (see DELTA_SKEW)
        uint32_t v = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t x = removed[(threadIdx.x * 4 + i) * REMOVED_STRIDE];
            if (x) removed[(threadIdx.x * 4 + i) * REMOVED_STRIDE] = 0;
            v += x;
        }
        v = wave_sum_u32(v);
        if (threadIdx.x == 0) st->removed = v;  // read by this same thread in apply_body
    }
    if (blockIdx.x < na) {
        // FOLDED (sharded training): `delta` is the all-reduced payload [4][vcap] + the global adj word
        apply_body<FOLDED, true>(mat, stride, delta, vcap, rowmax, st, Z, dirty_list, dirty_n, par, rec, iter, 1,
                                 FOLDED ? delta + 4 * (size_t)(vcap & 0xFFFFFFu) : nullptr);
        return;
    }
    if (st->status || st->defer) return;
    // staged headers: smask[w] bit s = slot 32*w + s has a new header in stage[32*w + s]
    const uint32_t step = (gridDim.x - na) * blockDim.x;
    for (uint32_t w = (blockIdx.x - na) * blockDim.x + threadIdx.x; w < nwords; w += step) {
        uint32_t m = smask[w];
        if (!m) continue;
        smask[w] = 0;
        while (m) {
            const uint32_t t = w * 32 + (uint32_t)__ffs((int)m) - 1u;
            m &= m - 1u;
            const StageRec r = stage[t];
            uint4 *dst = reinterpret_cast<uint4 *>(hdr_cur + t);
            dst[0] = make_uint4(r.h[0], r.h[1], r.h[2], r.h[3]);
            dst[1] = make_uint4(r.h[4], r.h[5], r.h[6], r.h[7]);
        }
    }
}

}  // namespace bpe
