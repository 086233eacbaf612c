// k_select.hip -- K2: arg-max with the reference's first-occurrence tie-break; stream views.
// Part of bpe_kernels.hip, which includes the parts in order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../bpe_device.h"
#include "k_common.hip"
#include "k_index.hip"

namespace bpe {

// ---------------------------------------------------------------------------
// stream views, used by the tie-break scans: contiguous (SlotRef, meta == nullptr), slotted
// with a meta word per slot (SlotRef, first form) or with 32-byte headers (SlotRefH).

__device__ __forceinline__ bool slot_get(const SlotRef &r, uint64_t n, uint64_t p, uint32_t &w) {
    if (!r.meta) {
        if (p >= n) return false;
        w = r.b0[p];
        return true;
    }
    const uint64_t t = p / TILE;
    if (t >= r.T) return false;
    const uint32_t m = r.meta[t];
    if ((uint32_t)(p % TILE) >= (m & 0x7FFFFFFFu)) return false;
    w = ((m >> 31) ? r.b1 : r.b0)[p];
    return true;
}
// the word that follows position p in stream order
__device__ __forceinline__ bool slot_next(const SlotRef &r, uint64_t n, uint64_t p, uint32_t &w) {
    if (!r.meta) return slot_get(r, n, p + 1, w);
    uint64_t t = p / TILE;
    if ((uint32_t)(p % TILE) + 1 < (r.meta[t] & 0x7FFFFFFFu)) return slot_get(r, n, p + 1, w);
    for (t = t + 1; t < r.T; t++)
        if (r.meta[t] & 0x7FFFFFFFu) return slot_get(r, n, t * TILE, w);
    return false;
}
__device__ __forceinline__ uint64_t slot_space(const SlotRef &r, uint64_t n) {
    return r.meta ? r.T * (uint64_t)TILE : n;
}
// slot u of the view as (length, where its words are); a contiguous stream is cut into TILE-sized pieces
__device__ __forceinline__ uint64_t view_slots(const SlotRef &r, uint64_t n) {
    return r.meta ? r.T : (n + TILE - 1) / TILE;
}
__device__ __forceinline__ void view_slot(const SlotRef &r, uint64_t n, uint64_t u, uint32_t &len,
                                          const uint32_t *&src) {
    if (!r.meta) {
        const uint64_t b = u * TILE;
        len = b >= n ? 0u : (uint32_t)min((uint64_t)TILE, n - b);
        src = r.b0 + b;
    } else {
        const uint32_t m = r.meta[u];
        len = m & 0x7FFFFFFFu;
        src = ((m >> 31) ? r.b1 : r.b0) + u * TILE;
    }
}

__device__ __forceinline__ bool slot_get(const SlotRefH &r, uint64_t, uint64_t p, uint32_t &w) {
    const uint64_t t = p / TILE;
    if (t >= r.T) return false;
    const uint32_t m = r.hdr[t].meta;
    if ((uint32_t)(p % TILE) >= (m & 0x7FFFFFFFu)) return false;
    w = ((m >> 31) ? r.b1 : r.b0)[p];
    return true;
}
__device__ __forceinline__ bool slot_next(const SlotRefH &r, uint64_t n, uint64_t p, uint32_t &w) {
    uint64_t t = p / TILE;
    if ((uint32_t)(p % TILE) + 1 < (r.hdr[t].meta & 0x7FFFFFFFu)) return slot_get(r, n, p + 1, w);
    for (t = t + 1; t < r.T; t++) {
        if (r.hdr[t].meta & 0x7FFFFFFFu) {
            w = r.hdr[t].w0;
            return true;
        }
    }
    return false;
}
__device__ __forceinline__ uint64_t slot_space(const SlotRefH &r, uint64_t) { return r.T * (uint64_t)TILE; }
__device__ __forceinline__ uint64_t view_slots(const SlotRefH &r, uint64_t) { return r.T; }
__device__ __forceinline__ void view_slot(const SlotRefH &r, uint64_t, uint64_t u, uint32_t &len,
                                          const uint32_t *&src) {
    const uint32_t m = r.hdr[u].meta;
    len = m & 0x7FFFFFFFu;
    src = ((m >> 31) ? r.b1 : r.b0) + u * TILE;
}

// K2, block 0: global max over rowmax, then every pair that attains it (the candidates of the
// reference's first-occurrence tie-break, F3).  rowarg[x] names the column that attains row x's
// maximum, so the tied pairs are read off directly; only a row whose maximum is attained by
// several columns (ROWARG_MULTI) is scanned.
__device__ __forceinline__ void select_body(const uint32_t *__restrict__ rowmax,
                                            const uint32_t *__restrict__ mat, uint32_t stride,
                                            uint32_t vcur, DevState *st) {
    __shared__ uint32_t s_red[16];
    __shared__ uint32_t s_M, s_nrows, s_nt;
    __shared__ uint32_t s_rows[ARGMAX_ROWS];
    __shared__ int32_t s_tied[2 * TIE_CAP];
    const uint32_t *__restrict__ rowarg = rowmax + stride;
    // every thread keeps its share of rowmax in registers: the second look (which rows attain the
    // maximum) needs no second trip to memory
    constexpr int RPT = 32;  // 32 x 1024 rows in registers; rows beyond (vocab > 32768) are read twice
    uint32_t rm[RPT];
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < RPT; i++) {
        const uint32_t x = threadIdx.x + 1024u * i;
        rm[i] = (x < vcur) ? rowmax[x] : 0u;
        m = max(m, rm[i]);
    }
    for (uint32_t x = threadIdx.x + 1024u * RPT; x < vcur; x += 1024) m = max(m, rowmax[x]);
    m = wave_max_u32(m);
    if (lane_id() == 0) s_red[wave_id()] = m;
    if (threadIdx.x == 0) {
        s_nrows = 0;
        s_nt = 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t M = 0;
        for (int i = 0; i < 16; i++) M = max(M, s_red[i]);
        s_M = M;
    }
    __syncthreads();
    const uint32_t M = s_M;
    if (M == 0) {  // stats is empty: max() raises ValueError in the reference (F6)
        if (threadIdx.x == 0) {
            st->status = ST_EMPTY;
            st->count = 0;
            st->found = 0;
            st->sel_tie = 0;
        }
        return;
    }
    auto row_at_max = [&](uint32_t x) {
        const uint32_t y = rowarg[x];
        if (y != ROWARG_MULTI) {
            const uint32_t s = atomicAdd(&s_nt, 1u);
            if (s < TIE_CAP) {
                s_tied[2 * s] = (int32_t)x;
                s_tied[2 * s + 1] = (int32_t)y;
            }
        } else {
            const uint32_t s = atomicAdd(&s_nrows, 1u);
            if (s < ARGMAX_ROWS) s_rows[s] = x;
        }
    };
#pragma unroll
    for (int i = 0; i < RPT; i++)
        if (rm[i] == M) row_at_max(threadIdx.x + 1024u * i);  // (M > 0, rows beyond vcur hold 0)
    for (uint32_t x = threadIdx.x + 1024u * RPT; x < vcur; x += 1024)
        if (rowmax[x] == M) row_at_max(x);
    __syncthreads();
    const uint32_t nrows = s_nrows;
    if (nrows <= ARGMAX_ROWS && s_nt <= TIE_CAP) {
        for (uint32_t r = 0; r < nrows; r++) {
            const uint32_t x = s_rows[r];
            const uint32_t *row = mat + (size_t)x * stride;
            for (uint32_t y = threadIdx.x; y < vcur; y += 1024) {
                if (row[y] == M) {
                    const uint32_t s = atomicAdd(&s_nt, 1u);
                    if (s < TIE_CAP) {
                        s_tied[2 * s] = (int32_t)x;
                        s_tied[2 * s + 1] = (int32_t)y;
                    }
                }
            }
        }
    }
    __syncthreads();
    // more than TIE_CAP pairs (or unscanned multi rows): the sweep tests positions against the table
    const uint32_t nt = (nrows > ARGMAX_ROWS) ? (TIE_CAP + 1) : min(s_nt, (uint32_t)TIE_CAP + 1);
    if (threadIdx.x < 2 * min(nt, (uint32_t)TIE_CAP)) st->tied[threadIdx.x] = s_tied[threadIdx.x];
    if (threadIdx.x == 0) {
        st->count = M;
        st->ntied = nt;
        st->firstpos = NOPOS;
        if (nt == 1) {
            st->found = 1;
            st->a = s_tied[0];
            st->b = s_tied[1];
            st->fin_a = s_tied[0];
            st->fin_b = s_tied[1];
            st->sel_tie = 0;
        } else {
            st->found = 0;
            st->sel_tie = 1;
        }
    }
}

// K2 kernel.  Block 0 decides (select_body); the other blocks wait for its decision (one flag,
// agent-scope release/acquire -- cdna_hip_programming.md G16) and, only if a tie is open, ALL
// blocks sweep the stream front to back, slot by slot (block k takes slots k, k + grid, ...;
// coalesced reads inside a slot), for the earliest position holding a tied pair; a block stops
// as soon as an earlier position has been reported.  The last block to finish its sweep (a
// ticket) makes the pair final in st (unless `dist`: a sharded stream only yields this rank's
// candidate, the ranks compare positions).  One launch; block 0 never waits, so there is no
// circular dependency whatever the residency.
// The block that makes the pair final (block 0 when there is no tie, else the last sweeper) also
// makes the candidate list of this iteration's sparse merge pass, when the host asked for one (C).
template <class Ref>
__global__ void __launch_bounds__(1024)
k_select(const uint32_t *__restrict__ rowmax, const uint32_t *__restrict__ mat, uint32_t stride,
         uint32_t vcur, DevState *st, Ref ref, int par, int dist, uint32_t epoch, CandArgs C) {
    __shared__ int32_t s_tied[2 * TIE_CAP];
    __shared__ uint32_t s_go, s_pair[3];
    __shared__ uint32_t s_bits[2048];  // tokens x with rowmax[x] == M (vocab <= 65536)
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) {
            st->adj = 0;  // (the previous pass's format-B "adjacent sites" count was folded into the table)
            __hip_atomic_store(&st->sel_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (st->status == 0) select_body(rowmax, mat, stride, vcur, st);
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&st->sel_flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_go = (st->status == 0 && st->sel_tie != 0);
            s_pair[0] = (st->status == 0 && st->found != 0);
            s_pair[1] = (uint32_t)st->a;
            s_pair[2] = (uint32_t)st->b;
        }
        __syncthreads();
        if (C.enable && s_pair[0] && s_pair[1] != s_pair[2]) build_cand_list(C, st, s_pair[1], s_pair[2]);
    } else if (threadIdx.x == 0) {
        bool ok = false;
        for (uint32_t spins = 0; spins < LOOKBACK_SPINS; spins++) {
            if (__hip_atomic_load(&st->sel_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) {
                ok = true;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        // never sweep on a decision that was not seen: a block that silently skipped its share
        // could leave a later tied position as "the earliest" (wrong merge, no error)
        if (!ok) atomicExch(&st->status, ST_LOOKBACK);
        s_go = (ok && st->status == 0 && st->sel_tie != 0);
    }
    __syncthreads();
    if (!s_go) return;
    const uint32_t nt = st->ntied;
    const uint32_t M = st->count;
    if (nt <= TIE_CAP) {
        if (threadIdx.x < 2 * nt) s_tied[threadIdx.x] = st->tied[threadIdx.x];
    } else {
        // too many tied pairs to list: a position can only hold one if its left token's row
        // attains M -- a bitmap in LDS filters before the (random-access) table look-up
        for (uint32_t i = threadIdx.x; i < 2048; i += 1024) s_bits[i] = 0;
        __syncthreads();
        for (uint32_t x = threadIdx.x; x < vcur; x += 1024)
            if (rowmax[x] == M) atomicOr(&s_bits[x >> 5], 1u << (x & 31));
    }
    __syncthreads();
    const uint64_t n = st->n[par];
    const uint64_t nslots = view_slots(ref, n);
    for (uint64_t u = blockIdx.x; u < nslots; u += gridDim.x) {
        if (__hip_atomic_load(&st->firstpos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < u * TILE) break;
        uint32_t len;
        const uint32_t *src;
        view_slot(ref, n, u, len, src);
        for (uint32_t q = threadIdx.x; q < len; q += 1024) {
            const uint32_t w0 = src[q];
            uint32_t w1;
            if (q + 1 < len) w1 = src[q + 1];
            else if (!slot_next(ref, n, u * TILE + q, w1)) continue;
            if (w1 & FLAG) continue;
            const uint32_t x = w0 & IDMASK, y = w1 & IDMASK;
            bool hit = false;
            if (nt <= TIE_CAP) {
                for (uint32_t t = 0; t < nt; t++)
                    hit |= (s_tied[2 * t] == (int32_t)x) & (s_tied[2 * t + 1] == (int32_t)y);
            } else if ((s_bits[x >> 5] >> (x & 31)) & 1u) {
                hit = mat[(size_t)x * stride + y] == M;
            }
            if (hit) {
                atomicMin(&st->firstpos, (unsigned long long)(u * TILE + q));
                break;  // later positions of this thread cannot be earlier
            }
        }
    }
    if (dist) return;
    // every atomicMin of this block has been performed before its ticket is drawn
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        s_pair[0] = 0;
        const uint32_t ticket = __hip_atomic_fetch_add(&st->sel_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ticket == gridDim.x - 1) {  // last sweeper: the minimum is final
            const unsigned long long p = __hip_atomic_load(&st->firstpos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t w0 = 0, w1 = 0;
            if (p != NOPOS && slot_get(ref, n, p, w0) && slot_next(ref, n, p, w1)) {
                st->a = (int32_t)(w0 & IDMASK);
                st->b = (int32_t)(w1 & IDMASK);
                st->fin_a = st->a;
                st->fin_b = st->b;
                st->found = 1;
                s_pair[0] = 1;
                s_pair[1] = w0 & IDMASK;
                s_pair[2] = w1 & IDMASK;
            }  // else: found stays 0 and the merge pass raises ST_INTERNAL
        }
    }
    __syncthreads();
    if (C.enable && s_pair[0] && s_pair[1] != s_pair[2]) build_cand_list(C, st, s_pair[1], s_pair[2]);
}

// The pair to merge as every kernel after K2 sees it: decided by k_select, or (sharded streams)
// the pair found at the earliest tied position.
__device__ __forceinline__ bool resolved_pair(const DevState *st, const uint32_t *__restrict__ ids,
                                              uint32_t &a, uint32_t &b) {
    if (st->found) {
        a = (uint32_t)st->a;
        b = (uint32_t)st->b;
        return true;
    }
    const unsigned long long p = st->firstpos;
    if (p == NOPOS) return false;
    a = ids[p] & IDMASK;
    b = ids[p + 1] & IDMASK;
    return true;
}

template <class Ref>
__device__ __forceinline__ bool resolved_pair(const DevState *st, const Ref &ref, uint64_t n,
                                              uint32_t &a, uint32_t &b) {
    if (st->found) {
        a = (uint32_t)st->a;
        b = (uint32_t)st->b;
        return true;
    }
    const unsigned long long p = st->firstpos;
    uint32_t w0, w1;
    if (p == NOPOS || !slot_get(ref, n, p, w0) || !slot_next(ref, n, p, w1)) return false;
    a = w0 & IDMASK;
    b = w1 & IDMASK;
    return true;
}

// single-step API (bpe_argmax): make the decision final in st
__global__ void k_finalize(SlotRef ref, int par, DevState *st) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (st->status == 0 && !st->found) {
        uint32_t a, b;
        if (!resolved_pair(st, ref, st->n[par], a, b)) {
            st->status = ST_INTERNAL;  // a tie was reported but no tied pair is in the stream
        } else {
            st->a = (int32_t)a;
            st->b = (int32_t)b;
            st->found = 1;
        }
    }
}

// host-chosen pair for the single-step bpe_merge()
__global__ void k_set_pair(DevState *st, int32_t a, int32_t b) {
    st->a = a;
    st->b = b;
    st->fin_a = a;
    st->fin_b = b;
    st->found = 1;
    st->status = 0;
    st->count = 0;
}

}  // namespace bpe
