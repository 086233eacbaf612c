// k_select.hip -- K2: arg-max with the reference's first-occurrence tie-break; stream views.
// Part of bpe_kernels.hip, which includes the parts in order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../bpe_device.h"
#include "k_common.hip"

namespace bpe {

// ---------------------------------------------------------------------------
// stream views (contiguous or slotted), used by the tie-break scans

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

// the pair test of the tie-break: is (a, w1) one of the pairs tied at the max?
__device__ __forceinline__ bool tie_hit(const int32_t *s_tied, uint32_t nt, uint32_t M,
                                        const uint32_t *__restrict__ mat, uint32_t stride,
                                        uint32_t a, uint32_t w1) {
    if (nt <= TIE_CAP) {
        bool hit = false;
        for (uint32_t t = 0; t < nt; t++)
            hit |= (s_tied[2 * t] == (int32_t)a) & (s_tied[2 * t + 1] == (int32_t)w1);
        return hit;
    }
    return mat[(size_t)a * stride + w1] == M;
}

// K2, single workgroup: global max over rowmax, gather every pair that attains
// it (the candidates of the reference's first-occurrence tie-break, F3), and --
// if there is a tie -- search the first TIE_WINDOW0 positions of the stream for
// the earliest tied pair.  Ties among frequent pairs always resolve there; the
// rest of the stream is k_tiebreak's job.
__device__ __forceinline__ void select_body(const uint32_t *__restrict__ rowmax,
                                            const uint32_t *__restrict__ mat, uint32_t stride,
                                            uint32_t vcur, DevState *st, const SlotRef &ref, int par,
                                            int dist) {
    __shared__ uint32_t s_red[16];
    __shared__ uint32_t s_M, s_nrows, s_nt, s_first;
    __shared__ uint32_t s_rows[ARGMAX_ROWS];
    __shared__ int32_t s_tied[2 * TIE_CAP];
    if (st->status) return;
    uint32_t m = 0;
    for (uint32_t x = threadIdx.x; x < vcur; x += 1024) m = max(m, rowmax[x]);
    m = wave_max_u32(m);
    if (lane_id() == 0) s_red[wave_id()] = m;
    if (threadIdx.x == 0) {
        s_nrows = 0;
        s_nt = 0;
        s_first = 0xFFFFFFFFu;
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
        }
        return;
    }
    for (uint32_t x = threadIdx.x; x < vcur; x += 1024) {
        if (rowmax[x] == M) {
            const uint32_t s = atomicAdd(&s_nrows, 1u);
            if (s < ARGMAX_ROWS) s_rows[s] = x;
        }
    }
    __syncthreads();
    const uint32_t nrows = s_nrows;
    if (nrows <= ARGMAX_ROWS) {
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
    const uint32_t nt = (nrows > ARGMAX_ROWS) ? (TIE_CAP + 1) : min(s_nt, (uint32_t)TIE_CAP + 1);
    if (threadIdx.x < 2 * min(nt, (uint32_t)TIE_CAP)) st->tied[threadIdx.x] = s_tied[threadIdx.x];
    if (nt > 1) {  // tie: first window, positions ascending per thread
        const uint64_t n = st->n[par];
        const uint32_t hi = (uint32_t)min((uint64_t)TIE_WINDOW0, slot_space(ref, n));
        if (ref.meta) {
            // slot by slot: one meta lookup per slot, coalesced reads inside it
            for (uint32_t u = 0; u < hi / TILE + 1 && (uint64_t)u < ref.T; u++) {
                if (__atomic_load_n(&s_first, __ATOMIC_RELAXED) != 0xFFFFFFFFu) break;  // earlier slot hit
                const uint32_t mu = ref.meta[u];
                const uint32_t len = mu & 0x7FFFFFFFu;
                const uint32_t *src = ((mu >> 31) ? ref.b1 : ref.b0) + (size_t)u * TILE;
                for (uint32_t q = threadIdx.x; q < len; q += 1024) {
                    uint32_t w1;
                    if (q + 1 < len) w1 = src[q + 1];
                    else if (!slot_next(ref, n, (uint64_t)u * TILE + q, w1)) continue;
                    if (w1 & FLAG) continue;
                    if (tie_hit(s_tied, nt, M, mat, stride, src[q] & IDMASK, w1 & IDMASK)) {
                        atomicMin(&s_first, u * TILE + q);
                        break;
                    }
                }
                __syncthreads();
            }
        } else
        for (uint32_t p = threadIdx.x; p < hi; p += 1024) {
            if (__atomic_load_n(&s_first, __ATOMIC_RELAXED) < p) break;  // an earlier hit exists
            uint32_t w0, w1;
            if (!slot_get(ref, n, p, w0) || !slot_next(ref, n, p, w1) || (w1 & FLAG)) continue;
            if (tie_hit(s_tied, nt, M, mat, stride, w0 & IDMASK, w1 & IDMASK)) {
                atomicMin(&s_first, p);
                break;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        st->count = M;
        st->ntied = nt;
        st->firstpos = NOPOS;
        if (nt == 1) {
            st->found = 1;
            st->a = s_tied[0];
            st->b = s_tied[1];
        } else if (s_first != 0xFFFFFFFFu && dist) {
            st->found = 0;  // sharded stream: only a candidate, the ranks compare positions
            st->firstpos = s_first;
        } else if (s_first != 0xFFFFFFFFu) {
            uint32_t w0 = 0, w1 = 0;
            slot_get(ref, st->n[par], s_first, w0);
            slot_next(ref, st->n[par], s_first, w1);
            st->found = 1;
            st->a = (int32_t)(w0 & IDMASK);
            st->b = (int32_t)(w1 & IDMASK);
        } else {
            st->found = 0;
        }
    }
}

// K2 kernel.  Block 0 decides (select_body); the other blocks wait for its
// decision (one flag, agent-scope release/acquire -- cdna_hip_programming.md
// G16) and, only if a tie is open, ALL blocks sweep the stream front to back
// for the earliest position holding a tied pair (each sweep step covers
// gridDim*1024 consecutive positions, so a block stops as soon as an earlier
// position has been reported).  One launch instead of two; block 0 never waits,
// so there is no circular dependency whatever the residency.
__global__ void __launch_bounds__(1024)
k_select(const uint32_t *__restrict__ rowmax, const uint32_t *__restrict__ mat, uint32_t stride,
         uint32_t vcur, DevState *st, SlotRef ref, int par, int dist, uint32_t epoch) {
    __shared__ int32_t s_tied[2 * TIE_CAP];
    __shared__ uint32_t s_go;
    if (blockIdx.x == 0) {
        select_body(rowmax, mat, stride, vcur, st, ref, par, dist);
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&st->sel_flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_go = (st->status == 0 && st->found == 0 && st->firstpos == NOPOS);
        }
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
        s_go = (ok && st->status == 0 && st->found == 0 &&
                __atomic_load_n(&st->firstpos, __ATOMIC_RELAXED) == NOPOS);
    }
    __syncthreads();
    if (!s_go) return;
    const uint32_t nt = st->ntied;
    const uint32_t M = st->count;
    if (nt <= TIE_CAP && threadIdx.x < 2 * nt) s_tied[threadIdx.x] = st->tied[threadIdx.x];
    __syncthreads();
    const uint64_t n = st->n[par];
    const uint64_t space = slot_space(ref, n);
    // every block, block 0 included, sweeps: position order = (sweep step, block, thread)
    const uint64_t total = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t p = TIE_WINDOW0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < space; p += total) {
        if (__atomic_load_n(&st->firstpos, __ATOMIC_RELAXED) < p) break;
        uint32_t w0, w1;
        if (!slot_get(ref, n, p, w0) || !slot_next(ref, n, p, w1) || (w1 & FLAG)) continue;
        if (tie_hit(s_tied, nt, M, mat, stride, w0 & IDMASK, w1 & IDMASK)) {
            atomicMin(&st->firstpos, (unsigned long long)p);
            break;  // later positions of this thread cannot be earlier
        }
    }
}

// The pair to merge as every kernel after K2 sees it: decided by k_select, or
// the pair found at the earliest tied position by k_tiebreak.
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

__device__ __forceinline__ bool resolved_pair(const DevState *st, const SlotRef &ref, uint64_t n,
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
    st->found = 1;
    st->status = 0;
    st->count = 0;
}

}  // namespace bpe
