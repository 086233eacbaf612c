// k_lookback.hip -- K3 (option merge=1): single-pass merge with decoupled look-back.
// Part of bpe_kernels.hip, which includes the parts in order.
// (no include guard: bpe_kernels.hip includes this part once per geometry, namespace BPE_G)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../bpe_device.h"

namespace bpe {
namespace BPE_G {

// ---------------------------------------------------------------------------
// Single-pass merge: summary, carry/offset resolution and rewrite in ONE sweep
// over the ids (reads 4N, writes 4N' -- the three-pass form reads 8N).
//
// Chained scan with TWO-LEVEL decoupled look-back.  Tile t publishes its
// transducer summary ("aggregate") as soon as it has read its ids; the last
// tile of every group of 64 also publishes the group's aggregate.  A tile then
// resolves its carry and output offset in two hops: (1) the <= 63 tiles before
// it in its own group, (2) the groups before its group, 64 per hop, until one
// is found whose inclusive prefix is known.  With a single level the prefix
// frontier advances 64 tiles per L2 round trip (~1 us) -- measured: that alone
// caps the pass at ~2 TB/s; with two levels it advances 4096 tiles per hop.
//
// Descriptors are single 8-byte words written/read with agent-scope relaxed
// atomics (sc1: they bypass the non-coherent per-CU L1 / per-XCD L2), so the
// data IS the flag and no fence is needed (cdna_hip_programming.md G16, R2).
// They carry an epoch, so they never need clearing between launches.
//   bits 63..62 status (1 aggregate, 2 inclusive prefix)   bits 61..42 epoch
//   aggregate: bits 0..19 k0, 20..39 k1, 40 o0, 41 o1  (kept ids / carry-out per carry-in)
//   prefix   : bits 0..35 inclusive kept count, bit 36 carry-out
// Progress: tiles are workgroup ids, dispatched in order, so every tile a
// workgroup waits on is resident or done, and aggregates are published before
// any waiting; the spin is bounded anyway and raises ST_LOOKBACK, never hangs.
__device__ __forceinline__ unsigned long long desc_load(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void desc_store(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long desc_pack_agg(const TS &v, unsigned long long tag) {
    return (1ull << 62) | tag | v.k0 | (v.k1 << 20) | ((unsigned long long)(v.o & 3u) << 40);
}
__device__ __forceinline__ unsigned long long desc_pack_prefix(unsigned long long incl, uint32_t sout,
                                                               unsigned long long tag) {
    return (2ull << 62) | tag | (incl & 0xFFFFFFFFFull) | ((unsigned long long)sout << 36);
}
// descriptor -> transducer (a prefix is a constant function)
__device__ __forceinline__ TS desc_unpack(unsigned long long d, uint32_t stt) {
    TS v;
    if (stt == 2) {
        v.k0 = v.k1 = d & 0xFFFFFFFFFull;
        v.o = ((d >> 36) & 1u) ? 3u : 0u;
    } else {
        v.k0 = d & 0xFFFFFu;
        v.k1 = (d >> 20) & 0xFFFFFu;
        v.o = (uint32_t)((d >> 40) & 3u);
    }
    return v;
}

// One look-back hop over descriptors arr[base], arr[base-1], ... (lane i reads
// arr[base-i]; indices below `floor` do not exist: below 0 they act as the
// prefix (0, carry 0), otherwise they are simply outside the window).  Waits
// until the nearest prefix and every nearer descriptor are published, composes
// them far -> near.  Returns the composition in `win`; found_prefix tells
// whether it is absolute.  false on timeout.
__device__ __forceinline__ bool lookback_hop(const unsigned long long *arr, long long base,
                                             long long floor_idx, int count, uint32_t epoch,
                                             TS &win, bool &found_prefix, uint32_t tune) {
    const int lane = lane_id();
    const long long idx = base - lane;
    const bool inwin = lane < count && idx >= floor_idx;
    unsigned long long d = 0;
    uint32_t stt = 0;
    unsigned long long pmask = 0;
    for (uint32_t spins = 0;; spins++) {
        if (inwin) {
            if (idx >= 0) {
                d = desc_load(&arr[idx]);
                stt = ((d >> 42) & EPOCH_MASK) == (epoch & EPOCH_MASK) ? (uint32_t)(d >> 62) : 0u;
            } else {
                d = 0;
                stt = 2;  // before the stream: prefix 0, carry 0
            }
        } else {
            stt = 1;  // outside the window: neutral
        }
        pmask = __ballot(inwin && stt == 2);
        const int np = pmask ? (__ffsll((long long)pmask) - 1) : 63;
        if (!__ballot(inwin && stt == 0 && lane <= np)) break;
        if (spins > LOOKBACK_SPINS) return false;
        // back off: every poll is an L2-bypassing load that competes with the stream
        for (uint32_t z = 0; z < (tune & 0xFFu); z++) __builtin_amdgcn_s_sleep(8);
    }
    const int np = pmask ? (__ffsll((long long)pmask) - 1) : 64;
    TS v;
    if (!inwin || lane > np) {
        v.k0 = v.k1 = 0;
        v.o = 2u;  // identity
    } else {
        v = desc_unpack(d, stt);
    }
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {  // ordered: far tiles first, lane 0 last
        TS far;
        far.k0 = __shfl_down(v.k0, sft);
        far.k1 = __shfl_down(v.k1, sft);
        far.o = (uint32_t)__shfl_down((int)v.o, sft);
        if (lane + sft < 64) v = ts_then(far, v);
    }
    win.k0 = __shfl(v.k0, 0);
    win.k1 = __shfl(v.k1, 0);
    win.o = (uint32_t)__shfl((int)v.o, 0);
    found_prefix = pmask != 0;
    return true;
}

template <bool DELTA>
__global__ void __launch_bounds__(MT)
k_merge_lookback(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, DevState *st, int par,
                 unsigned long long *__restrict__ desc, unsigned long long *__restrict__ gdesc,
                 uint32_t epoch, uint32_t newid, uint32_t *__restrict__ delta, uint32_t vcap,
                 IterRec *rec, int iter, uint32_t *dirty_n, uint32_t tune) {
    __shared__ int s_wave[MT / 64];
    __shared__ uint32_t s_wsum[MT / 64];
    __shared__ SummaryLds s_sum;
    __shared__ unsigned long long s_excl;
    __shared__ uint32_t s_sin, s_fail;
    const uint64_t n = st->n[par];
    const uint64_t tile = blockIdx.x;
    const uint64_t tile_base = tile * TILE;
    uint32_t a = 0, b = 0;
    const bool ok = (st->status == 0) && resolved_pair(st, in, a, b);
    if (!ok) {
        // nothing to merge: tile 0 reports (empty stats, or a tie nobody resolved)
        if (tile == 0 && threadIdx.x == 0) {
            if (st->status == 0) st->status = ST_INTERNAL;
            if (dirty_n) *dirty_n = 0;
            if (rec) {
                rec[iter].a = st->a;
                rec[iter].b = st->b;
                rec[iter].count = st->count;
                rec[iter].status = st->status;
                rec[iter].new_len = n;
                __threadfence_system();
                rec[iter].seq = (unsigned long long)iter + 1;
            }
        }
        return;
    }
    if (tile_base >= n) return;
    const int len = (int)min((uint64_t)TILE, n - tile_base);
    Tile t;
    tile_load(t, in, n, tile_base, a, b, s_wave);
    const uint64_t w = tile_summary(t, len, s_sum);
    TS own;  // the tile as a transducer
    own.k0 = own.k1 = 0;
    uint32_t o0 = 0, o1 = 1;
    tile_step(w, (uint32_t)len, 0u, own.k0, o0);
    tile_step(w, (uint32_t)len, 1u, own.k1, o1);
    own.o = o0 | (o1 << 1);
    const unsigned long long tag = ((unsigned long long)(epoch & EPOCH_MASK)) << 42;
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const long long grp = (long long)(tile >> 6);
        const int li = (int)(tile & 63);
        bool fail = false;
        if (lane == 0) desc_store(&desc[tile], desc_pack_agg(own, tag));
        const bool fake = (tune >> 8) & 1u;  // measurement only: skip the waiting (wrong output)
        // Both windows are polled in the same round trip: lane i reads the
        // descriptor of tile t-1-i (my group only) AND of group grp-1-i.
        TS pre;
        pre.k0 = pre.k1 = 0;
        pre.o = 2u;
        if (!fake) {
            const long long i1 = (long long)tile - 1 - lane;   // level 1
            const bool in1 = lane < li;
            const long long i2 = grp - 1 - lane;               // level 2
            unsigned long long d1 = 0, d2 = 0;
            uint32_t s1 = 1, s2 = 1;
            unsigned long long p1 = 0, p2 = 0;
            bool done1 = false, pubbed = false;
            TS w1;
            w1.k0 = w1.k1 = 0;
            w1.o = 2u;
            for (uint32_t spins = 0;; spins++) {
                if (in1 && !done1) d1 = desc_load(&desc[i1]);
                if (i2 >= 0) d2 = desc_load(&gdesc[i2]);
                if (in1 && !done1)
                    s1 = ((d1 >> 42) & EPOCH_MASK) == (epoch & EPOCH_MASK) ? (uint32_t)(d1 >> 62) : 0u;
                s2 = (i2 >= 0) ? (((d2 >> 42) & EPOCH_MASK) == (epoch & EPOCH_MASK) ? (uint32_t)(d2 >> 62) : 0u)
                               : 2u;  // before the stream: prefix 0, carry 0
                if (i2 < 0) d2 = 0;
                if (!done1) {
                    p1 = __ballot(in1 && s1 == 2);
                    const int np1 = p1 ? (__ffsll((long long)p1) - 1) : 63;
                    done1 = !__ballot(in1 && s1 == 0 && lane <= np1);
                    if (done1) {  // compose my group's tiles before me, far -> near
                        const int np = p1 ? (__ffsll((long long)p1) - 1) : 64;
                        TS v;
                        if (!in1 || lane > np) {
                            v.k0 = v.k1 = 0;
                            v.o = 2u;
                        } else {
                            v = desc_unpack(d1, s1);
                        }
#pragma unroll
                        for (int sft = 1; sft < 64; sft <<= 1) {
                            TS far;
                            far.k0 = __shfl_down(v.k0, sft);
                            far.k1 = __shfl_down(v.k1, sft);
                            far.o = (uint32_t)__shfl_down((int)v.o, sft);
                            if (lane + sft < 64) v = ts_then(far, v);
                        }
                        w1.k0 = __shfl(v.k0, 0);
                        w1.k1 = __shfl(v.k1, 0);
                        w1.o = (uint32_t)__shfl((int)v.o, 0);
                    }
                }
                if (done1 && !p1 && li == 63 && !pubbed) {  // my group's aggregate, as early as possible
                    if (lane == 0) desc_store(&gdesc[grp], desc_pack_agg(ts_then(w1, own), tag));
                    pubbed = true;
                }
                if (done1 && p1) {  // a prefix inside my own group: absolute already
                    pre = w1;
                    break;
                }
                p2 = __ballot(s2 == 2);
                const int np2 = p2 ? (__ffsll((long long)p2) - 1) : 63;
                const bool done2 = !__ballot(s2 == 0 && lane <= np2);
                if (done1 && done2) {
                    const int np = p2 ? (__ffsll((long long)p2) - 1) : 64;
                    TS v;
                    if (lane > np) {
                        v.k0 = v.k1 = 0;
                        v.o = 2u;
                    } else {
                        v = desc_unpack(d2, s2);
                    }
#pragma unroll
                    for (int sft = 1; sft < 64; sft <<= 1) {
                        TS far;
                        far.k0 = __shfl_down(v.k0, sft);
                        far.k1 = __shfl_down(v.k1, sft);
                        far.o = (uint32_t)__shfl_down((int)v.o, sft);
                        if (lane + sft < 64) v = ts_then(far, v);
                    }
                    TS w2;
                    w2.k0 = __shfl(v.k0, 0);
                    w2.k1 = __shfl(v.k1, 0);
                    w2.o = (uint32_t)__shfl((int)v.o, 0);
                    pre = ts_then(w2, w1);
                    if (!p2) {  // 64 groups of aggregates and still no prefix: keep walking back
                        long long gb = grp - 1 - 64;
                        for (;;) {
                            TS win;
                            bool found = false;
                            if (!lookback_hop(gdesc, gb, -(1ll << 62), 64, epoch, win, found, tune)) {
                                fail = true;
                                break;
                            }
                            pre = ts_then(win, pre);
                            if (found) break;
                            gb -= 64;
                        }
                    }
                    break;
                }
                if (spins > LOOKBACK_SPINS) {
                    fail = true;
                    break;
                }
                for (uint32_t z = 0; z < (tune & 0xFFu); z++) __builtin_amdgcn_s_sleep(8);
            }
        }
        if (lane == 0) {
            if (fake) pre.k0 = tile * TILE;
            const unsigned long long excl = pre.k0;  // chain starts at a prefix: input-independent
            const uint32_t sin = pre.o & 1u;
            const unsigned long long incl = excl + (sin ? own.k1 : own.k0);
            const uint32_t sout = sin ? o1 : o0;
            const unsigned long long pd = desc_pack_prefix(incl, sout, tag);
            desc_store(&desc[tile], pd);
            if (li == 63) desc_store(&gdesc[grp], pd);
            s_excl = excl;
            s_sin = sin;
            s_fail = fail;
            if (fail) atomicExch(&st->status, ST_LOOKBACK);
            if (tile_base + TILE >= n) {  // last tile: totals, report, final pair
                st->n[par ^ 1] = incl;
                st->fin_a = (int32_t)a;
                st->fin_b = (int32_t)b;
                if (dirty_n) *dirty_n = 0;
                if (rec) {
                    rec[iter].a = (int32_t)a;
                    rec[iter].b = (int32_t)b;
                    rec[iter].count = st->count;
                    rec[iter].status = fail ? ST_LOOKBACK : 0u;
                    rec[iter].new_len = incl;
                    __threadfence_system();
                    rec[iter].seq = (unsigned long long)iter + 1;
                }
            }
        }
    }
    __syncthreads();
    if (s_fail) return;
    tile_rewrite<DELTA, false>(t, s_sin, a, b, newid, out + s_excl, s_wsum, delta, vcap, len, nullptr,
                               nullptr);
}

}  // namespace BPE_G
}  // namespace bpe
