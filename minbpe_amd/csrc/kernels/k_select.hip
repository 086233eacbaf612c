// k_select.hip -- K2: arg-max with the reference's first-occurrence tie-break; stream views.
// Part of bpe_kernels.hip, which includes the parts in order.
// (no include guard: bpe_kernels.hip includes this part once per geometry, namespace BPE_G)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../bpe_device.h"

namespace bpe {
namespace BPE_G {

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
    const uint64_t t = p / TILE2;
    if (t >= r.T) return false;
    const uint32_t m = r.hdr[t].meta;
    if ((uint32_t)(p % TILE2) >= (m & 0x7FFFFFFFu)) return false;
    w = ((m >> 31) ? r.b1 : r.b0)[p];
    return true;
}
__device__ __forceinline__ bool slot_next(const SlotRefH &r, uint64_t n, uint64_t p, uint32_t &w) {
    uint64_t t = p / TILE2;
    if ((uint32_t)(p % TILE2) + 1 < (r.hdr[t].meta & 0x7FFFFFFFu)) return slot_get(r, n, p + 1, w);
    for (t = t + 1; t < r.T; t++) {
        if (r.hdr[t].meta & 0x7FFFFFFFu) {
            w = r.hdr[t].w0;
            return true;
        }
    }
    return false;
}
__device__ __forceinline__ uint64_t slot_space(const SlotRefH &r, uint64_t) { return r.T * (uint64_t)TILE2; }
__device__ __forceinline__ uint64_t view_slots(const SlotRefH &r, uint64_t) { return r.T; }
__device__ __forceinline__ void view_slot(const SlotRefH &r, uint64_t, uint64_t u, uint32_t &len,
                                          const uint32_t *&src) {
    const uint32_t m = r.hdr[u].meta;
    len = m & 0x7FFFFFFFu;
    src = ((m >> 31) ? r.b1 : r.b0) + u * TILE2;
}
// slot-space positions: slot u starts at u * view_tile
__device__ __forceinline__ uint64_t view_tile(const SlotRef &) { return TILE; }
__device__ __forceinline__ uint64_t view_tile(const SlotRefH &) { return TILE2; }

// K2, block 0: global max over rowmax, then every pair that attains it (the candidates of the
// reference's first-occurrence tie-break, F3).  rowarg[x] names the column that attains row x's
// maximum, so the tied pairs are read off directly; only a row whose maximum is attained by
// several columns (ROWARG_MULTI) is scanned.
// select_core leaves the result in LDS / registers (every thread gets M and the number of tied
// pairs, TIE_CAP + 1 = too many to list; the pairs are in s_tied); select_body also writes it to st.
constexpr int SEL_RPT = 20;  // 20 x 1024 row maxima in registers; rows beyond (vocab > 20480) are read twice (L2 hits).  32 left the selection kernels
                               // (128 VGPRs at 1024 threads) ten of them in scratch memory, reloaded one by one inside select_core
// A lean iteration re-scans a few rows in the same launch that selects (k_rowsel_lean, k_lean.hip):
// those rows are EXCLUDED from the row-maxima array (a bitmap in LDS) and come in as extra
// (row, maximum, column) items instead.
struct SelExtra {
    const uint32_t *excl;  // LDS bitmap of excluded rows, or nullptr
    uint32_t n;            // extra items ...
    const uint32_t *row, *m, *arg;  // ... in LDS
};
// issue the loads of this thread's share of the row maxima (rowmax[2x] = maximum of row x,
// rowmax[2x + 1] = the column that attains it, same cache line)
__device__ __forceinline__ void select_load(const uint32_t *__restrict__ rowmax, uint32_t vcur, uint32_t (&rm)[SEL_RPT]) {
    const uint2 *__restrict__ rowma = reinterpret_cast<const uint2 *>(rowmax);
#pragma unroll
    for (int i = 0; i < SEL_RPT; i++) {
        const uint32_t x = threadIdx.x + 1024u * i;
        rm[i] = (x < vcur) ? rowma[x].x : 0u;
    }
}
__device__ __forceinline__ void select_core(const uint32_t *__restrict__ rowmax,
                                            const uint32_t *__restrict__ mat, uint32_t stride,
                                            uint32_t vcur, int32_t *s_tied, uint32_t *s_bits, uint32_t &M_out,
                                            uint32_t &nt_out, uint32_t (&rm)[SEL_RPT], const SelExtra &E,
                                            const uint32_t below = 0xFFFFFFFFu) {
    __shared__ uint32_t s_red[16];
    __shared__ uint32_t s_M, s_nrows, s_nt;
    __shared__ uint32_t s_rows[ARGMAX_ROWS];
    // every thread keeps its share in registers: the second look (which rows attain the
    // maximum) needs no second trip to memory
    const uint2 *__restrict__ rowma = reinterpret_cast<const uint2 *>(rowmax);
    constexpr int RPT = SEL_RPT;
    auto excluded = [&](uint32_t x) -> bool { return E.excl && ((E.excl[x >> 5] >> (x & 31)) & 1u); };
    // `below` (a second look, at a level under the maximum): row maxima of `below` or more do not count --
    // their pairs are already taken
    auto capped = [&](uint32_t v) -> uint32_t { return v < below ? v : 0u; };
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < RPT; i++) {
        if (excluded(threadIdx.x + 1024u * i)) rm[i] = 0u;
        rm[i] = capped(rm[i]);
        m = max(m, rm[i]);
    }
    for (uint32_t x = threadIdx.x + 1024u * RPT; x < vcur; x += 1024)
        if (!excluded(x)) m = max(m, capped(rowma[x].x));
    for (uint32_t i = threadIdx.x; i < E.n; i += 1024) m = max(m, capped(E.m[i]));
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
    M_out = M;
    nt_out = 0;
    if (M == 0) return;  // stats is empty: max() raises ValueError in the reference (F6)
    auto row_at_max = [&](uint32_t x, uint32_t y) {
        atomicOr(&s_bits[x >> 5], 1u << (x & 31));  // (zeroed by the caller; read only if a tie is open)
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
        if (rm[i] == M)  // (M > 0, rows beyond vcur hold 0; the column sits in the cache line just read)
            row_at_max(threadIdx.x + 1024u * i, rowma[threadIdx.x + 1024u * i].y);
    for (uint32_t x = threadIdx.x + 1024u * RPT; x < vcur; x += 1024) {
        if (excluded(x)) continue;
        const uint2 v = rowma[x];
        if (v.x == M) row_at_max(x, v.y);
    }
    for (uint32_t i = threadIdx.x; i < E.n; i += 1024)
        if (E.m[i] == M) row_at_max(E.row[i], E.arg[i]);
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
    nt_out = (nrows > ARGMAX_ROWS) ? (TIE_CAP + 1) : min(s_nt, (uint32_t)TIE_CAP + 1);
}

__device__ __forceinline__ void select_body(const uint32_t *__restrict__ rowmax,
                                            const uint32_t *__restrict__ mat, uint32_t stride,
                                            uint32_t vcur, DevState *st, int32_t *s_tied, uint32_t *s_bits) {
    uint32_t M, nt;
    uint32_t rm[SEL_RPT];
    select_load(rowmax, vcur, rm);
    select_core(rowmax, mat, stride, vcur, s_tied, s_bits, M, nt, rm, SelExtra{nullptr, 0u, nullptr, nullptr, nullptr});
    if (M == 0) {
        if (threadIdx.x == 0) {
            st->status = ST_EMPTY;
            st->count = 0;
            st->found = 0;
            st->sel_tie = 0;
        }
        return;
    }
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

// Tie-break through the inverted slot index (second slotted form, index live, no short slots, at
// most TIE_CAP tied pairs): block 0 finds the earliest occurrence of every tied pair by itself --
// wave v takes pairs v, v + 16, ...: the filter rows of the pair's three hashes give the slots that
// may hold it, in stream order; the wave scans each such slot (2048 words per round trip) until it
// finds the pair -- and the lowest position wins.  No other block, no hand-off.  Returns
// position << 7 | pair index (NOPOS: not found, the caller falls back to the sweep).
__device__ __forceinline__ unsigned long long slot_find_pair(const SlotRefH &ref, uint32_t t, uint32_t x, uint32_t y) {
    const int lane = lane_id();
    // this slot's header and the next one's, issued together (the caller made sure no slot is
    // short or empty: st->gap == 0, so the word after the slot is the next slot's first)
    // ... and the slot's words with them, speculatively from buffer 0 (where a slot lives unless an
    // a == b pass moved it) and whatever its length: one round trip instead of two.  Every slot has
    // TILE2 words of room, and the word after them is the next slot's room or the buffer's padding.
    constexpr int SB = TILE2 / 256;  // the whole slot in one batch: 4 stripes of 256 words
    const uint32_t *src = ref.b0 + (size_t)t * TILE2;
    uint4 v[SB];
    uint32_t nx[SB];
#pragma unroll
    for (int j = 0; j < SB; j++) {
        const uint32_t q = j * 256 + lane * 4;
        v[j] = *reinterpret_cast<const uint4 *>(src + q);
        nx[j] = src[q + 4];
    }
    const uint32_t m = ref.hdr[t].meta;
    const uint32_t after = (t + 1 < ref.T) ? ref.hdr[t + 1].w0 : INVALID_WORD;
    const uint32_t len = m & 0x7FFFFFFFu;
    if (len == 0) return NOPOS;
    if (m >> 31) {  // (uniform, rare) the slot lives in the other buffer: load again
        src = ref.b1 + (size_t)t * TILE2;
#pragma unroll
        for (int j = 0; j < SB; j++) {
            const uint32_t q = j * 256 + lane * 4;
            v[j] = *reinterpret_cast<const uint4 *>(src + q);
            nx[j] = src[q + 4];
        }
    }
    {
        const uint32_t base = 0;
#pragma unroll
        for (int j = 0; j < SB; j++) {
            const uint32_t q = j * 256 + lane * 4;
            if (q >= len) v[j] = make_uint4(INVALID_WORD, INVALID_WORD, INVALID_WORD, INVALID_WORD);
            if (q + 4 >= len) nx[j] = INVALID_WORD;
        }
#pragma unroll
        for (int j = 0; j < SB; j++) {
            const uint32_t q = base + j * 256 + lane * 4;
            uint32_t w[5] = {v[j].x, v[j].y, v[j].z, v[j].w, nx[j]};
            uint32_t hit = 4;
#pragma unroll
            for (int k = 3; k >= 0; k--) {
                if (q + k < len) {
                    const uint32_t nxt = (q + k + 1 < len) ? w[k + 1] : after;
                    if ((w[k] & IDMASK) == x && (nxt & NWMASK) == y) hit = k;
                }
            }
            const unsigned long long bal = __ballot(hit < 4);
            if (bal) {
                const int fl = __ffsll((long long)bal) - 1;
                const uint32_t hk = (uint32_t)__builtin_amdgcn_readlane((int)hit, fl);
                return (unsigned long long)t * TILE2 + base + j * 256 + fl * 4 + hk;
            }
        }
    }
    return NOPOS;
}

// s_pos != nullptr: EVERY tied pair's first occurrence is found (s_pos[p], NOPOS if the index leads to
// none) -- nobody stops because another pair occurs earlier; the chained merges of k_sel_lean need the
// whole order.
__device__ __forceinline__ unsigned long long tie_by_index(const SlotRefH &ref, const CandArgs &C,
                                                          const int32_t *s_tied, uint32_t nt,
                                                          unsigned long long *s_pos = nullptr) {
    __shared__ unsigned long long s_best;
    if (threadIdx.x == 0) s_best = NOPOS;
    __syncthreads();
    const int lane = lane_id();
    const uint32_t nwords = (C.T + 31) / 32;
    const bool all = s_pos != nullptr;
    for (uint32_t p = wave_id(); p < nt; p += 16) {
        const uint32_t x = (uint32_t)s_tied[2 * p], y = (uint32_t)s_tied[2 * p + 1];
        uint32_t h1, h2, h3;
        pair_hash(x, y, h1, h2, h3);
        unsigned long long found = NOPOS;
        bool done = false;
        for (uint32_t wb = 0; wb < nwords && !done; wb += 64) {
            if (!all && (__atomic_load_n(&s_best, __ATOMIC_RELAXED) >> 7) < (unsigned long long)wb * 32 * TILE2) break;  // cannot win
            const uint32_t w = wb + lane;
            uint32_t mk = 0;
            if (w < nwords) {
                mk = (C.idx[(size_t)h1 * C.stride + w] & C.idx[(size_t)h2 * C.stride + w] &
                      C.idx[(size_t)h3 * C.stride + w]) | C.dirty[w];
                const uint32_t left = C.T - w * 32;
                if (left < 32) mk &= (1u << left) - 1u;
            }
            unsigned long long bal = __ballot(mk != 0);
            while (bal && !done) {
                const int lw = __ffsll((long long)bal) - 1;
                bal &= bal - 1;
                uint32_t mm = (uint32_t)__builtin_amdgcn_readlane((int)mk, lw);
                while (mm) {
                    const uint32_t t = (wb + lw) * 32 + (uint32_t)__ffs((int)mm) - 1u;
                    mm &= mm - 1u;
                    if (!all && (__atomic_load_n(&s_best, __ATOMIC_RELAXED) >> 7) < (unsigned long long)t * TILE2) {
                        done = true;  // another pair already occurs before this slot
                        break;
                    }
                    const unsigned long long pos = slot_find_pair(ref, t, x, y);
                    if (pos != NOPOS) {
                        found = pos;
                        done = true;
                        break;
                    }
                }
            }
        }
        if (found != NOPOS && lane == 0) atomicMin(&s_best, (found << 7) | p);  // (p < TIE_CAP <= 128)
        if (all && lane == 0) s_pos[p] = found;
    }
    __syncthreads();
    return s_best;  // position << 7 | index of the pair in s_tied
}
// (the contiguous / first-form views have no index: never called)
__device__ __forceinline__ unsigned long long tie_by_index(const SlotRef &, const CandArgs &, const int32_t *,
                                                          uint32_t, unsigned long long * = nullptr) {
    return NOPOS;
}

// Tie-break, first resort: block 0 looks through the first TIE_WIN slots of the stream by itself.
// With many pairs tied (late in training: dozens) one of them almost surely occurs that early,
// and a hit inside the window is final -- every position beyond it is later.  One round trip for
// the window's headers, one for its words (16-byte loads, all in flight together).  Returns
// position << 32 | a << 16 | b  (NOPOS: no tied pair in the window).
constexpr int TIE_WIN = 8;
template <class Ref>
__device__ __forceinline__ unsigned long long tie_window(const Ref &ref, uint64_t n, const uint32_t *s_bits,
                                                         const int32_t *s_tied, uint32_t nt, uint32_t M,
                                                         const uint32_t *__restrict__ mat, uint32_t stride) {
    __shared__ unsigned long long s_win;
    if (threadIdx.x == 0) s_win = NOPOS;
    __syncthreads();
    const uint32_t nslots = (uint32_t)min((uint64_t)TIE_WIN, view_slots(ref, n));
    uint32_t len[TIE_WIN];
    const uint32_t *src[TIE_WIN];
#pragma unroll
    for (int u = 0; u < TIE_WIN; u++) {
        len[u] = 0;
        src[u] = nullptr;
        if ((uint32_t)u < nslots) view_slot(ref, n, (uint64_t)u, len[u], src[u]);
    }
    uint4 v[TIE_WIN];
    uint32_t nx[TIE_WIN];
    const uint32_t q = threadIdx.x * 4;
#pragma unroll
    for (int u = 0; u < TIE_WIN; u++) {
        v[u] = make_uint4(INVALID_WORD, INVALID_WORD, INVALID_WORD, INVALID_WORD);
        nx[u] = INVALID_WORD;
        if (q < len[u]) {
            v[u] = *reinterpret_cast<const uint4 *>(src[u] + q);
            if (q + 4 < len[u]) nx[u] = src[u][q + 4];
        }
    }
#pragma unroll
    for (int u = 0; u < TIE_WIN; u++) {
        if (q >= len[u]) continue;
        const uint32_t w[5] = {v[u].x, v[u].y, v[u].z, v[u].w, nx[u]};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (q + k >= len[u]) break;
            uint32_t w1 = w[k + 1];
            if (q + k + 1 >= len[u] && !slot_next(ref, n, (uint64_t)u * view_tile(ref) + q + k, w1)) continue;  // (last word of the slot)
            if (w1 & FLAG) continue;
            const uint32_t x = w[k] & IDMASK, y = w1 & IDMASK;
            if (!((s_bits[x >> 5] >> (x & 31)) & 1u)) continue;
            bool hit = false;
            if (nt <= TIE_CAP) {
                for (uint32_t t = 0; t < nt; t++)
                    hit |= (s_tied[2 * t] == (int32_t)x) & (s_tied[2 * t + 1] == (int32_t)y);
            } else {
                hit = mat[(size_t)x * stride + y] == M;
            }
            if (hit) {
                atomicMin(&s_win, ((unsigned long long)((uint64_t)u * view_tile(ref) + q + k) << 32) | (x << 16) | y);
                break;  // (my later positions are later)
            }
        }
    }
    __syncthreads();
    return s_win;
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
            st->chain_n = 0;  // (whatever chain of merges a lean selection had lined up, k_lean.hip, is void)
            __hip_atomic_store(&st->sel_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        for (uint32_t i = threadIdx.x; i < 2048; i += 1024) s_bits[i] = 0;
        __syncthreads();
        if (st->status == 0) select_body(rowmax, mat, stride, vcur, st, s_tied, s_bits);
        __syncthreads();
        if (threadIdx.x == 0) {  // (thread 0 wrote these fields itself; the block gets them through LDS)
            s_pair[0] = (!dist && st->status == 0 && st->sel_tie != 0);
            s_pair[1] = st->ntied;
            s_pair[2] = st->count;
        }
        __syncthreads();
        if (s_pair[0] && C.tie_window) {
            // a tie: the first slots of the stream, by this block alone
            const unsigned long long key = tie_window(ref, st->n[par], s_bits, s_tied, s_pair[1], s_pair[2], mat, stride);
            if (threadIdx.x == 0 && key != NOPOS) {
                st->a = (int32_t)((key >> 16) & 0xFFFFu);
                st->b = (int32_t)(key & 0xFFFFu);
                st->fin_a = st->a;
                st->fin_b = st->b;
                st->firstpos = key >> 32;
                st->found = 1;
                st->sel_tie = 0;
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            s_pair[0] = (C.tie_index && !dist && st->status == 0 && st->sel_tie != 0 && st->ntied <= TIE_CAP &&
                         st->gap == 0);
            s_pair[1] = st->ntied;
        }
        __syncthreads();
        if (s_pair[0]) {
            const unsigned long long key = tie_by_index(ref, C, s_tied, s_pair[1]);
            if (threadIdx.x == 0 && key != NOPOS) {
                const uint32_t pi = (uint32_t)(key & 127u);
                st->a = s_tied[2 * pi];
                st->b = s_tied[2 * pi + 1];
                st->fin_a = st->a;
                st->fin_b = st->b;
                st->firstpos = key >> 7;
                st->found = 1;
                st->sel_tie = 0;
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            s_go = (st->status == 0 && st->sel_tie != 0);
            if (s_go) {
                // the other blocks will read what this one wrote (tied list, count, firstpos): publish it
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&st->sel_flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                // nobody sweeps: the flag alone (high bit) tells the other blocks to leave
                __hip_atomic_store(&st->sel_flag, epoch | 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            s_pair[0] = (st->status == 0 && st->found != 0);
            s_pair[1] = (uint32_t)st->a;
            s_pair[2] = (uint32_t)st->b;
        }
        __syncthreads();
        if (C.enable && s_pair[0] && (s_pair[1] != s_pair[2] || C.aa)) build_cand_list(C, st, s_pair[1], s_pair[2]);
    } else if (threadIdx.x == 0) {
        bool ok = false, leave = false;
        for (uint32_t spins = 0; spins < LOOKBACK_SPINS; spins++) {
            const uint32_t f = __hip_atomic_load(&st->sel_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((f & 0x7FFFFFFFu) == (epoch & 0x7FFFFFFFu)) {
                ok = true;
                leave = (f >> 31) != 0;
                break;
            }
            __builtin_amdgcn_s_sleep(24);
        }
        if (leave) {
            s_go = 0;
        } else {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            // never sweep on a decision that was not seen: a block that silently skipped its share
            // could leave a later tied position as "the earliest" (wrong merge, no error)
            if (!ok) atomicExch(&st->status, ST_LOOKBACK);
            s_go = (ok && st->status == 0 && st->sel_tie != 0);
        }
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
            if (rowmax[2 * x] == M) atomicOr(&s_bits[x >> 5], 1u << (x & 31));
    }
    __syncthreads();
    const uint64_t n = st->n[par];
    const uint64_t nslots = view_slots(ref, n);
    for (uint64_t u = blockIdx.x; u < nslots; u += gridDim.x) {
        if (__hip_atomic_load(&st->firstpos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < u * view_tile(ref)) break;
        uint32_t len;
        const uint32_t *src;
        view_slot(ref, n, u, len, src);
        for (uint32_t q = threadIdx.x; q < len; q += 1024) {
            const uint32_t w0 = src[q];
            uint32_t w1;
            if (q + 1 < len) w1 = src[q + 1];
            else if (!slot_next(ref, n, u * view_tile(ref) + q, w1)) continue;
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
                atomicMin(&st->firstpos, (unsigned long long)(u * view_tile(ref) + q));
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
    if (C.enable && s_pair[0] && (s_pair[1] != s_pair[2] || C.aa)) build_cand_list(C, st, s_pair[1], s_pair[2]);
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

}  // namespace BPE_G
}  // namespace bpe
