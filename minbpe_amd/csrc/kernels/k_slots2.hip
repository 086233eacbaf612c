// k_slots2.hip -- K3 in the training loop, second form: in-place slotted merge for a != b
// (dense and sparse passes), the a == b pass, the inverted slot index, re-packing.
// Part of bpe_kernels.hip, which includes the parts in order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../bpe_device.h"
#include "k_index.hip"
#include "k_merge.hip"
#include "k_lookback.hip"

namespace bpe {

// ---------------------------------------------------------------------------
// The stream is a sequence of T slots of TILE words; slot t holds `len` ids at the start of
// its TILE-word home in buffer 0 or 1 (SlotHdr.meta = len | buffer << 31).  Stream order is slot
// order, so first-occurrence order (F3) is that of slot-space positions t * TILE + offset.
//
// A merge of (a, b), a != b, touches a slot only where `a` is directly followed by `b`:
//   * the slot that owns the `a` of a site gets the new token there, and OWES the whole
//     pair-table update of that site (delta format B below), whichever slots the words around
//     the site live in -- two words of left context and three of right context come from the
//     neighbours' headers;
//   * a slot whose first word is the `b` of a site that started in the previous slot drops it.
// Every other slot is untouched and owes nothing.  A changed slot is compacted IN PLACE: the
// workgroup holds all 4096 words in registers before it stores, the output is staged in LDS and
// written back with 16-byte stores; nobody else reads a slot's words during an a != b pass
// (neighbours read HEADERS, which are double-buffered or staged).
//   * dense pass (k_merge_ab_dense): one workgroup per slot, headers ping-pong between two arrays;
//   * sparse pass (k_merge_ab_sparse): a resident grid; each workgroup owns a contiguous range
//     of slots and visits only those the inverted index cannot rule out; new headers go to a
//     staging list that the table-update kernel commits (the header array must stay intact
//     while neighbours read it).
// a == b needs the parity of runs of `a` across slot boundaries (F2): k_merge_aa keeps the
// first form's out-of-place scheme (write the other buffer, flip the slot's buffer bit; a run
// is walked back through the previous slots' words, which are not overwritten in that pass).
//
// Delta format B (a != b): with (a,b) -> Z at a site "L a b R",
//     (L,a) -> (L,Z)   unless L is the `b` of the previous site      SL[L] += w
//     (b,R) -> (Z,R)   unless R starts the next site                 SR[R] += w
//     (b,a) -> (Z,Z)   when R starts the next site                   adj   += w
// (w = the chunk's weight; no left pair if `a` starts a chunk, no right pair if R does or the
// stream ends).  SL / SR are vectors 0 and 1 of the replicated delta buffer; the table update
// turns them into the four vectors of format A (decL = incL = SL, decR = SR + adj at a,
// incR = SR + adj at Z).

struct AbArgs {
    uint32_t *b0, *b1;        // the two id buffers
    const SlotHdr *hdr_in;    // headers as they stand before this pass
    SlotHdr *hdr_out;         // dense: the other header array; sparse: unused
    StageRec *stage;          // sparse: staged headers, stage[t] for slot t ...
    uint32_t *smask;          // ... and [slot / 32]: which slots have one (no shared counter: every
                              // changed slot would queue up behind it, ~11 ns each)
    uint32_t T;
    DevState *st;
    uint32_t newid;
    uint32_t *delta;          // [replica][4][vcap]
    uint32_t vcap;            // row stride | log2(replicas) << 24
    uint32_t *idx;            // inverted index [IDX_H][istride] (word = slot / 32, bit = slot % 32), or nullptr
    uint32_t istride;
    const uint32_t *cand;     // sparse: the slots to visit (st->ncand of them, from k_select)
    uint32_t *removed;        // [256] ids removed by this pass, spread over counters (t & 255): one
                              // counter would serialise every changed slot of a dense pass (~11 ns each)
    uint32_t *dirty_n;        // reset here for the table update that follows
};

struct alignas(16) AbLds {
    uint32_t out[TILE];       // the compacted slot, staged for 16-byte stores (16-byte aligned)
    uint4 hdr[6];             // headers of slots t-1, t, t+1
    uint32_t ctx[8];          // slow path: halo0..2, prev2, prev1, index of the previous non-empty slot
    uint32_t wsum[MT / 64];
    uint32_t wmin[MT / 64];   // per wave: output offset of its first site
};

__global__ void __launch_bounds__(256)
k_slot2_init(SlotHdr *__restrict__ hdr, uint64_t T, DevState *st, int par, uint32_t which,
             const uint32_t *__restrict__ ids) {
    const uint64_t n = st->n[par];
    if (blockIdx.x == 0 && threadIdx.x == 0) st->gap = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += stride) {
        const uint64_t b0 = t * TILE;
        const uint32_t len = b0 >= n ? 0u : (uint32_t)min((uint64_t)TILE, n - b0);
        SlotHdr h;
        h.w0 = len > 0 ? ids[b0] : INVALID_WORD;
        h.w1 = len > 1 ? ids[b0 + 1] : INVALID_WORD;
        h.w2 = len > 2 ? ids[b0 + 2] : INVALID_WORD;
        h.meta = len | (which << 31);
        h.l0 = len > 1 ? ids[b0 + len - 2] : INVALID_WORD;
        h.l1 = len > 0 ? ids[b0 + len - 1] : INVALID_WORD;
        h.pad0 = h.pad1 = 0;
        hdr[t] = h;
    }
}

// the three words after slot t and the two before it, from headers only (any run of short or
// empty neighbours; rare)
__device__ __forceinline__ void slot_context_walk(const SlotHdr *__restrict__ hdr, uint32_t t, uint32_t T,
                                                  uint32_t *ctx) {
    uint32_t h[3] = {INVALID_WORD, INVALID_WORD, INVALID_WORD};
    uint32_t tn = 0xFFFFFFFFu;  // the next non-empty slot
    int got = 0;
    for (uint32_t u = t + 1; u < T && got < 3; u++) {
        const SlotHdr hh = hdr[u];
        const uint32_t lu = hh.meta & 0x7FFFFFFFu;
        const uint32_t ww[3] = {hh.w0, hh.w1, hh.w2};
        if (lu && tn == 0xFFFFFFFFu) tn = u;
        for (uint32_t i = 0; i < lu && i < 3 && got < 3; i++) h[got++] = ww[i];
    }
    uint32_t p1 = INVALID_WORD, p2 = INVALID_WORD, tp = 0xFFFFFFFFu;
    got = 0;
    for (uint32_t u = t; u-- > 0 && got < 2;) {
        const SlotHdr hh = hdr[u];
        const uint32_t lu = hh.meta & 0x7FFFFFFFu;
        if (lu == 0) continue;
        if (got == 0) {
            p1 = hh.l1;
            tp = u;
            got = 1;
            if (lu >= 2) {
                p2 = hh.l0;
                got = 2;
            }
        } else {
            p2 = hh.l1;
            got = 2;
        }
    }
    ctx[0] = h[0];
    ctx[1] = h[1];
    ctx[2] = h[2];
    ctx[3] = p2;
    ctx[4] = p1;
    ctx[5] = tp;
    ctx[6] = tn;
}

// One slot of an a != b pass, by one 256-thread workgroup.  Returns to the caller in every case
// (the sparse pass loops over slots); ends with all LDS reads of this slot done only after the
// caller's next __syncthreads().
// INDEXED: the inverted slot index is live and learns the pairs this pass creates (costs ~10
// VGPRs; the early dense passes, which run before the index exists, use the variant without).
template <bool SPARSE, bool INDEXED>
__device__ __forceinline__ void merge_ab_tile(AbLds &S, const uint32_t t, const AbArgs &A, const uint32_t a,
                                              const uint32_t b) {
    const int lane = lane_id(), wave = wave_id(), tid = threadIdx.x;
    const int wrel = wave * WAVE_SPAN;
    // ---- (1) every load that does not depend on another one -------------------------------
    // the slot itself, speculatively from buffer 0 (a slot lives in buffer 1 only between an
    // a == b pass that rewrote it and the next re-packing), all TILE words whatever its length
    const uint32_t *src = A.b0 + (size_t)t * TILE;
    uint4 rv[MJ];
    uint32_t rt[3], rh[2];  // words after / before this wave's span (wave-uniform addresses)
#pragma unroll
    for (int j = 0; j < MJ; j++) rv[j] = *reinterpret_cast<const uint4 *>(src + wrel + j * 256 + lane * 4);
#pragma unroll
    for (int i = 0; i < 3; i++) rt[i] = (wave < MT / 64 - 1) ? src[wrel + WAVE_SPAN + i] : 0u;
#pragma unroll
    for (int i = 0; i < 2; i++) rh[i] = (wave > 0) ? src[wrel - 2 + i] : 0u;
    if (tid < 6) {
        const long long hi = 2ll * (long long)t - 2 + tid;
        uint4 v = (tid & 1) ? make_uint4(INVALID_WORD, INVALID_WORD, 0u, 0u)
                            : make_uint4(INVALID_WORD, INVALID_WORD, INVALID_WORD, 0u);
        if (hi >= 0 && hi < 2ll * (long long)A.T) v = reinterpret_cast<const uint4 *>(A.hdr_in)[hi];
        S.hdr[tid] = v;
    }
    __syncthreads();
    const uint4 me0 = S.hdr[2], me1 = S.hdr[3];
    const uint32_t len = me0.w & 0x7FFFFFFFu, buf = me0.w >> 31;
    if (len == 0) {
        if (!SPARSE && tid == 0) {
            reinterpret_cast<uint4 *>(A.hdr_out)[2 * (size_t)t] = me0;
            reinterpret_cast<uint4 *>(A.hdr_out)[2 * (size_t)t + 1] = me1;
        }
        return;
    }
    if (buf) {  // (uniform) the slot lives in the other buffer: load again
        src = A.b1 + (size_t)t * TILE;
#pragma unroll
        for (int j = 0; j < MJ; j++) rv[j] = *reinterpret_cast<const uint4 *>(src + wrel + j * 256 + lane * 4);
#pragma unroll
        for (int i = 0; i < 3; i++) rt[i] = (wave < MT / 64 - 1) ? src[wrel + WAVE_SPAN + i] : 0u;
#pragma unroll
        for (int i = 0; i < 2; i++) rh[i] = (wave > 0) ? src[wrel - 2 + i] : 0u;
    }
    // ---- (2) context: three words after the slot, two before it ----------------------------
    uint32_t halo0 = S.hdr[4].x, halo1 = S.hdr[4].y, halo2 = S.hdr[4].z;
    uint32_t prev2 = S.hdr[1].x, prev1 = S.hdr[1].y;
    uint32_t tprev = t - 1;  // the slot that owns the word before mine (t == 0: none, and prev1 is invalid)
    uint32_t tnext = t + 1;  // ... and the words after mine
    {
        const uint32_t nlen = S.hdr[4].w & 0x7FFFFFFFu, plen = S.hdr[0].w & 0x7FFFFFFFu;
        if ((t + 1 < A.T && nlen < 3) || (t > 0 && plen < 2)) {  // (uniform, rare)
            if (tid == 0) slot_context_walk(A.hdr_in, t, A.T, S.ctx);
            __syncthreads();
            halo0 = S.ctx[0];
            halo1 = S.ctx[1];
            halo2 = S.ctx[2];
            prev2 = S.ctx[3];
            prev1 = S.ctx[4];
            tprev = S.ctx[5];
            tnext = S.ctx[6];
        }
    }
    // ---- (3) my words: positions >= len come from the halo, then nothing -------------------
    auto at = [&](int q, uint32_t own) -> uint32_t {
        const int d = q - (int)len;
        return d < 0 ? own : (d == 0 ? halo0 : (d == 1 ? halo1 : (d == 2 ? halo2 : INVALID_WORD)));
    };
    uint32_t x[MJ][4];
#pragma unroll
    for (int j = 0; j < MJ; j++) {
        const int q0 = wrel + j * 256 + lane * 4;
        x[j][0] = at(q0 + 0, rv[j].x);
        x[j][1] = at(q0 + 1, rv[j].y);
        x[j][2] = at(q0 + 2, rv[j].z);
        x[j][3] = at(q0 + 3, rv[j].w);
    }
    uint32_t tail[3], head[2];
#pragma unroll
    for (int i = 0; i < 3; i++) tail[i] = at(wrel + WAVE_SPAN + i, rt[i]);
    head[0] = wave > 0 ? at(wrel - 2, rh[0]) : prev2;
    head[1] = wave > 0 ? at(wrel - 1, rh[1]) : prev1;
    // ---- (4) r bits: r[q] = 1 iff (word[q], word[q+1]) is the pair; a != b, so m = r ----------
    uint32_t rb[MJ], valid[MJ];
    uint32_t anyr = 0;
#pragma unroll
    for (int j = 0; j < MJ; j++) {
        const uint32_t up = (j < MJ - 1) ? lane_first(x[(j + 1) % MJ][0]) : tail[0];
        const uint32_t nx = lane_next(x[j][0], up);
        uint32_t r = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t nxt = (k < 3) ? x[j][k + 1] : nx;
            r |= (uint32_t)(((x[j][k] & IDMASK) == a) & ((nxt & NWMASK) == b)) << k;
        }
        const int nb = (int)len - (wrel + j * 256 + lane * 4);
        valid[j] = nb >= 4 ? 0xFu : (nb <= 0 ? 0u : ((1u << nb) - 1u));
        rb[j] = r;
        anyr |= r & valid[j];
    }
    // carry: my first word is the `b` of a site that starts at the previous slot's last word
    const uint32_t s = (uint32_t)((prev1 != INVALID_WORD) & ((prev1 & IDMASK) == a) & ((me0.x & NWMASK) == b));
    const bool sites = __syncthreads_or((int)(anyr != 0)) != 0;
    if (!sites && !s) {
        // nothing in this slot changes and it owes no table update
        if (!SPARSE && tid == 0) {
            reinterpret_cast<uint4 *>(A.hdr_out)[2 * (size_t)t] = me0;
            reinterpret_cast<uint4 *>(A.hdr_out)[2 * (size_t)t + 1] = me1;
        }
        return;
    }
    // ---- (5) kept flags, output offsets -------------------------------------------------------
    uint32_t mb[MJ], kb[MJ], ex[MJ];
    const uint32_t rhead = wave > 0 ? (uint32_t)(((head[1] & IDMASK) == a) & ((x[0][0] & NWMASK) == b)) : s;
    uint32_t cnt[MJ];
#pragma unroll
    for (int j = 0; j < MJ; j++) {
        mb[j] = rb[j] & valid[j];
        // r of the position before my group: the previous lane's bit 3, the previous stripe's
        // last lane, or (first group of the wave) the word before the wave's span
        const uint32_t upr = (j > 0) ? ((lane_last(rb[(j + MJ - 1) % MJ]) >> 3) & 1u) : rhead;
        const uint32_t mprev = (uint32_t)dpp_mov<0x138>((int)upr, (int)((rb[j] >> 3) & 1u));  // wave_shr:1
        const uint32_t mp = (lane == 0) ? upr : mprev;
        kb[j] = ~((mb[j] << 1) | mp) & valid[j] & 0xFu;
        cnt[j] = (uint32_t)__popc(kb[j]);
    }
    static_assert(MJ == 4, "packed scan below assumes four stripes");
    const uint32_t i01 = wave_iscan_add(cnt[0] | (cnt[1] << 16));
    const uint32_t i23 = wave_iscan_add(cnt[2] | (cnt[3] << 16));
    const uint32_t t01 = lane_last(i01), t23 = lane_last(i23);
    const uint32_t tot0 = t01 & 0xFFFFu, tot1 = t01 >> 16, tot2 = t23 & 0xFFFFu, tot3 = t23 >> 16;
    ex[0] = (i01 & 0xFFFFu) - cnt[0];
    ex[1] = tot0 + (i01 >> 16) - cnt[1];
    ex[2] = tot0 + tot1 + (i23 & 0xFFFFu) - cnt[2];
    ex[3] = tot0 + tot1 + tot2 + (i23 >> 16) - cnt[3];
    if (lane == 0) S.wsum[wave] = tot0 + tot1 + tot2 + tot3;
    __syncthreads();
    uint32_t wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < MT / 64; w++) {
        const uint32_t v = S.wsum[w];
        if (w < wave) wbase += v;
        total += v;
    }
    // ---- (6) stage the compacted slot in LDS, then 16-byte stores back to its home -------------
    // (everything before the first site keeps its place and value: only the rest is stored)
    {
        uint32_t fc = 0x7FFFFFFFu;
#pragma unroll
        for (int j = 0; j < MJ; j++) {
            if (mb[j]) {
                const uint32_t k0 = (uint32_t)__ffs((int)mb[j]) - 1u;
                fc = min(fc, wbase + ex[j] + (uint32_t)__popc(kb[j] & ((1u << k0) - 1u)));
            }
        }
        fc = (uint32_t)wave_min_i32((int)fc);
        if (lane == 0) S.wmin[wave] = fc;
    }
#pragma unroll
    for (int j = 0; j < MJ; j++) {
        uint32_t o = wbase + ex[j];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if ((kb[j] >> k) & 1u) {
                const uint32_t w = x[j][k];
                S.out[o++] = ((mb[j] >> k) & 1u) ? (A.newid | (w & (FLAG | WMASK))) : w;
            }
        }
    }
    __syncthreads();
    {
        uint32_t *dst = (buf ? A.b1 : A.b0) + (size_t)t * TILE;
        uint32_t first = 0;  // a dropped first word moves everything
        if (!s) first = min(min(S.wmin[0], S.wmin[1]), min(S.wmin[2], S.wmin[3])) & ~3u;
        for (uint32_t i = first + (uint32_t)tid * 4; i < total; i += MT * 4)
            *reinterpret_cast<uint4 *>(dst + i) = *reinterpret_cast<const uint4 *>(&S.out[i]);
    }
    if (tid == 0) {
        uint32_t h[8];
        h[0] = total > 0 ? S.out[0] : INVALID_WORD;
        h[1] = total > 1 ? S.out[1] : INVALID_WORD;
        h[2] = total > 2 ? S.out[2] : INVALID_WORD;
        h[3] = total | (buf << 31);
        h[4] = total > 1 ? S.out[total - 2] : INVALID_WORD;
        h[5] = total > 0 ? S.out[total - 1] : INVALID_WORD;
        h[6] = h[7] = 0;
        if (!SPARSE) {
            reinterpret_cast<uint4 *>(A.hdr_out)[2 * (size_t)t] = make_uint4(h[0], h[1], h[2], h[3]);
            reinterpret_cast<uint4 *>(A.hdr_out)[2 * (size_t)t + 1] = make_uint4(h[4], h[5], 0u, 0u);
        } else {
            StageRec *r = A.stage + t;
            r->t = t;
#pragma unroll
            for (int i = 0; i < 8; i++) r->h[i] = h[i];
            atomicOr(&A.smask[t >> 5], 1u << (t & 31));
        }
        atomicAdd(&A.removed[t & 255u], len - total);
        if (total < 3 && t + 1 < A.T) A.st->gap = 1;
    }
    // ---- (7) pair-table delta of my sites (format B); their new pairs enter the index -----------
    if (!sites) return;  // carry only: the site belongs to the previous slot
    if (!A.delta) return;  // (experiment "exp_no_delta": time the pass without its table bookkeeping; results are wrong)
    const uint32_t nrep = 1u << (A.vcap >> 24);
    const uint32_t vc = A.vcap & 0xFFFFFFu;
    uint32_t *dl = A.delta + (size_t)(t & (nrep - 1)) * 4 * vc;  // SL
    uint32_t *dr = dl + vc;                                                        // SR
    uint32_t adj = 0;
#pragma unroll
    for (int j = 0; j < MJ; j++) {
        if (!__any(mb[j] != 0)) continue;  // (uniform per wave) no site in this stripe
        // two words before my group, three after it
        uint32_t upm2, upm1, dn0, dn1, dn2;
        if (j > 0) {
            upm2 = lane_last(x[(j + MJ - 1) % MJ][2]);
            upm1 = lane_last(x[(j + MJ - 1) % MJ][3]);
        } else {
            upm2 = head[0];
            upm1 = head[1];
        }
        if (j < MJ - 1) {
            dn0 = lane_first(x[(j + 1) % MJ][0]);
            dn1 = lane_first(x[(j + 1) % MJ][1]);
            dn2 = lane_first(x[(j + 1) % MJ][2]);
        } else {
            dn0 = tail[0];
            dn1 = tail[1];
            dn2 = tail[2];
        }
        uint32_t W[9];
        {
            const uint32_t pm2 = (uint32_t)dpp_mov<0x138>((int)upm2, (int)x[j][2]);
            const uint32_t pm1 = (uint32_t)dpp_mov<0x138>((int)upm1, (int)x[j][3]);
            W[0] = lane == 0 ? upm2 : pm2;
            W[1] = lane == 0 ? upm1 : pm1;
        }
        W[2] = x[j][0];
        W[3] = x[j][1];
        W[4] = x[j][2];
        W[5] = x[j][3];
        W[6] = lane_next(x[j][0], dn0);
        W[7] = lane_next(x[j][1], dn1);
        W[8] = lane_next(x[j][2], dn2);
        if (mb[j] == 0) continue;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (!((mb[j] >> k) & 1u)) continue;
            const uint32_t wa = W[k + 2];
            const uint32_t wt = word_weight(wa);
            const uint32_t L = W[k + 1], LL = W[k];
            if (!(wa & FLAG) && L != INVALID_WORD) {
                const bool ltail = ((LL & IDMASK) == a) & ((L & NWMASK) == b);
                if (!ltail) {
                    atomicAdd(&dl[L & IDMASK], wt);
                    // the new pair (L, Z) enters the filter of the slot that holds L, and mine too if
                    // that is another slot (a boundary pair is known to both slots it touches: the
                    // one that owns its site and the one that drops the site's second word)
                    if (INDEXED) {
                        index_add(A.idx, A.istride, t, L & IDMASK, A.newid);
                        if (wrel + j * 256 + lane * 4 + k == 0) index_add(A.idx, A.istride, tprev, L & IDMASK, A.newid);
                    }
                }
            }
            const uint32_t R = W[k + 4], RR = W[k + 5];
            if (!(R & FLAG)) {  // (INVALID_WORD has the flag bit set: end of stream)
                const bool rsite = ((R & IDMASK) == a) & ((RR & NWMASK) == b);
                if (rsite) adj += wt;
                else atomicAdd(&dr[R & IDMASK], wt);
                if (INDEXED) {
                    const uint32_t y = rsite ? A.newid : (R & IDMASK);
                    index_add(A.idx, A.istride, t, A.newid, y);
                    if (wrel + j * 256 + lane * 4 + k + 2 >= (int)len) index_add(A.idx, A.istride, tnext, A.newid, y);
                }
            }
        }
    }
    if (__any(adj != 0)) {
        adj = wave_sum_u32(adj);
        if (lane == 0) atomicAdd(&A.st->adj, adj);
    }
}

// dense a != b pass: one workgroup per slot
template <bool INDEXED>
__global__ void __launch_bounds__(MT, INDEXED ? 5 : 7)
k_merge_ab_dense(AbArgs A) {
    __shared__ AbLds S;
    const DevState *st = A.st;
    if (blockIdx.x == 0 && threadIdx.x == 0) *A.dirty_n = 0;
    if (blockIdx.x >= A.T || st->status) return;
    if (!st->found) {
        if (blockIdx.x == 0 && threadIdx.x == 0) A.st->status = ST_INTERNAL;
        return;
    }
    const uint32_t a = (uint32_t)st->a, b = (uint32_t)st->b;
    if (a == b) return;  // k_merge_aa's pass
    merge_ab_tile<false, INDEXED>(S, blockIdx.x, A, a, b);
}

// sparse a != b pass: a resident grid works through the candidate list that the deciding block of
// k_select made from the index (st->ncand slots in A.cand: every slot whose filter admits the
// pair, plus the slots an a == b pass rewrote since the index was built; all slots while
// st->gap is up) -- evenly dealt, so the pass ends when ceil(ncand / grid) tiles are done.
#ifndef AB_SPARSE_WAVES
#define AB_SPARSE_WAVES 4
#endif
__global__ void __launch_bounds__(MT, AB_SPARSE_WAVES)
k_merge_ab_sparse(AbArgs A) {
    __shared__ AbLds S;
    const DevState *st = A.st;
    if (blockIdx.x == 0 && threadIdx.x == 0) *A.dirty_n = 0;
    if (st->status) return;
    if (!st->found) {
        if (blockIdx.x == 0 && threadIdx.x == 0) A.st->status = ST_INTERNAL;
        return;
    }
    const uint32_t a = (uint32_t)st->a, b = (uint32_t)st->b;
    if (a == b) return;
    const uint32_t n = st->ncand;
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        merge_ab_tile<true, true>(S, A.cand[i], A, a, b);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// a == b pass (runs only when the decided pair has a == b; a resident grid striding over ALL
// slots).  The first form's algorithm (k_slots.hip: run-parity carry found by walking the run
// back through the previous slots, out-of-place rewrite, format A delta, every pair charged to
// its left element) on the 32-byte headers.
struct AaArgs {
    const uint32_t *b0, *b1;
    uint32_t *w0, *w1;
    const SlotHdr *hdr_in;
    SlotHdr *hdr_out;         // every slot's header is written here ...
    StageRec *stage;          // ... unless this is set (the a != b kernel of this iteration is the
                              // sparse one, whose staged headers get committed): changed slots only
    uint32_t *smask;
    uint32_t T;
    DevState *st;
    uint32_t newid;
    uint32_t *delta;
    uint32_t vcap;
    unsigned long long *sdesc;
    uint32_t epoch;
    uint32_t *dirty;          // index live: [slot / 32], set for the slots this pass rewrites
    uint32_t *removed;        // [256]
};

__device__ __forceinline__ void merge_aa_tile(const uint32_t t, const AaArgs &A, const uint32_t a) {
    __shared__ int s_wave[MT / 64];
    __shared__ uint32_t s_wsum[MT / 64];
    __shared__ uint32_t s_ctx[8];   // halo0..2, prev2, prev1, carry
    __shared__ uint32_t s_hdr[8];   // my header (copied, then rewritten by the tile)
    const uint32_t b = a;
    const SlotHdr hme = A.hdr_in[t];
    const uint32_t mi = hme.meta;
    const int len = (int)(mi & 0x7FFFFFFFu);
    uint4 *out4 = reinterpret_cast<uint4 *>(A.hdr_out + t);
    auto keep_header = [&]() {  // (thread 0) the slot stays as it is
        if (!A.stage) {
            out4[0] = make_uint4(hme.w0, hme.w1, hme.w2, hme.meta);
            out4[1] = make_uint4(hme.l0, hme.l1, 0u, 0u);
        }
    };
    if (len == 0) {
        if (threadIdx.x == 0) keep_header();
        return;
    }
    const uint32_t cur = mi >> 31;
    const uint32_t *src = (cur ? A.b1 : A.b0) + (size_t)t * TILE;
    if (threadIdx.x == 0) {
        slot_context_walk(A.hdr_in, t, A.T, s_ctx);
        s_hdr[0] = hme.w0;
        s_hdr[1] = hme.w1;
        s_hdr[2] = hme.w2;
        s_hdr[3] = hme.meta;
        s_hdr[4] = hme.l0;
        s_hdr[5] = hme.l1;
    }
    SlotRaw raw;
    slot_raw_load(raw, src, len);
    __syncthreads();
    const uint32_t halo[3] = {s_ctx[0], s_ctx[1], s_ctx[2]};
    const uint32_t prev = s_ctx[4];
    const uint32_t s_first_word = hme.w0;
    Tile tl;
    tile_from_slot(tl, raw, len, halo);
    tile_rbits(tl, a, b);
    uint32_t s = (uint32_t)((prev != INVALID_WORD) & ((prev & IDMASK) == a) & ((s_first_word & NWMASK) == b));
    {
        // the carry is the PARITY of the run of a's that ends at the previous slot's last id (F2).
        // Walk that run backwards, 64 ids per step; only if it swallows the whole previous slot does
        // this tile need that slot's own carry (published below by every tile of the pass; every
        // workgroup of the resident grid takes its slots in ascending order, so the wait ends).
        const unsigned long long tag = ((unsigned long long)(A.epoch & EPOCH_MASK)) << 42;
        if (wave_id() == 0) {
            const int lane = lane_id();
            uint32_t sc = 0;
            bool failed = false;
            if (s) {  // the boundary pair matches: r[last of previous slot] = 1
                uint32_t u = t;
                uint32_t mu = 0;
                while (u-- > 0) {
                    mu = A.hdr_in[u].meta;
                    if (mu & 0x7FFFFFFFu) break;
                }
                const int lu = (int)(mu & 0x7FFFFFFFu);
                const uint32_t *pu = ((mu >> 31) ? A.b1 : A.b0) + (size_t)u * TILE;
                int ones = 0;       // r-ones counted so far, walking back from the last id
                bool open = true;   // no zero met yet
                uint32_t nextw = s_first_word;  // the word after the current position
                for (int base = lu - 1; base >= 0 && open; base -= 64) {
                    const int q = base - lane;
                    const uint32_t xq = (q >= 0) ? pu[q] : INVALID_WORD;
                    uint32_t nx = (uint32_t)__shfl_up((int)xq, 1);
                    if (lane == 0) nx = nextw;
                    const bool r = (q >= 0) && ((xq & IDMASK) == a) && ((nx & NWMASK) == a);
                    const unsigned long long zeros = __ballot(!r);
                    if (zeros) {
                        ones += __ffsll((long long)zeros) - 1;
                        // a zero caused by running off the slot (q < 0) means the whole slot is ones
                        const int zl = __ffsll((long long)zeros) - 1;
                        open = (base - zl < 0);
                        break;
                    }
                    ones += 64;
                    nextw = (uint32_t)__shfl((int)xq, 63);
                }
                if (!open || ones < lu) {
                    sc = (uint32_t)(ones & 1);  // m[last] = r[last] & (run length odd)
                } else {
                    // the whole previous slot is one run: m[q] = (q even) ^ its carry
                    uint32_t su = 0;
                    bool got = false;
                    for (uint32_t spins = 0; spins < LOOKBACK_SPINS; spins++) {
                        const unsigned long long d = desc_load(&A.sdesc[u]);
                        if ((d >> 42) == (tag >> 42) + (1ull << 20)) {  // status bit above the epoch
                            su = (uint32_t)(d & 1u);
                            got = true;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(8);
                    }
                    failed = !got;
                    sc = (uint32_t)(((lu - 1) & 1) == 0) ^ su;
                }
            }
            if (lane == 0) {
                s_ctx[5] = sc;
                desc_store(&A.sdesc[t], tag | (1ull << 62) | sc);
                if (failed) atomicExch(&A.st->status, ST_LOOKBACK);
            }
        }
        __syncthreads();
        s = s_ctx[5];
    }
    // fast path: no match at any owned position, none at the first word after the slot, no carry
    {
        uint32_t anyr = s;
        const int qw = wave_id() * WAVE_SPAN + lane_id() * 4;
#pragma unroll
        for (int j = 0; j < MJ; j++) {
            const int q0 = qw + j * 256;
            const int nb = len + 1 - q0;  // keep the bits of positions q <= len
            const uint32_t keep = nb >= 4 ? 0xFu : (nb <= 0 ? 0u : ((1u << nb) - 1u));
            anyr |= tl.rb[j] & keep;
        }
        if (len == TILE && wave_id() == MT / 64 - 1)
            anyr |= (uint32_t)(((tl.tail[0] & IDMASK) == a) & ((tl.tail[1] & NWMASK) == b));
        if (!__syncthreads_or((int)(anyr != 0))) {
            if (threadIdx.x == 0) keep_header();
            return;
        }
    }
    tile_lzscan(tl, s_wave);
    uint32_t kept = 0;
    bool changed = false;
    uint32_t *dst = (cur ? A.w0 : A.w1) + (size_t)t * TILE;  // the OTHER buffer
    tile_rewrite<true, true, 1>(tl, s, a, b, A.newid, dst, s_wsum, A.delta, A.vcap, len, &kept, &changed, s_hdr);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (changed) {
            for (uint32_t i = kept; i < 3; i++) s_hdr[i] = INVALID_WORD;  // fewer than 3 ids left
            if (kept < 2) s_hdr[4] = INVALID_WORD;
            if (kept < 1) s_hdr[5] = INVALID_WORD;
            const uint32_t meta = kept | ((cur ^ 1u) << 31);
            if (!A.stage) {
                out4[0] = make_uint4(s_hdr[0], s_hdr[1], s_hdr[2], meta);
                out4[1] = make_uint4(s_hdr[4], s_hdr[5], 0u, 0u);
            } else {
                StageRec *r = A.stage + t;
                atomicOr(&A.smask[t >> 5], 1u << (t & 31));
                r->t = t;
                r->h[0] = s_hdr[0];
                r->h[1] = s_hdr[1];
                r->h[2] = s_hdr[2];
                r->h[3] = meta;
                r->h[4] = s_hdr[4];
                r->h[5] = s_hdr[5];
                r->h[6] = r->h[7] = 0;
            }
            atomicAdd(&A.removed[t & 255u], (uint32_t)len - kept);
            if (kept < 3 && t + 1 < A.T) A.st->gap = 1;
            if (A.dirty) {
                // its pairs changed, and so did the boundary pairs it shares with both neighbours:
                // none of that is in the index until the next build
                atomicOr(&A.dirty[t >> 5], 1u << (t & 31));
                if (t > 0) atomicOr(&A.dirty[(t - 1) >> 5], 1u << ((t - 1) & 31));
                if (t + 1 < A.T) atomicOr(&A.dirty[(t + 1) >> 5], 1u << ((t + 1) & 31));
            }
        } else {
            keep_header();
        }
    }
}

__global__ void __launch_bounds__(MT)
k_merge_aa(AaArgs A) {
    const DevState *st = A.st;
    // (one thread reads the status for the whole workgroup: another workgroup may raise it while
    // this one starts, and a barrier-laden tile must be entered by all waves or by none)
    __shared__ uint32_t s_run;
    if (threadIdx.x == 0) s_run = (st->status == 0 && st->found != 0);  // (a missing decision is reported by the a != b kernel)
    __syncthreads();
    if (!s_run) return;
    const uint32_t a = (uint32_t)st->a;
    if (a != (uint32_t)st->b) return;
    for (uint32_t t = blockIdx.x; t < A.T; t += gridDim.x) {
        merge_aa_tile(t, A, a);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// Index build: one workgroup per group of 32 slots, the group's IDX_H x 32-bit filter in LDS
// (128 KiB), three ds_or per pair, then one coalesced write of the group's row into a
// group-major scratch image; k_index_transpose turns that into the bucket-major index.  No global
// atomics.  Also clears the group's `dirty` word.
__global__ void __launch_bounds__(1024)
k_index_build(const uint32_t *__restrict__ b0, const uint32_t *__restrict__ b1, const SlotHdr *__restrict__ hdr,
              uint32_t T, uint32_t *__restrict__ idx, uint32_t *__restrict__ dirty) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_mask[];
    const uint32_t g = blockIdx.x;
    for (uint32_t i = threadIdx.x; i < IDX_H; i += 1024) s_mask[i] = 0;
    if (threadIdx.x == 0) dirty[g] = 0;
    __syncthreads();
    for (uint32_t sl = 0; sl < 32; sl++) {
        const uint32_t t = g * 32 + sl;
        if (t >= T) break;
        const uint32_t m = hdr[t].meta;
        const uint32_t len = m & 0x7FFFFFFFu;
        const uint32_t *src = ((m >> 31) ? b1 : b0) + (size_t)t * TILE;
        const uint32_t q = threadIdx.x * 4;
        if (q >= len) continue;
        const uint4 v = *reinterpret_cast<const uint4 *>(src + q);
        uint32_t x[5] = {v.x, v.y, v.z, v.w, INVALID_WORD};
        if (q + 4 < len) {
            x[4] = src[q + 4];
        } else {
            // the word after the slot: first word of the next non-empty slot (headers)
            uint32_t nxt = INVALID_WORD;
            for (uint32_t u = t + 1; u < T; u++) {
                if (hdr[u].meta & 0x7FFFFFFFu) {
                    nxt = hdr[u].w0;
                    break;
                }
            }
            const uint32_t d = len - q;  // 1..4 valid words here; word d is `nxt`
#pragma unroll
            for (int k = 1; k <= 4; k++)
                if ((uint32_t)k == d) x[k] = nxt;
        }
        const uint32_t bit = 1u << sl;
        if (q == 0 && t > 0 && !(x[0] & FLAG)) {
            // the pair that ends at my first word: it belongs to the slot before, but a merge of it
            // removes my first word, so my filter knows it too
            uint32_t pw = INVALID_WORD;
            for (uint32_t u = t; u-- > 0;) {
                if (hdr[u].meta & 0x7FFFFFFFu) {
                    pw = hdr[u].l1;
                    break;
                }
            }
            if (pw != INVALID_WORD) {
                uint32_t h1, h2, h3;
                pair_hash(pw & IDMASK, x[0] & IDMASK, h1, h2, h3);
                atomicOr(&s_mask[h1], bit);
                atomicOr(&s_mask[h2], bit);
                atomicOr(&s_mask[h3], bit);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (q + k < len && !(x[k + 1] & FLAG)) {
                uint32_t h1, h2, h3;
                pair_hash(x[k] & IDMASK, x[k + 1] & IDMASK, h1, h2, h3);
                atomicOr(&s_mask[h1], bit);
                atomicOr(&s_mask[h2], bit);
                atomicOr(&s_mask[h3], bit);
            }
        }
    }
    __syncthreads();
    uint32_t *row = idx + (size_t)g * IDX_H;
    for (uint32_t i = threadIdx.x; i < IDX_H; i += 1024) row[i] = s_mask[i];
}

// scratch[g][h] -> idx[h][g]  (32 x 32 tiles through LDS, both sides coalesced)
__global__ void __launch_bounds__(256)
k_index_transpose(const uint32_t *__restrict__ scratch, uint32_t ngroups, uint32_t *__restrict__ idx,
                  uint32_t stride) {
    __shared__ uint32_t tile[32][33];
    const uint32_t g0 = blockIdx.x * 32, h0 = blockIdx.y * 32;
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t g = g0 + ty + 8 * i;
        tile[ty + 8 * i][tx] = (g < ngroups) ? scratch[(size_t)g * IDX_H + h0 + tx] : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t h = h0 + ty + 8 * i, g = g0 + tx;
        if (g < ngroups) idx[(size_t)h * stride + g] = tile[tx][ty + 8 * i];
    }
}

// ---------------------------------------------------------------------------
// re-packing: slots -> contiguous
__global__ void __launch_bounds__(256)
k_slot2_lens(const SlotHdr *__restrict__ hdr, uint64_t T, uint32_t *__restrict__ lens) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += stride)
        lens[t] = hdr[t].meta & 0x7FFFFFFFu;
}
__global__ void __launch_bounds__(256)
k_slot2_compact(const uint32_t *__restrict__ b0, const uint32_t *__restrict__ b1,
                const SlotHdr *__restrict__ hdr, const unsigned long long *__restrict__ off,
                uint32_t *__restrict__ out) {
    const uint64_t t = blockIdx.x;
    const uint32_t m = hdr[t].meta;
    const uint32_t len = m & 0x7FFFFFFFu;
    const uint32_t *src = ((m >> 31) ? b1 : b0) + t * TILE;
    uint32_t *dst = out + off[t];
    for (uint32_t i = threadIdx.x; i < len; i += 256) dst[i] = src[i];
}

}  // namespace bpe
