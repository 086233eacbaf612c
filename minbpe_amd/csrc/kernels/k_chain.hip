// k_chain.hip -- chain steps: the lean iteration (k_lean.hip) taken two steps further.
//
// Late in training almost every selection ends in a tie (93-97 % of them after merge 20,000 of the 1 GB
// run), and a lean iteration is nothing but launches and dependent round trips (~25 us, whatever the pair).
// Two facts about the reference's loop (get_stats -> max -> merge, base.py:13-41, basic.py:31-42,
// regex.py:49-63) let one step do the work of several iterations, exactly:
//
//  (1) THE LIST.  Let M be the maximum count and T the pairs that attain it, in order of first occurrence
//      (= dict order = the order max() breaks ties in).  A merge of (a,b) -> Z changes the counts of pairs
//      that share a token with it and creates pairs with Z; nothing can rise above M, and a created pair
//      reaches M only by taking over EVERY occurrence of a listed pair (L,a) or (b,R) -- it then stands
//      exactly where that pair stood in the order.  So after any merges the set of pairs at M, in order,
//      is the old list with some entries dropped and some replaced in place.  A listed pair (x,y) can only
//      have become one of (x,y), (Zx,y), (x,Zy), (Zx,Zy) (Zx: the token made from a merged pair ending in
//      x, Zy: from one starting with y): the one of the four whose count in the updated table is M takes
//      its place, none -> it leaves the list.  ONE full selection per level M, then table look-ups until
//      the list is empty.  (tests/test_list_model.py: CPU model against the reference semantics.)
//  (2) THE BATCH.  The longest prefix of the list whose pairs have a != b and share no token is what the
//      reference merges next, in that order, whatever those merges create (a pair that shares a token with
//      a merged one is not in the prefix, so whatever takes its place comes after the prefix).  (Round 6: the
//      prefix may share SECOND tokens -- (a, b), (c, b): only a pair that could chain onto a site of the batch,
//      x a second token or y a first one, must stay out; k_pool.hip.  First tokens stay distinct: the pass
//      below looks a pair up by its first token.)  Their
//      rewrites commute (no two sites overlap): ONE pass over the candidate slots merges them all, and
//      charges every site its pair-table delta as the sequential merges would have
//      (delta format B per pair j: a left neighbour that ends a site of pair i < j already reads Z_i, one of
//      pair i > j still reads b_i; same on the right; i == j is format B's adj).
//
// A step is three launches, like a lean iteration:
//   k_pool_sel     (k_pool.hip) the list generalised to every pair at or above a threshold, kept exact across steps: the
//                  next batch off the pool; a rebuild re-scans the flagged rows with workgroups 1...  (Rounds 4-5 kept
//                  the list itself -- k_chain_sel, LIST / FULL modes, one count level at a time; retired in round 6: the
//                  rule is pinned by tests/test_list_model.py and tests/test_level_model.py, the pool by
//                  tests/test_pool_model.py.)
//   k_merge_chain  every workgroup lists its candidate slots (the filter rows of ALL the batch's pairs) and
//                  its waves rewrite them (merge_chain_wave).
//   k_apply_chain  table update for every pair of the batch, one token per thread; rows a_j, b_j, Z_j and the
//                  rows whose maximum may have dropped are flagged for the next FULL selection; the
//                  iteration records of all k merges, the stream length, st->iter += k, the next step's mode.
// The device counts the merges (st->iter): the host does not know how many a step will do.  What a step cannot
// settle (more than TIE_CAP tied pairs, short slots about, a == b at the head of the list) is deferred to the
// general path exactly as a lean iteration's is.
// Part of bpe_kernels.hip, which includes the parts in order.
// (no include guard: bpe_kernels.hip includes this part once per geometry, namespace BPE_G)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../bpe_device.h"

namespace bpe {
namespace BPE_G {

constexpr uint32_t CH_EX_CAP = 2048;  // flagged rows one FULL selection takes from the re-scanning workgroups

// index of the batch pair that (w0, w1) is a site of, or -1.  A word that starts a chunk carries its flag into
// the comparison (no pair spans chunks); INVALID_WORD matches no id.
__device__ __forceinline__ int chain_match(const uint32_t *pa, const uint32_t *pb, uint32_t K, uint32_t w0, uint32_t w1) {
    const uint32_t x = w0 & IDMASK, y = w1 & NWMASK;
    int hit = -1;
    for (uint32_t p = 0; p < K; p++)
        if (x == pa[p] && y == pb[p]) hit = (int)p;
    return (w0 == INVALID_WORD) ? -1 : hit;
}

// One slot of a chain step's merge pass, by ONE WAVE: merge_ab_wave (k_slots2.hip) for K >= 2 pairs with distinct first
// tokens, none chaining onto another (second tokens may be shared), at once.  DENSE == false: the sparse form (staged headers, global delta replicas, index live).  DENSE == true: the
// early passes, where every slot is visited and a pair has millions of sites: headers go to the other header array,
// the delta into the workgroup's LDS tables sd (per pair p, at sd + p * CH_SD: SL[CH_DCAP] | SR[CH_DCAP] | adj |
// ids removed; every id is below CH_DCAP), flushed by the kernel when its slots are done.  pa / pb: the pairs (LDS), pb1[p + 1] = pb[p] with
// pb1[0] a word that matches nothing, z0: pair p becomes z0 + p.
// (the tables of a DENSE chain step cover ids below CH_DCAP, not LDSD_CAP: the dense phase ends when the index comes up --
// merge ~300 of a 1 GB stream -- and a table of 1024 ids lets eight pairs' tables fit beside the sixteen waves' staging,
// where 1920 ids allowed four; a stream whose dense phase outlasts id 1024 goes on with the general path's single merges)
constexpr uint32_t CH_DCAP = 1024;
constexpr uint32_t CH_SD = 2 * CH_DCAP + 2;  // words of one pair's LDS delta tables
// the loads of a slot that depend on nothing but its number: its words (speculatively from buffer 0, where a slot lives
// unless an a == b pass moved it) and the headers of slots t - 1, t, t + 1 as six 16-byte pieces on lanes 0..5
__device__ __forceinline__ void chain_slot_load(const AbArgs &A, const uint32_t t, uint4 (&rv)[MJ], uint4 &hv) {
    const int lane = lane_id();
    const uint32_t *src = A.b0 + (size_t)t * TILE2;
#pragma unroll
    for (int j = 0; j < MJ; j++) rv[j] = *reinterpret_cast<const uint4 *>(src + j * 256 + lane * 4);
    hv = (lane & 1) ? make_uint4(INVALID_WORD, INVALID_WORD, 0u, 0u) : make_uint4(INVALID_WORD, INVALID_WORD, INVALID_WORD, 0u);
    const long long hi = 2ll * (long long)t - 2 + lane;
    if (lane < 6 && hi >= 0 && hi < 2ll * (long long)A.T) hv = reinterpret_cast<const uint4 *>(A.hdr_in)[hi];
}
// (rv / hv: the slot's loads, issued by the caller -- k_merge_chain issues the NEXT slot's before it works on this one)
template <bool DENSE, bool THROUGH = false>
__device__ __forceinline__ void merge_chain_wave(uint32_t *__restrict__ out, uint32_t *__restrict__ sd, const uint32_t t, const AbArgs &A,
                                                 const uint32_t *pa, const uint32_t *pb, const uint32_t *pb1,
                                                 const uint32_t K, const uint32_t z0, const uint32_t brep,
                                                 uint4 (&rv)[MJ], const uint4 hv,
                                                 const uint32_t *ph = nullptr, const uint32_t hm = 0,
                                                 const uint32_t tl_in = 0xFFFFFFFFu) {
    const int lane = lane_id();
    // (tl_in: the caller read st->tlive once -- the kernel stores to *st, so a load here is repeated for every slot, a
    // dependent round trip at the head of each)
    const uint32_t Tl = tl_in != 0xFFFFFFFFu ? tl_in : min(A.T, A.st->tlive);
    const uint32_t *src;
    const uint32_t meta = bcast(hv.w, 2);
    const uint32_t len = meta & 0x7FFFFFFFu, buf = meta >> 31;
    auto keep_header = [&]() {  // dense: the slot stays as it is (lanes 2 and 3 hold its header)
        if (DENSE && (lane == 2 || lane == 3)) reinterpret_cast<uint4 *>(A.hdr_out)[2 * (size_t)t + (lane - 2)] = hv;
    };
    if (len == 0) {
        keep_header();
        return;
    }
    if (buf) {  // (uniform) the slot lives in the other buffer: load again
        src = A.b1 + (size_t)t * TILE2;
#pragma unroll
        for (int j = 0; j < MJ; j++) rv[j] = *reinterpret_cast<const uint4 *>(src + j * 256 + lane * 4);
    }
    // ---- (2) context: three words after the slot, two before it ----------------------------
    const uint32_t first = bcast(hv.x, 2);
    uint32_t halo0 = bcast(hv.x, 4), halo1 = bcast(hv.y, 4), halo2 = bcast(hv.z, 4);
    uint32_t prev2 = bcast(hv.x, 1), prev1 = bcast(hv.y, 1);
    uint32_t tprev = t - 1;
    uint32_t tnext = t + 1;
    {
        const uint32_t nlen = bcast(hv.w, 4) & 0x7FFFFFFFu, plen = bcast(hv.w, 0) & 0x7FFFFFFFu;
        if ((t + 1 < Tl && nlen < 3) || (t > 0 && plen < 2)) {  // (uniform, rare)
            uint32_t ctx[7] = {0, 0, 0, 0, 0, 0, 0};
            if (lane == 0) slot_context_walk(A.hdr_in, t, Tl, ctx);
            halo0 = bcast(ctx[0], 0);
            halo1 = bcast(ctx[1], 0);
            halo2 = bcast(ctx[2], 0);
            prev2 = bcast(ctx[3], 0);
            prev1 = bcast(ctx[4], 0);
            tprev = bcast(ctx[5], 0);
            tnext = bcast(ctx[6], 0);
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
        const int q0 = j * 256 + lane * 4;
        if ((j + 1) * 256 <= (int)len) {  // (uniform) a stripe of own words: nothing to select
            x[j][0] = rv[j].x;
            x[j][1] = rv[j].y;
            x[j][2] = rv[j].z;
            x[j][3] = rv[j].w;
        } else {
            x[j][0] = at(q0 + 0, rv[j].x);
            x[j][1] = at(q0 + 1, rv[j].y);
            x[j][2] = at(q0 + 2, rv[j].z);
            x[j][3] = at(q0 + 3, rv[j].w);
        }
    }
    uint32_t tail[3];
#pragma unroll
    for (int i = 0; i < 3; i++) tail[i] = at(TILE2 + i, 0u);
    // ---- (4) r bits and, per site, WHICH pair (a nibble per word: pair index + 1) -------------
    uint32_t rb[MJ], jc[MJ], valid[MJ];
#pragma unroll
    for (int j = 0; j < MJ; j++) {
        rb[j] = 0;
        jc[j] = 0;
    }
    // Two stages, so that the work per word does not grow with K twice over (four waves share a SIMD: with
    // every word compared against every pair the pass was bound by instruction issue, not by the slot's
    // latency): which pair's FIRST token is this word (K compares), then ONE compare of the next word with that
    // pair's second token, fetched from LDS by pair number (pb1[0] matches nothing: no pair).
    uint32_t s = 0;  // carry: my first word is the second word of a site that starts at the previous slot's last word
    uint32_t ia[MJ];  // a nibble per word: the pair whose first token it is, + 1 (tokens are distinct: at most one)
#pragma unroll
    for (int j = 0; j < MJ; j++) ia[j] = 0;
    uint32_t ip = 0;
    if (hm) {
        // a batch of five pairs or more: ONE look-up per word instead of K compares -- ph is a 256-entry table in LDS,
        // (id * hm >> 8) & 255 is free of collisions among the batch's first tokens (chain_hash_build), an entry is
        // id << 8 | pair number + 1
        auto code = [&](uint32_t w) -> uint32_t {
            const uint32_t id = w & IDMASK;
            const uint32_t e = ph[(__umul24(id, hm) >> 8) & 255u];
            return (e >> 8) == id ? (e & 15u) : 0u;
        };
#pragma unroll
        for (int j = 0; j < MJ; j++) {
#pragma unroll
            for (int k = 0; k < 4; k++) ia[j] |= code(x[j][k]) << (4 * k);
        }
        ip = code(prev1);
    } else {
        for (uint32_t p = 0; p < K; p++) {
            const uint32_t a = pa[p];
#pragma unroll
            for (int j = 0; j < MJ; j++) {
#pragma unroll
                for (int k = 0; k < 4; k++) ia[j] |= ((x[j][k] & IDMASK) == a) ? (p + 1u) << (4 * k) : 0u;
            }
            ip = ((prev1 & IDMASK) == a) ? p + 1u : ip;
        }
    }
#pragma unroll
    for (int j = 0; j < MJ; j++) {
        const uint32_t up = (j < MJ - 1) ? lane_first(x[(j + 1) % MJ][0]) : tail[0];
        const uint32_t nxw = lane_next(x[j][0], up);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t nxt = (k < 3) ? x[j][k + 1] : nxw;
            const uint32_t i = (ia[j] >> (4 * k)) & 15u;
            const uint32_t m = (uint32_t)((nxt & NWMASK) == pb1[i]);
            rb[j] |= m << k;
            jc[j] |= (m ? i : 0u) << (4 * k);
        }
    }
    s = (uint32_t)((prev1 != INVALID_WORD) & ((first & NWMASK) == pb1[ip]));
#pragma unroll
    for (int j = 0; j < MJ; j++) {
        const int nb = (int)len - (j * 256 + lane * 4);
        valid[j] = nb >= 4 ? 0xFu : (nb <= 0 ? 0u : ((1u << nb) - 1u));
    }
    uint32_t anyr = 0;
#pragma unroll
    for (int j = 0; j < MJ; j++) anyr |= rb[j] & valid[j];
    const bool sites = __any(anyr != 0) != 0;
    if (!sites && !s) {  // nothing in this slot changes and it owes no table update
        keep_header();
        return;
    }
    // ---- (5) kept flags, output offsets -------------------------------------------------------
    uint32_t mb[MJ], kb[MJ], ex[MJ], cnt[MJ];
#pragma unroll
    for (int j = 0; j < MJ; j++) {
        mb[j] = rb[j] & valid[j];
        const uint32_t upr = (j > 0) ? ((lane_last(rb[(j + MJ - 1) % MJ]) >> 3) & 1u) : s;
        const uint32_t mp = (uint32_t)dpp_mov<0x138>((int)upr, (int)((rb[j] >> 3) & 1u));  // wave_shr:1, lane 0 keeps upr
        kb[j] = ~((mb[j] << 1) | mp) & valid[j] & 0xFu;
        cnt[j] = (uint32_t)__popc(kb[j]);
    }
    uint32_t total = 0;
    if constexpr (MJ == 4) {  // (two stripes' counts per scan)
        const uint32_t i01 = wave_iscan_add(cnt[0] | (cnt[1] << 16));
        const uint32_t i23 = wave_iscan_add(cnt[2] | (cnt[3] << 16));
        const uint32_t t01 = lane_last(i01), t23 = lane_last(i23);
        const uint32_t tot0 = t01 & 0xFFFFu, tot1 = t01 >> 16, tot2 = t23 & 0xFFFFu, tot3 = t23 >> 16;
        ex[0] = (i01 & 0xFFFFu) - cnt[0];
        ex[1] = tot0 + (i01 >> 16) - cnt[1];
        ex[2] = tot0 + tot1 + (i23 & 0xFFFFu) - cnt[2];
        ex[3] = tot0 + tot1 + tot2 + (i23 >> 16) - cnt[3];
        total = tot0 + tot1 + tot2 + tot3;
    } else {
#pragma unroll
        for (int j = 0; j < MJ; j++) {
            const uint32_t inc = wave_iscan_add(cnt[j]);
            ex[j] = total + inc - cnt[j];
            total += lane_last(inc);
        }
    }
    // ---- (6) stage the compacted slot in LDS, then 16-byte stores back to its home -------------
    uint32_t fstore = 0;  // a dropped first word moves everything
    if (!s) {
        uint32_t fc = 0x7FFFFFFFu;
#pragma unroll
        for (int j = 0; j < MJ; j++) {
            if (mb[j]) {
                const uint32_t k0 = (uint32_t)__ffs((int)mb[j]) - 1u;
                fc = min(fc, ex[j] + (uint32_t)__popc(kb[j] & ((1u << k0) - 1u)));
            }
        }
        fstore = (uint32_t)wave_min_i32((int)fc) & ~3u;
    }
#pragma unroll
    for (int j = 0; j < MJ; j++) {
        uint32_t o = ex[j];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if ((kb[j] >> k) & 1u) {
                const uint32_t w = x[j][k];
                const uint32_t z = z0 + ((jc[j] >> (4 * k)) & 15u) - 1u;
                out[o++] = ((mb[j] >> k) & 1u) ? (z | (w & (FLAG | WMASK))) : w;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
        uint32_t *dst = (buf ? A.b1 : A.b0) + (size_t)t * TILE2;
        for (uint32_t i = fstore + (uint32_t)lane * 4; i < total; i += 256)
            *reinterpret_cast<uint4 *>(dst + i) = *reinterpret_cast<const uint4 *>(&out[i]);
    }
    if (lane == 0) {
        uint32_t h[8];
        h[0] = total > 0 ? out[0] : INVALID_WORD;
        h[1] = total > 1 ? out[1] : INVALID_WORD;
        h[2] = total > 2 ? out[2] : INVALID_WORD;
        h[3] = total | (buf << 31);
        h[4] = total > 1 ? out[total - 2] : INVALID_WORD;
        h[5] = total > 0 ? out[total - 1] : INVALID_WORD;
        h[6] = h[7] = 0;
        if (DENSE) {
            reinterpret_cast<uint4 *>(A.hdr_out)[2 * (size_t)t] = make_uint4(h[0], h[1], h[2], h[3]);
            reinterpret_cast<uint4 *>(A.hdr_out)[2 * (size_t)t + 1] = make_uint4(h[4], h[5], 0u, 0u);
        } else {
            stage_put<THROUGH>(A.stage + t, t, h);
            atomicOr(&A.smask[t >> 5], 1u << (t & 31));
        }
        if (total < 3 && t + 1 < Tl) A.st->gap = 1;
    }
    if (!sites) return;  // carry only: the site belongs to the previous slot
    // ---- (7) pair-table delta of my sites, as the sequential merges would charge it ------------
    // Which pair (if any) the two words before a site, and the two after it, are a site of: the codes of the
    // positions two to the left and two to the right -- computed above for every position, so a site only looks
    // at its neighbours' nibbles (no search through the batch per site: the early passes have a site in nearly
    // every group of words, and were bound by instruction issue while every site compared its neighbours with
    // every pair).  Positions -2 and -1 (the previous slot's last words) and 1024, 1025 come from the context words.
    const uint32_t vc = A.vcap & 0xFFFFFFu;
    // (uniform values, K compares each on the vector unit: looked up only when a site reads them -- position -2 by a site
    // on the slot's first word, positions 1024 / 1025 by sites on its last two words)
    uint32_t cm2 = 0, ct0 = 0, ct1 = 0;
    if (bcast(mb[0], 0) & 1u) cm2 = (uint32_t)(chain_match(pa, pb, K, prev2, prev1) + 1);
    const uint32_t cm1 = s ? ip : 0u;
    if (bcast(mb[MJ - 1], 63) & 0xCu) {
        ct0 = (uint32_t)(chain_match(pa, pb, K, tail[0], tail[1]) + 1);
        ct1 = (uint32_t)(chain_match(pa, pb, K, tail[1], tail[2]) + 1);
    }
#pragma unroll
    for (int j = 0; j < MJ; j++) {
        if (!__any(mb[j] != 0)) continue;  // (uniform) no site in this stripe
        uint32_t upm1, dn0;
        if (j > 0) upm1 = lane_last(x[(j + MJ - 1) % MJ][3]);
        else upm1 = prev1;
        if (j < MJ - 1) dn0 = lane_first(x[(j + 1) % MJ][0]);
        else dn0 = tail[0];
        // the words at positions q0 - 1 .. q0 + 5 (q0 = my first position)
        uint32_t W[7];
        W[0] = (uint32_t)dpp_mov<0x138>((int)upm1, (int)x[j][3]);  // (lane 0 keeps upm1)
        W[1] = x[j][0];
        W[2] = x[j][1];
        W[3] = x[j][2];
        W[4] = x[j][3];
        W[5] = lane_next(x[j][0], dn0);
        W[6] = lane_next(x[j][1], (j < MJ - 1) ? lane_first(x[(j + 1) % MJ][1]) : tail[1]);
        // the codes of positions q0 - 2 .. q0 + 5, a nibble each
        const uint32_t lo_fill = (j > 0) ? ((lane_last(jc[(j + MJ - 1) % MJ]) >> 8) & 0xFFu) : (cm2 | (cm1 << 4));
        const uint32_t hi_fill = (j < MJ - 1) ? (lane_first(jc[(j + 1) % MJ]) & 0xFFu) : (ct0 | (ct1 << 4));
        const uint32_t lo = (uint32_t)dpp_mov<0x138>((int)lo_fill, (int)((jc[j] >> 8) & 0xFFu));
        const uint32_t hi = lane_next(jc[j] & 0xFFu, hi_fill);
        const uint32_t win = lo | (jc[j] << 8) | (hi << 24);
        if (mb[j] == 0) continue;
        // A lane's four words hold at most TWO sites (no two sites of a batch overlap), and a sparse
        // pass has one or two sites in a whole slot: the lanes take their FIRST site together, then their second, instead
        // of one round per word position with a lane or two active in each (the site's words by select, not by index).
        uint32_t todo = mb[j];
#pragma unroll
        for (int si = 0; si < 2; si++) {
            if (todo == 0) continue;
            const uint32_t k = (uint32_t)__ffs((int)todo) - 1u;
            todo &= todo - 1u;
            const int q = j * 256 + lane * 4 + (int)k;
            const uint32_t wk = win >> (4u * k);
            const uint32_t p = ((wk >> 8) & 15u) - 1u;
            const int li = (int)(wk & 15u) - 1, ri = (int)((wk >> 16) & 15u) - 1;
            const uint32_t Z = z0 + p;
            uint32_t *dl, *dr;  // SL, SR of pair p
            if (DENSE) {
                dl = sd + p * CH_SD;
                dr = dl + CH_DCAP;
                atomicAdd(&dl[2 * CH_DCAP + 1], 1u);
            } else {
                // ids removed, per pair: every site is counted by the slot that owns its first word (one atomic per
                // site, CH_RMV counters per pair)
                if (A.removed) atomicAdd(&A.removed[(p * (uint32_t)CH_RMV + (t & (uint32_t)(CH_RMV - 1))) * REMOVED_STRIDE], 1u);
                dl = A.delta + delta_rep_off(p * (uint32_t)CH_RSTRIDE + (t & (brep - 1u)), vc);
                dr = dl + vc;
            }
            const bool k1 = (k & 1u) != 0, k2 = (k & 2u) != 0;
            const uint32_t wa = k2 ? (k1 ? W[4] : W[3]) : (k1 ? W[2] : W[1]);
            const uint32_t wt = word_weight(wa);
            const uint32_t Lw = k2 ? (k1 ? W[3] : W[2]) : (k1 ? W[1] : W[0]);
            if (!(wa & FLAG) && Lw != INVALID_WORD) {
                // the left neighbour ends a site of pair li: when pair p is merged it already reads Z_li
                // (li < p) or still b_li (li > p); li == p is the same pair twice in a row -- format B's adj,
                // charged by the left site
                if (li != (int)p) {
                    const uint32_t Lseq = (li >= 0 && (uint32_t)li < p) ? z0 + (uint32_t)li : (Lw & IDMASK);
                    atomicAdd(&dl[Lseq], wt);
                    if (!DENSE || A.idx) {
                        const uint32_t Lfin = li >= 0 ? z0 + (uint32_t)li : (Lw & IDMASK);
                        index_add(A.idx, A.istride, t, Lfin, Z);
                        if (q == 0) index_add(A.idx, A.istride, tprev, Lfin, Z);
                    }
                }
            }
            const uint32_t R = k2 ? (k1 ? W[6] : W[5]) : (k1 ? W[4] : W[3]);
            if (!(R & FLAG)) {  // (INVALID_WORD has the flag bit set: end of stream)
                if (ri == (int)p) {
                    if (DENSE) atomicAdd(&dl[2 * CH_DCAP], wt);
                    else atomicAdd(&A.st->badj[p], wt);
                } else {
                    atomicAdd(&dr[(ri >= 0 && (uint32_t)ri < p) ? z0 + (uint32_t)ri : (R & IDMASK)], wt);
                }
                if (!DENSE || A.idx) {
                    const uint32_t y = ri >= 0 ? z0 + (uint32_t)ri : (R & IDMASK);
                    index_add(A.idx, A.istride, t, Z, y);
                    if (q + 2 >= (int)len) index_add(A.idx, A.istride, tnext, Z, y);
                }
            }
        }
    }
}

// The look-up table of a batch's first tokens (merge_chain_wave): wave 0 tries 64 multipliers at once, lane l its own,
// and takes the first under which no two of the K tokens fall into the same of the 256 buckets (K <= 15: two in three
// multipliers do); returns it to every thread (0: none found, or a batch too small to pay -- the pass compares).
// s_ph: 256 words, s_hm: one word.  Every thread calls.
// the multiplier itself (one wave; returns it to lane 0 .. 63 alike): 0 = none found, or a batch too small to pay
__device__ __forceinline__ uint32_t chain_hash_find(const uint32_t *pa, uint32_t K) {
    if (K < 5) return 0u;
    const uint32_t m = 129u + 2u * (uint32_t)lane_id();
    bool ok = true;
    for (uint32_t i = 1; i < K; i++) {
        const uint32_t hi = (__umul24(pa[i], m) >> 8) & 255u;
        for (uint32_t j = 0; j < i; j++) ok &= hi != ((__umul24(pa[j], m) >> 8) & 255u);
    }
    const unsigned long long bal = __ballot(ok);
    return bal ? 129u + 2u * (uint32_t)(__ffsll((long long)bal) - 1) : 0u;
}
// (pre_ok: the selection already found the multiplier, pre_hm -- k_pool.hip: pool_finish; DevState::bhm)
__device__ __forceinline__ uint32_t chain_hash_build(uint32_t *s_ph, uint32_t *s_hm, const uint32_t *s_pa, uint32_t K,
                                                     bool pre_ok = false, uint32_t pre_hm = 0) {
    if (threadIdx.x < 256) s_ph[threadIdx.x] = 0xFFFFFFFFu;
    uint32_t hm = pre_hm;
    if (!pre_ok) {  // (uniform)
        if (threadIdx.x == 0) *s_hm = 0;
        __syncthreads();
        if (wave_id() == 0) {
            const uint32_t f = chain_hash_find(s_pa, K);
            if (lane_id() == 0) *s_hm = f;
        }
        __syncthreads();
        hm = *s_hm;
    } else {
        __syncthreads();
    }
    if (hm && threadIdx.x < K) s_ph[(__umul24(s_pa[threadIdx.x], hm) >> 8) & 255u] = (s_pa[threadIdx.x] << 8) | (threadIdx.x + 1u);
    __syncthreads();
    return hm;
}

// ---------------------------------------------------------------------------
// merge pass of a chain step: k_merge_ab_lean for the K pairs of the batch (index live).  The body is shared by
// k_merge_chain (its own launch: the batch comes from st) and k_step (k_step.hip: one launch per step, the batch comes
// from the deciding workgroup's published line); nblk workgroups take part, this one is number blk.
struct MergeLds {
    alignas(16) uint32_t s_out[LEAN_MT / 64][TILE2];
    uint32_t s_list[LEAN_SUB * 32];
    uint32_t s_tot[2];
    uint32_t s_pa[CH_KMAX], s_pb[CH_KMAX], s_pb1[CH_KMAX + 1];
    uint32_t s_ph[256], s_hm;
};
// (s_pa / s_pb / s_pb1 are filled and a barrier has passed; every thread of the workgroup calls; THROUGH: k_step -- the
// staged headers are committed by other workgroups of the same launch)
template <bool THROUGH = false>
__device__ __forceinline__ void merge_chain_body(const AbArgs &A, const uint32_t *__restrict__ idx_dirty, uint32_t use_index,
                                                 MergeLds &L, const uint32_t K, const uint32_t z0, const uint32_t brep,
                                                 const uint32_t blk, const uint32_t nblk, unsigned long long *dbg = nullptr,
                                                 const bool have_st = false, const uint32_t tlive_in = 0, const uint32_t gap_in = 0,
                                                 const bool pre_ok = false, const uint32_t pre_hm = 0) {
    auto dstamp = [&](int i) {  // (debug, BPE_STEP_STAMPS)
        if (dbg && threadIdx.x == 0) dbg[i] = wall_clock64();
    };
    DevState *st = A.st;
    auto &s_out = L.s_out;
    auto &s_list = L.s_list;
    auto &s_tot = L.s_tot;
    auto &s_pa = L.s_pa;
    auto &s_pb = L.s_pb;
    auto &s_pb1 = L.s_pb1;
    const uint32_t hm = chain_hash_build(L.s_ph, &L.s_hm, s_pa, K, pre_ok, pre_hm);
    const uint32_t *s_ph = L.s_ph;
    // (have_st: the caller fetched the state words this pass needs in one round trip with the batch)
    const uint32_t Tl = min(A.T, have_st ? tlive_in : st->tlive);
    const uint32_t gap = have_st ? gap_in : st->gap;
    constexpr uint32_t NWV = LEAN_MT / 64;
    // a batch of one: the single-pair rewrite (merge_ab_wave, k_slots2.hip) -- no per-pair loops, format B's
    // adj in st->adj, all 256 removal counters
    AbArgs A1 = A;
    A1.newid = z0;
    const uint32_t a0 = s_pa[0], b0 = s_pb[0];
    auto do_slot = [&](uint32_t t) {
        if (K == 1) {
            merge_ab_wave<true, true, false, THROUGH>(s_out[wave_id()], nullptr, t, A1, a0, b0, Tl);
        } else {
            uint4 rv[MJ], hv;
            chain_slot_load(A, t, rv, hv);
            merge_chain_wave<false, THROUGH>(s_out[wave_id()], nullptr, t, A, s_pa, s_pb, s_pb1, K, z0, brep, rv, hv, s_ph, hm, Tl);
        }
    };
    if (!(use_index & 1u) || gap != 0) {  // short slots about: visit everything
        const uint32_t nw = nblk * NWV;
        for (uint32_t t = blk * NWV + wave_id(); t < Tl; t += nw) do_slot(t);
        return;
    }
    dstamp(8);
    const uint32_t nwords = (Tl + 31) / 32;
    const uint32_t per = (nwords + nblk - 1) / nblk;  // mask words of one workgroup
    const uint32_t wlo = blk * per, whi = min(nwords, wlo + per);
    for (uint32_t sub = wlo; sub < whi; sub += LEAN_SUB) {
        uint32_t mk = 0;
        const uint32_t w = sub + threadIdx.x;
        if (threadIdx.x < LEAN_SUB && w < whi) {
            mk = idx_dirty[w];
            for (uint32_t p = 0; p < K; p++) {
                uint32_t h1, h2, h3;
                pair_hash(s_pa[p], s_pb[p], h1, h2, h3);
                mk |= A.idx[(size_t)h1 * A.istride + w] & A.idx[(size_t)h2 * A.istride + w] &
                      A.idx[(size_t)h3 * A.istride + w];
            }
            const uint32_t left = Tl - w * 32;
            if (left < 32) mk &= (1u << left) - 1u;
        }
        const uint32_t c = (uint32_t)__popc(mk);
        const uint32_t inc = wave_iscan_add(c);
        if (threadIdx.x < 128 && lane_id() == 63) s_tot[wave_id()] = inc;
        __syncthreads();
        const uint32_t n = s_tot[0] + s_tot[1];
        if (threadIdx.x < LEAN_SUB) {
            uint32_t o = inc - c + (wave_id() == 1 ? s_tot[0] : 0u);
            while (mk) {
                s_list[o++] = w * 32 + (uint32_t)__ffs((int)mk) - 1u;
                mk &= mk - 1u;
            }
        }
        __syncthreads();
        if (sub == wlo) dstamp(9);
        if (K == 1 || MJ > 1 || !(use_index & 2u)) {
            for (uint32_t i = wave_id(); i < n; i += NWV) do_slot(s_list[i]);
        } else {
            // 256-id slots: a slot is 16 bytes per lane and six header pieces -- the wave's NEXT candidate travels while it
            // works on this one (a wave goes through dozens of candidates in a mid-training sweep, one round trip each)
            uint32_t i = wave_id();
            uint4 rv[MJ], hv, nrv[MJ], nhv;
            if (i < n) chain_slot_load(A, s_list[i], rv, hv);
            for (; i < n; i += NWV) {
                const bool more = i + NWV < n;
                if (more) chain_slot_load(A, s_list[i + NWV], nrv, nhv);
                merge_chain_wave<false, THROUGH>(s_out[wave_id()], nullptr, s_list[i], A, s_pa, s_pb, s_pb1, K, z0, brep, rv, hv, s_ph, hm, Tl);
                if (more) {
#pragma unroll
                    for (int j = 0; j < MJ; j++) rv[j] = nrv[j];
                    hv = nhv;
                }
            }
        }
        __syncthreads();  // (the list is rewritten by the next round)
    }
}
// The words of DevState a chain step's merge pass / table update needs, fetched by 64 lanes at once (ONE round trip: the
// fields used to be read where the code came to them -- status, then the batch, then tlive / gap: three dependent round
// trips, ~4.5 us at the head of every launch): [p] = ba[p], [16 + p] = bb[p], [32 + p] = badj[p], then the scalars below.
enum { SW_BK = 48, SW_BZ0, SW_BREP, SW_STATUS, SW_DEFER, SW_SEL_RAN, SW_TLIVE, SW_GAP, SW_BHM, SW_BHM_KEY, SW_ADJ, SW_N };
static_assert(SW_N <= 64, "one wave fetches the state words");
__device__ __forceinline__ void step_words_fetch(const DevState *st, uint32_t *s_w) {
    const uint32_t i = threadIdx.x;
    if (i < (uint32_t)SW_N) {
        const uint32_t *base = reinterpret_cast<const uint32_t *>(st);
        uint32_t off;
        if (i < 16) off = (uint32_t)offsetof(DevState, ba) / 4 + i;
        else if (i < 32) off = (uint32_t)offsetof(DevState, bb) / 4 + (i - 16);
        else if (i < 48) off = (uint32_t)offsetof(DevState, badj) / 4 + (i - 32);
        else {
            constexpr uint32_t o[SW_N - 48] = {
                (uint32_t)offsetof(DevState, bk) / 4,      (uint32_t)offsetof(DevState, bz0) / 4,     (uint32_t)offsetof(DevState, brep) / 4,
                (uint32_t)offsetof(DevState, status) / 4,  (uint32_t)offsetof(DevState, defer) / 4,   (uint32_t)offsetof(DevState, sel_ran) / 4,
                (uint32_t)offsetof(DevState, tlive) / 4,   (uint32_t)offsetof(DevState, gap) / 4,     (uint32_t)offsetof(DevState, bhm) / 4,
                (uint32_t)offsetof(DevState, bhm_key) / 4, (uint32_t)offsetof(DevState, adj) / 4};
            off = o[0];
#pragma unroll
            for (int k = 1; k < SW_N - 48; k++) off = (i == 48u + (uint32_t)k) ? o[k] : off;
        }
        s_w[i] = base[off];
    }
    __syncthreads();
}
__global__ void __launch_bounds__(LEAN_MT)
k_merge_chain(AbArgs A, const uint32_t *__restrict__ idx_dirty, uint32_t use_index, uint32_t *__restrict__ dbits) {
    __shared__ MergeLds L;
    __shared__ uint32_t s_w[64];
    DevState *st = A.st;
    step_words_fetch(st, s_w);
    // (the flagged rows were re-scanned by the selection launch before this one if that was a FULL one)
    const uint32_t ran = s_w[SW_SEL_RAN];
    if (blockIdx.x == 0 && ran) {
        for (uint32_t i = threadIdx.x; i < DBITS_WORDS; i += LEAN_MT) dbits[i] = 0;
        __syncthreads();
        if (threadIdx.x == 0) st->sel_ran = 0;
    }
    const uint32_t K = s_w[SW_BK];
    if (s_w[SW_STATUS] || s_w[SW_DEFER] || K == 0) return;
    const uint32_t z0 = s_w[SW_BZ0], brep = s_w[SW_BREP];
    if (threadIdx.x < CH_KMAX) {
        L.s_pa[threadIdx.x] = threadIdx.x < K ? s_w[threadIdx.x] : 0xFFFFFFFFu;
        L.s_pb[threadIdx.x] = threadIdx.x < K ? s_w[16 + threadIdx.x] : 0xFFFFFFFFu;
        L.s_pb1[threadIdx.x + 1] = threadIdx.x < K ? s_w[16 + threadIdx.x] : 0xFFFFFFFFu;
        if (threadIdx.x == 0) L.s_pb1[0] = 0xFFFFFFFFu;  // (a masked word has its weight bits clear: never equal)
    }
    __syncthreads();
    merge_chain_body(A, idx_dirty, use_index, L, K, z0, brep, blockIdx.x, gridDim.x, nullptr, true, s_w[SW_TLIVE], s_w[SW_GAP],
                     s_w[SW_BHM_KEY] == ((z0 << 8) | K), s_w[SW_BHM]);
}

// ---------------------------------------------------------------------------
// merge pass of a DENSE chain step: the early merges, whose pairs sit in nearly every slot (the inverted index does
// not exist yet, and a pair has 10^5 .. 10^7 sites: the delta goes through LDS tables, as in k_merge_ab_dense_early).
// A batch of K pairs costs ONE sweep over the stream instead of K.  One 1024-thread workgroup per CU, wave w of the
// grid takes slots w, w + waves, ...; every slot's header is written to the other header array (the host flips the
// arrays after every dense step -- a step that merges nothing copies the headers across, so that the flip stands).
constexpr int CH_KDENSE = 6;  // most pairs of a dense step's batch: their LDS tables must fit next to the staging (8 pairs measured no faster than 6; the first-token look-up table of the sparse pass measured no faster than K compares here)
static_assert((LEAN_MT / 64) * TILE2 * 4 + CH_KDENSE * (int)CH_SD * 4 + 256 <= 160 * 1024, "dense chain pass: LDS");
__global__ void __launch_bounds__(LEAN_MT)
k_merge_chain_dense(AbArgs A, uint32_t *__restrict__ dbits) {
    __shared__ __attribute__((aligned(16))) uint32_t s_out[LEAN_MT / 64][TILE2];
    __shared__ uint32_t s_sd[CH_KDENSE * CH_SD];
    __shared__ uint32_t s_pa[CH_KMAX], s_pb[CH_KMAX], s_pb1[CH_KMAX + 1];
    DevState *st = A.st;
    const uint32_t ran = st->sel_ran;
    if (blockIdx.x == 0 && ran) {
        for (uint32_t i = threadIdx.x; i < DBITS_WORDS; i += LEAN_MT) dbits[i] = 0;
        __syncthreads();
        if (threadIdx.x == 0) st->sel_ran = 0;
    }
    const uint32_t K = st->bk;
    if (!(st->status || st->defer) && K == 1) return;  // (k_merge_chain_dense1's pass)
    if (st->status || st->defer || K == 0 || K > (uint32_t)CH_KDENSE) {
        // nothing to merge: the headers cross over unchanged
        const uint4 *hi = reinterpret_cast<const uint4 *>(A.hdr_in);
        uint4 *ho = reinterpret_cast<uint4 *>(A.hdr_out);
        for (size_t i = (size_t)blockIdx.x * LEAN_MT + threadIdx.x; i < 2 * (size_t)A.T; i += (size_t)gridDim.x * LEAN_MT) ho[i] = hi[i];
        if (K > (uint32_t)CH_KDENSE && blockIdx.x == 0 && threadIdx.x == 0) st->status = ST_INTERNAL;
        return;
    }
    const uint32_t z0 = st->bz0, brep = st->brep;
    if (threadIdx.x < CH_KMAX) {
        s_pa[threadIdx.x] = threadIdx.x < K ? (uint32_t)st->ba[threadIdx.x] : 0xFFFFFFFFu;
        s_pb[threadIdx.x] = threadIdx.x < K ? (uint32_t)st->bb[threadIdx.x] : 0xFFFFFFFFu;
        s_pb1[threadIdx.x + 1] = threadIdx.x < K ? (uint32_t)st->bb[threadIdx.x] : 0xFFFFFFFFu;
        if (threadIdx.x == 0) s_pb1[0] = 0xFFFFFFFFu;
    }
    for (uint32_t i = threadIdx.x; i < K * CH_SD; i += LEAN_MT) s_sd[i] = 0;
    __syncthreads();
    constexpr uint32_t NWV = LEAN_MT / 64;
    const uint32_t nw = gridDim.x * NWV;
    const uint32_t Tl = min(A.T, st->tlive);
    for (uint32_t t = blockIdx.x * NWV + wave_id(); t < A.T; t += nw)
    {
        uint4 rv[MJ], hv;
        chain_slot_load(A, t, rv, hv);
        merge_chain_wave<true>(s_out[wave_id()], s_sd, t, A, s_pa, s_pb, s_pb1, K, z0, (uint32_t)CH_RSTRIDE, rv, hv, nullptr, 0, Tl);
    }
    __syncthreads();
    // flush: the tables of pair p into one of its CH_RSTRIDE replica blocks
    const uint32_t vc = A.vcap & 0xFFFFFFu;
    const uint32_t lim = min(vc, (uint32_t)CH_DCAP);
    for (uint32_t p = 0; p < K; p++) {
        const uint32_t *sd = s_sd + p * CH_SD;
        // (as many replica blocks per pair as the table update will fold: st->brep)
        uint32_t *g = A.delta + delta_rep_off(p * (uint32_t)CH_RSTRIDE + (blockIdx.x & (brep - 1u)), vc);
        for (uint32_t i = threadIdx.x; i < lim; i += LEAN_MT) {
            const uint32_t l = sd[i], r = sd[CH_DCAP + i];
            if (l) atomicAdd(&g[i], l);
            if (r) atomicAdd(&g[vc + i], r);
        }
        if (threadIdx.x == 0) {
            const uint32_t adj = sd[2 * CH_DCAP], rem = sd[2 * CH_DCAP + 1];
            if (adj) atomicAdd(&st->badj[p], adj);
            if (rem && A.removed) atomicAdd(&A.removed[(p * (uint32_t)CH_RMV + (blockIdx.x & (uint32_t)(CH_RMV - 1))) * REMOVED_STRIDE], rem);
        }
    }
}
// ... and the dense step whose batch is ONE pair: k_merge_ab_dense_early (k_slots2.hip) with the pair and its new
// id taken from the device's own state (a resident grid of 256-thread workgroups, five per CU; the delta through
// LDS, flushed into one of the pass's replica blocks; format B's adj in st->adj, all 256 removal counters).
__global__ void __launch_bounds__(MT, 5)
k_merge_chain_dense1(AbArgs A) {
    __shared__ __attribute__((aligned(16))) uint32_t s_out[MT / 64][TILE2];
    __shared__ uint32_t s_delta[2 * LDSD_CAP + 2];
    const DevState *st = A.st;
    if (st->status || st->defer || st->bk != 1) return;
    const uint32_t a = (uint32_t)st->ba[0], b = (uint32_t)st->bb[0];
    A.newid = st->bz0;
    ldsd_clear(s_delta);
    const uint32_t nw = gridDim.x * (MT / 64);
    const uint32_t Tl = min(A.T, st->tlive);
    for (uint32_t t = blockIdx.x * (MT / 64) + wave_id(); t < A.T; t += nw)
        merge_ab_wave<false, false, true>(s_out[wave_id()], s_delta, t, A, a, b, Tl);
    ldsd_flush(s_delta, A);
}

// ---------------------------------------------------------------------------
// table update of a chain step, in three parts shared by k_apply_chain (its own launch) and k_step (k_step.hip: the tail
// of the step's one launch):
//   apply_chain_tokens   one token per thread, every pair of the batch;
//   apply_chain_records  64 threads: the stream length, the iteration records of the step's merges, the step record;
//   apply_chain_commit   the staged headers.
// OWNED == false: the flag words (dbits) are shared with whoever set them before -- atomicOr.  OWNED == true (k_step):
// nobody else touches them during this phase, and the wave that holds tokens [64 k, 64 k + 64) is the only writer of
// words 2 k and 2 k + 1: it stores them whole -- and starts from zero when the step's selection re-scanned every flagged
// row (`ran`), which is how k_step clears the flags without a pass of its own.
// The batch comes from the caller, in LDS: pairs[p] = a << 16 | b, adjs[p] = format B's adj of pair p (adjs[0] for a batch of
// one: merge_ab_wave's st->adj), brep_in.  OWNED also means: this is a phase of k_step -- what other workgroups wrote
// earlier in this launch (the delta words by device atomics, the row maxima of re-scanned rows) is read with agent-scope
// loads (k_common.hip).
template <bool OWNED>
__device__ __forceinline__ void apply_chain_tokens(const uint32_t t, uint32_t *__restrict__ mat, uint32_t stride,
                                                   uint32_t *__restrict__ delta, uint32_t vcap, const uint32_t *__restrict__ rowmax,
                                                   uint32_t *__restrict__ dbits, uint4 *__restrict__ sums,
                                                   const uint32_t *__restrict__ folded, uint32_t fS, const uint32_t *__restrict__ ftail,
                                                   const uint32_t K, const uint32_t z0, const uint32_t ran,
                                                   const uint32_t *pairs, const uint32_t *adjs, const uint32_t brep_in) {
    auto dld = [&](const uint32_t *p) -> uint32_t { return OWNED ? ld_agent(p) : *p; };
    auto pair_a = [&](uint32_t p) -> uint32_t { return pairs[p] >> 16; };
    auto pair_b = [&](uint32_t p) -> uint32_t { return pairs[p] & 0xFFFFu; };
    {
        const uint32_t Zlast = z0 + K - 1u;
        if ((t & ~255u) > Zlast) return;  // (the host sized the grid for the most a step can reach)
        const bool live = t <= Zlast;  // (dead lanes stay for the wave reductions below)
        const uint32_t nrep = 1u << (vcap >> 24);
        const uint32_t vc = vcap & 0xFFFFFFu;
        uint2 rm = make_uint2(0u, 0u);
        uint32_t prevflag = 0, oldword = 0;
        if (OWNED) {
            oldword = ran ? 0u : dbits[t >> 5];  // (every lane of the half-wave reads its word: one request)
            prevflag = (oldword >> (t & 31)) & 1u;
            if (live) {  // (a row re-scanned by this launch's selection has its maximum from another workgroup)
                const unsigned long long v = ld_agent64(reinterpret_cast<const unsigned long long *>(rowmax) + t);
                rm = make_uint2((uint32_t)v, (uint32_t)(v >> 32));
            }
        } else if (live) {
            prevflag = (dbits[t >> 5] >> (t & 31)) & 1u;  // (flagged by an earlier step of this level, not re-scanned yet)
            rm = reinterpret_cast<const uint2 *>(rowmax)[t];
        }
        bool flagged = false;
        if (K == 1) {
            // ---- one pair: all nrep replicas, (t,a) loaded up front (no returning atomic) ----------------
            const uint32_t a = pair_a(0), b = pair_b(0), Z = z0;
            const uint32_t adj = folded ? ftail[0] : adjs[0];  // (merge_ab_wave's adj)
            constexpr int RB = OWNED ? 8 : 16;  // replicas in flight at a time (k_step: 128 registers per lane)
            uint32_t x[RB][2];
            auto load_batch = [&](uint32_t r0) {
#pragma unroll
                for (int k = 0; k < RB; k++) {
                    const uint32_t r = r0 + k;
                    x[k][0] = (live && r < nrep) ? dld(&delta[delta_rep_off(r, vc) + t]) : 0u;
                    x[k][1] = (live && r < nrep) ? dld(&delta[delta_rep_off(r, vc) + vc + t]) : 0u;
                }
            };
            if (!folded) load_batch(0);
            const uint32_t old_ta = live ? mat[(size_t)t * stride + a] : 0u;
            uint32_t sl = 0, sr = 0;
            if (folded) {
                sl = live ? folded[t] : 0u;
                sr = live ? folded[(size_t)fS + t] : 0u;
            }
            for (uint32_t r0 = 0; !folded;) {
#pragma unroll
                for (int k = 0; k < RB; k++) {
                    if (x[k][0]) delta[delta_rep_off(r0 + k, vc) + t] = 0;
                    if (x[k][1]) delta[delta_rep_off(r0 + k, vc) + vc + t] = 0;
                    sl += x[k][0];
                    sr += x[k][1];
                }
                r0 += RB;
                if (r0 >= nrep) break;
                load_batch(r0);
            }
            const uint32_t dr = sr + (t == a ? adj : 0u), ir = sr + (t == Z ? adj : 0u);
            if (sl) {
                atomicSub(&mat[(size_t)t * stride + a], sl);
                atomicAdd(&mat[(size_t)t * stride + Z], sl);
                flagged |= old_ta == rm.x;  // row t's maximum moves only if (t,a) attained it ((t,Z) = sl <= what (t,a) lost)
            }
            if (dr) atomicSub(&mat[(size_t)b * stride + t], dr);
            if (ir) atomicAdd(&mat[(size_t)Z * stride + t], ir);
            if (live && t == b) mat[(size_t)a * stride + b] = 0;  // no (a,b) survives the merge (F2)
            flagged |= live && ((t == a) | (t == b) | (t == Z));
        } else {
            // ---- a batch: brep replicas per pair, eight pairs' worth of loads in flight at a time -------------------------
            const uint32_t brep = folded ? 0u : brep_in;
            constexpr int G = OWNED ? 4 : 8;  // (k_step's workgroups are 1024 threads: 128 registers per lane, not 256)
            for (uint32_t g = 0; g < K; g += G) {  // (uniform)
                uint32_t x[G][CH_REP][2];
                if (brep == (uint32_t)CH_REP) {
#pragma unroll
                    for (int q = 0; q < G; q++) {
#pragma unroll
                        for (int r = 0; r < CH_REP; r++) {
                            const size_t o = delta_rep_off((g + (uint32_t)q) * (uint32_t)CH_RSTRIDE + (uint32_t)r, vc);
                            x[q][r][0] = (live && g + (uint32_t)q < K) ? dld(&delta[o + t]) : 0u;
                            x[q][r][1] = (live && g + (uint32_t)q < K) ? dld(&delta[o + vc + t]) : 0u;
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < G; q++) {
                    const uint32_t p = g + (uint32_t)q;
                    if (p >= K) break;  // (uniform)
                    const uint32_t a = pair_a(p), b = pair_b(p), Z = z0 + p;
                    const uint32_t adj = folded ? ftail[p] : adjs[p];
                    uint32_t sl = 0, sr = 0;
                    if (folded) {
                        sl = live ? folded[(size_t)(2 * p) * fS + t] : 0u;
                        sr = live ? folded[(size_t)(2 * p + 1) * fS + t] : 0u;
                    } else if (brep == (uint32_t)CH_REP) {
#pragma unroll
                        for (int r = 0; r < CH_REP; r++) {
                            const size_t o = delta_rep_off(p * (uint32_t)CH_RSTRIDE + (uint32_t)r, vc);
                            if (x[q][r][0]) delta[o + t] = 0;
                            if (x[q][r][1]) delta[o + vc + t] = 0;
                            sl += x[q][r][0];
                            sr += x[q][r][1];
                        }
                    } else {  // (pairs with thousands of sites: all CH_RSTRIDE replicas, one pair at a time)
                        constexpr int YB = OWNED ? CH_RSTRIDE / 2 : CH_RSTRIDE;
#pragma unroll
                        for (int r0 = 0; r0 < CH_RSTRIDE; r0 += YB) {
                            uint32_t y[YB][2];
#pragma unroll
                            for (int r = 0; r < YB; r++) {
                                const size_t o = delta_rep_off(p * (uint32_t)CH_RSTRIDE + (uint32_t)(r0 + r), vc);
                                y[r][0] = live ? dld(&delta[o + t]) : 0u;
                                y[r][1] = live ? dld(&delta[o + vc + t]) : 0u;
                            }
#pragma unroll
                            for (int r = 0; r < YB; r++) {
                                const size_t o = delta_rep_off(p * (uint32_t)CH_RSTRIDE + (uint32_t)(r0 + r), vc);
                                if (y[r][0]) delta[o + t] = 0;
                                if (y[r][1]) delta[o + vc + t] = 0;
                                sl += y[r][0];
                                sr += y[r][1];
                            }
                        }
                    }
                    const uint32_t dr = sr + (t == a ? adj : 0u), ir = sr + (t == Z ? adj : 0u);
                    if (sl) {
                        // (the value before: decides whether row t's maximum may have moved.  Rows of the batch's own
                        // tokens are flagged anyway, every other row is touched by this thread alone)
                        const uint32_t old = atomicSub(&mat[(size_t)t * stride + a], sl);
                        atomicAdd(&mat[(size_t)t * stride + Z], sl);
                        flagged |= old == rm.x;
                    }
                    if (dr) atomicSub(&mat[(size_t)b * stride + t], dr);
                    if (ir) atomicAdd(&mat[(size_t)Z * stride + t], ir);
                    if (live && t == b) mat[(size_t)a * stride + b] = 0;
                    flagged |= live && ((t == a) | (t == b) | (t == Z));
                }
            }
        }
        if (OWNED) {
            const unsigned long long fb = __ballot(flagged);
            const uint32_t mine = (lane_id() < 32) ? (uint32_t)fb : (uint32_t)(fb >> 32);
            const uint32_t neww = oldword | mine;
            if ((lane_id() & 31) == 0 && (ran || neww != oldword)) dbits[t >> 5] = neww;
        } else if (flagged && !prevflag) {
            atomicOr(&dbits[t >> 5], 1u << (t & 31));
        }
        // ---- what the next FULL selection needs: the largest maximum of my wave's 64 rows (flagged rows left
        // out: they are re-scanned), the first row that attains it, that row's arg, how many rows attain it
        const uint32_t vs = (!live || flagged || prevflag) ? 0u : rm.x;
        const uint32_t gw = t >> 6, base = gw * 64u;
        const int lane = lane_id();
        const uint32_t m = wave_umax_dpp(vs);
        const unsigned long long bal = __ballot(m != 0 && vs == m);
        const int fl = bal ? __ffsll((long long)bal) - 1 : 0;
        const uint32_t arg = (uint32_t)__shfl((int)rm.y, fl);
        if (lane == 0) sums[gw] = make_uint4(m, base + (uint32_t)fl, arg, (uint32_t)__popcll(bal));
    }
}
// the step's records (the first 64 threads of ONE workgroup call)
// (k_step: the workgroup that SELECTED calls -- what the selection left in st is then its own workgroup's writes; the
// removal counters, filled by every workgroup's device atomics earlier in the same launch, are read at agent scope)
__device__ __forceinline__ void apply_chain_records(DevState *st, int par, IterRec *rec, StepRec *srec, uint32_t step,
                                                    uint32_t *__restrict__ removed, const uint32_t K, const bool noop,
                                                    const uint32_t status, const uint32_t defer, const uint32_t remote) {
    {
        // ids removed by the merge pass: CH_RMV counters per pair of the batch, one per 256-byte line (lane l: counters
        // 4l .. 4l + 3, all of pair l / 4)
        // (removed == nullptr: an unweighted stream -- a merge of a != b removes exactly as many ids as the pair counts, base.py:25-41:
        // nobody counted, the batch's counts are the answer)
        uint32_t v = 0;
        if (removed) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t x = ld_agent(&removed[(threadIdx.x * 4 + i) * REMOVED_STRIDE]);
                if (x) removed[(threadIdx.x * 4 + i) * REMOVED_STRIDE] = 0;
                v += x;
            }
        }
        static_assert(CH_RMV == 16, "lanes 4p .. 4p + 3 hold the removals of pair p");
        v += (uint32_t)__shfl_xor((int)v, 1);
        v += (uint32_t)__shfl_xor((int)v, 2);
        uint32_t rem[CH_KMAX];
#pragma unroll
        for (int p = 0; p < CH_KMAX; p++) rem[p] = (uint32_t)__shfl((int)v, 4 * p);
        if (K == 1) {  // (a pair merged alone spreads over all the counters)
            uint32_t tot = 0;
#pragma unroll
            for (int p = 0; p < CH_KMAX; p++) tot += rem[p];
            rem[0] = tot;
        }
        if (!removed) {
#pragma unroll
            for (int p = 0; p < CH_KMAX; p++) rem[p] = st->bcnt[p];
        }
        if (threadIdx.x == 0) {
            const unsigned long long n = st->n[par];
            unsigned long long nn = n;
            const uint32_t iter = st->iter;
            const uint32_t mode_used = st->sel_mode;  // (CH_FULL / CH_LIST: left from the list selection; the pool does not use it)
            uint32_t k_done = 0;
            if (!noop) {
                for (uint32_t p = 0; p < K; p++) {
                    nn -= rem[p];
                    iter_rec_put(rec + iter + p, st->ba[p], st->bb[p], st->bcnt[p], ST_OK, nn);
                }
                k_done = K;
                st->iter = iter + K;
                // the list minus the batch: anything left -> the next step takes its pairs off it
                st->sel_mode = (st->tl_n > st->tl_skip) ? CH_LIST : CH_FULL;  // (tl_skip: the batch's share of the list)
            } else if (status == 0 && !defer) {
                st->sel_mode = CH_FULL;  // (an emptied list, or training is over: select when asked again)
            }
            if (status == 0) st->n[par ^ 1] = nn;  // (a step that merged nothing carries the length forward)
            if (remote && st->status == 0) st->status = ST_INTERNAL;
            st->removed = 0;
            st->pool_hint = st->pool_hint_next;  // (k_pool.hip: what this step's selection announced for the next one)
            StepRec *sr = srec + (step % STEP_RING);
            // (pad: defer 1 = a == b heads the list, 2 = a tie the step could not settle)
            step_rec_put(sr, iter, k_done, (status == 0 && defer) ? (uint32_t)ST_DEFER : status, mode_used | (defer << 8), nn);
            // ONE wait for everything above to be acknowledged, then the sequence words (the host waits on the step's, then on
            // each merge's: every record it reads is final once its own sequence word shows)
            __builtin_amdgcn_s_waitcnt(0);
            for (uint32_t p = 0; p < k_done; p++) iter_rec_seal(rec + iter + p, (unsigned long long)(iter + p) + 1);
            step_rec_seal(sr, (unsigned long long)step + 1);
        }
    }
}
// staged headers: smask[w] bit s = slot 32*w + s has a new header in stage[32*w + s]; this thread takes mask words
// first, first + stp, ...
template <bool THROUGH = false>
__device__ __forceinline__ void apply_chain_commit(uint32_t first, uint32_t stp, uint32_t nwords, uint32_t *__restrict__ smask,
                                                   const StageRec *__restrict__ stage, SlotHdr *__restrict__ hdr_cur) {
    for (uint32_t w = first; w < nwords; w += stp) {
        uint32_t m = THROUGH ? ld_agent(&smask[w]) : smask[w];  // (set by device atomics; in k_step by other workgroups of this launch)
        if (!m) continue;
        smask[w] = 0;
        // four headers in flight at a time: a mid-training sweep changes half the slots of a word, and with one staged
        // header loaded, waited for and stored per turn the commit was 17 dependent round trips -- 50 of a step's 60 us of
        // table update (SQ counters per phase, profiles/r6_notes.md)
        while (m) {
            uint32_t tt[4], h[4][8];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                tt[i] = m ? w * 32 + (uint32_t)__ffs((int)m) - 1u : 0xFFFFFFFFu;
                m &= m - 1u;  // (0 stays 0)
                if (tt[i] != 0xFFFFFFFFu) stage_get<THROUGH>(stage + tt[i], h[i]);
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (tt[i] == 0xFFFFFFFFu) continue;
                uint4 *dst = reinterpret_cast<uint4 *>(hdr_cur + tt[i]);
                dst[0] = make_uint4(h[i][0], h[i][1], h[i][2], h[i][3]);
                dst[1] = make_uint4(h[i][4], h[i][5], h[i][6], h[i][7]);
            }
        }
    }
}
// Workgroups [0, na): one token per thread, every pair of the batch.  Workgroups [na, grid): commit the staged headers;
// the first of them also makes the stream length, the iteration records of the step's merges, the step record, and the
// next step's mode.
__global__ void __launch_bounds__(256)
k_apply_chain(uint32_t *__restrict__ mat, uint32_t stride, uint32_t *__restrict__ delta, uint32_t vcap,
              const uint32_t *__restrict__ rowmax, DevState *st, uint32_t *__restrict__ dbits, int par, IterRec *rec,
              StepRec *srec, uint32_t step, uint32_t na, SlotHdr *__restrict__ hdr_cur, const StageRec *__restrict__ stage,
              uint32_t *__restrict__ removed, uint32_t *__restrict__ smask, uint32_t nwords, uint4 *__restrict__ sums,
              const uint32_t *__restrict__ folded, uint32_t fS, const uint32_t *__restrict__ ftail) {
    // folded != nullptr (sharded training): the batch's delta is the all-reduced payload of k_dp_fold_chain -- pair p's
    // SL at folded[2p fS ..), SR at folded[(2p + 1) fS ..), its adj in ftail[p] -- instead of this rank's replica blocks
    // (sharded: ftail[16] = the number of ranks whose status was raised when they folded this step's delta -- a
    // failure inside any rank's merge pass stops every rank at this same merge)
    __shared__ uint32_t s_w[64], s_pairs[CH_KMAX];
    step_words_fetch(st, s_w);
    const uint32_t remote = (folded && ftail[16] != 0) ? 1u : 0u;
    const uint32_t status = s_w[SW_STATUS] ? s_w[SW_STATUS] : (remote ? ST_INTERNAL : 0u), defer = s_w[SW_DEFER];
    const uint32_t K = s_w[SW_BK], z0 = s_w[SW_BZ0];
    const bool noop = status || defer || K == 0;
    if (blockIdx.x < na) {
        if (noop) return;
        if (threadIdx.x < CH_KMAX) s_pairs[threadIdx.x] = (s_w[threadIdx.x] << 16) | (s_w[16 + threadIdx.x] & 0xFFFFu);
        if (threadIdx.x == 0 && K == 1) s_w[32] = s_w[SW_ADJ];  // (a batch of one: merge_ab_wave's adj word)
        __syncthreads();
        apply_chain_tokens<false>(blockIdx.x * 256u + threadIdx.x, mat, stride, delta, vcap, rowmax, dbits, sums, folded, fS,
                                  ftail, K, z0, 0u, s_pairs, s_w + 32, s_w[SW_BREP]);
        return;
    }
    if (blockIdx.x == na && threadIdx.x < 64) apply_chain_records(st, par, rec, srec, step, removed, K, noop, status, defer, remote);
    if (noop) return;
    apply_chain_commit((blockIdx.x - na) * blockDim.x + threadIdx.x, (gridDim.x - na) * blockDim.x, nwords, smask, stage, hdr_cur);
}

// ---------------------------------------------------------------------------
// Sharded chain steps (api_dp.hip: dp_train_loop).  Per step: k_pool_sel (dpkey; k_pool.hip) -> MIN all-reduce ->
// k_pool_sel_dp -> k_merge_chain -> k_dp_fold_chain -> SUM all-reduce -> k_apply_chain (folded payload, status word).
// k_dp_fold_chain: this rank's delta of the step's batch, folded over its replica blocks into the SUM payload:
// pair p's SL at folded[2p S ..), its SR at folded[(2p + 1) S ..), S = the payload's vector stride (>= every id in
// use + 1); tail = folded + 2 kcap S: [p] = format B's adj of pair p (p < 16), [16] = 1 if this rank's status is raised (the
// sum tells every rank before the table update: all ranks stop at the same merge).
__global__ void __launch_bounds__(256)
k_dp_fold_chain(uint32_t *__restrict__ delta, uint32_t dl, const DevState *__restrict__ st, uint32_t *__restrict__ folded,
                uint32_t S, uint32_t *__restrict__ tail) {
    const uint32_t status = st->status, K = st->bk, z0 = st->bz0;
    const bool noop = status || st->defer || K == 0;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t < 32) tail[t] = t == 16 ? (status ? 1u : 0u) : ((!noop && t < K && t < 16) ? (K == 1 ? st->adj : st->badj[t]) : 0u);
#ifdef BPE_DP_DEBUG
    if (t >= 32 && t < 64) {  // (debug builds: the step's selection state rides in the payload's padding)
        uint32_t v = 0;
        if (t == 32) v = st->iter;
        else if (t == 33) v = st->sel_mode;
        else if (t == 34) v = K;
        else if (t == 35) v = st->tl_n;
        else if (t == 36) v = st->tl_M;
        else if (t == 37) v = st->tl_skip;
        else if (t == 38) v = st->defer;
        else if (t == 39) v = st->gap;
        else if (t < 48) v = (uint32_t)st->ba[t - 40] << 16 | (uint32_t)st->bb[t - 40];
        else v = (uint32_t)st->chain[2 * (t - 48)] << 16 | (uint32_t)st->chain[2 * (t - 48) + 1];
        tail[t] = v;
    }
#endif
    if (noop || t > z0 + K - 1u || t >= S) return;
    const uint32_t nrep = 1u << (dl >> 24), vc = dl & 0xFFFFFFu;
    if (K == 1) {
        uint32_t sl = 0, sr = 0;
        for (uint32_t r = 0; r < nrep; r++) {
            const size_t o = delta_rep_off(r, vc);
            const uint32_t x = delta[o + t], y = delta[o + vc + t];
            if (x) delta[o + t] = 0;
            if (y) delta[o + vc + t] = 0;
            sl += x;
            sr += y;
        }
        folded[t] = sl;
        folded[(size_t)S + t] = sr;
        return;
    }
    const uint32_t brep = st->brep;
    for (uint32_t p = 0; p < K; p++) {
        // (every replica's two words in flight before the first is looked at: the loop with the clearing stores inside
        // was a chain of round trips, 16 us per step at K = 8)
        uint32_t x[CH_RSTRIDE][2];
#pragma unroll
        for (int r = 0; r < CH_RSTRIDE; r++) {
            const size_t o = delta_rep_off(p * (uint32_t)CH_RSTRIDE + (uint32_t)r, vc);
            x[r][0] = (uint32_t)r < brep ? delta[o + t] : 0u;
            x[r][1] = (uint32_t)r < brep ? delta[o + vc + t] : 0u;
        }
        uint32_t sl = 0, sr = 0;
#pragma unroll
        for (int r = 0; r < CH_RSTRIDE; r++) {
            const size_t o = delta_rep_off(p * (uint32_t)CH_RSTRIDE + (uint32_t)r, vc);
            if (x[r][0]) delta[o + t] = 0;
            if (x[r][1]) delta[o + vc + t] = 0;
            sl += x[r][0];
            sr += x[r][1];
        }
        folded[(size_t)(2 * p) * S + t] = sl;
        folded[(size_t)(2 * p + 1) * S + t] = sr;
    }
}
// general iterations of the sharded loop (bpe_dp_apply): after the SUM all-reduce -- some rank's merge pass failed -> nobody
// applies this merge (a chain step's table update reads the word itself)
__global__ void k_dp_after_sum(DevState *st, const uint32_t *__restrict__ tail) {
    if (tail[8] != 0 && st->status == 0) st->status = ST_INTERNAL;
}

// ---------------------------------------------------------------------------
// ENCODE AS A REPLAY OF TRAINING (api_encode.hip: one giant chunk, BasicTokenizer.encode, basic.py:57-74).  The reference's
// encode merges the lowest-ranked pair present until none is left; a pair created by merge r ranks above r, so
// that is the merges applied in rank order -- which is what the training loop does with its own selections.  Here the
// selection is GIVEN: merge r is pairs[r] whatever its count (zero sites: nothing happens), and everything else -- slots,
// the inverted index, sparse sweeps, batches of token-disjoint pairs in one sweep, the table update that keeps the counts
// a sparse pass is planned by -- is the training engine's.
// k_forced_sel: a chain step's batch = the longest run of the next merges with a != b, no shared token, and no token that
// the run itself creates (<= kcap).
__global__ void __launch_bounds__(64)
k_forced_sel(DevState *st, const int32_t *__restrict__ pairs, const uint32_t *__restrict__ mat, uint32_t stride, uint32_t kcap) {
    const uint32_t lane = threadIdx.x;
    const uint32_t status = st->status, defer = st->defer, iter = st->iter, nm = st->num_merges;
    if (status || defer) return;
    if (iter >= nm) {  // the merge list is used up: this step and the ones behind it do nothing
        if (lane == 0) st->bk = 0;
        return;
    }
    const uint32_t r = iter + lane;
    const bool in = lane < kcap && r < nm;
    const uint32_t a = in ? (uint32_t)pairs[2 * r] : 0xFFFFFFFEu, b = in ? (uint32_t)pairs[2 * r + 1] : 0xFFFFFFFDu;
    // (a pair made of a token that an earlier merge of this very run creates has no site yet: it ends the run too)
    uint32_t bad = in ? ((a == b) | (a >= 256u + iter) | (b >= 256u + iter)) : 1u;
#pragma unroll
    for (int j = 0; j < CH_KSWEEP - 1; j++) {
        const uint32_t aj = (uint32_t)__shfl((int)a, j), bj = (uint32_t)__shfl((int)b, j);
        if ((uint32_t)j < lane) bad |= (aj == a) | (aj == b) | (bj == a) | (bj == b);
    }
    const unsigned long long bb = __ballot(bad != 0);
    const uint32_t K = bb ? min(kcap, (uint32_t)__ffsll((long long)bb) - 1u) : kcap;
    const uint32_t c = lane < K ? mat[(size_t)a * stride + b] : 0u;
    const uint32_t cmax = wave_umax_dpp(c);
    __shared__ uint32_t s_pa[CH_KMAX];
    if (lane < CH_KMAX) s_pa[lane] = a;
    __syncthreads();
    const uint32_t hm = chain_hash_find(s_pa, K);
    if (lane == 0) {
        st_agent(&st->adj, 0u);
        st->count = c;
        st->ntied = 1;
        st->firstpos = NOPOS;
        st->sel_tie = 0;
        st->a = (int32_t)a;
        st->b = (int32_t)b;
        st->fin_a = (int32_t)a;
        st->fin_b = (int32_t)b;
        st->bk = K;
        st->bz0 = 256u + iter;
        st->tl_n = st->tl_skip = 0;
        st->dp_wait = 0;
        st->sel_mode = CH_LIST;
        st->found = K ? 1u : 0u;
        if (K == 0) st->defer = 1;  // a == b heads the run: the general path's merge (lane 0's tokens are older than its own merge: never the other reason)
        st->brep = cmax > CH_REP_COUNT ? (uint32_t)CH_RSTRIDE : (uint32_t)CH_REP;
        st->bhm = hm;
        st->bhm_key = ((256u + iter) << 8) | K;
        st->pool_n = 0;
    }
    if (lane < K) {
        st->ba[lane] = (int32_t)a;
        st->bb[lane] = (int32_t)b;
        st_agent(&st->badj[lane], 0u);
        st->bcnt[lane] = c;
    }
}
// k_forced_pair: the general path's selection (the first merge, every a == b merge): merge `iter` is pairs[iter]
__global__ void k_forced_pair(DevState *st, const int32_t *__restrict__ pairs, uint32_t iter, const uint32_t *__restrict__ mat,
                              uint32_t stride) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    st->adj = 0;
    st->chain_n = 0;
    if (st->status) return;
    const int32_t a = pairs[2 * iter], b = pairs[2 * iter + 1];
    st->a = a;
    st->b = b;
    st->fin_a = a;
    st->fin_b = b;
    st->count = mat[(size_t)(uint32_t)a * stride + (uint32_t)b];
    st->ntied = 1;
    st->firstpos = NOPOS;
    st->sel_tie = 0;
    st->found = 1;
    st->ncand = 0;
}

// host: entering chain steps after general iterations (the device counts the merges from here on)
__global__ void k_set_iter(DevState *st, uint32_t iter, uint32_t num_merges) {
    st->iter = iter;
    st->num_merges = num_merges;
    st->sel_mode = CH_FULL;
    st->tl_n = st->tl_skip = 0;
    st->bk = 0;
    st->scan_a = st->scan_b = st->scan_z = NOROW;  // (rows to re-scan are named by the flag words alone)
    st->chain_n = 0;
    st->dp_wait = 0;
    st->pool_n = 0;  // (k_pool.hip: the first selection gathers the pool)
    st->pool_hint = st->pool_hint_next = 1;
    st->pool_epoch = 0;
}
// host: a deferred chain step is about to be re-run through the general path
__global__ void k_clear_defer_chain(DevState *st) {
    st->defer = 0;
    st->chain_n = 0;
    st->sel_mode = CH_FULL;
    st->tl_n = st->tl_skip = 0;
    st->bk = 0;
    st->dp_wait = 0;
    st->pool_n = 0;  // (the general path's merge is not one the pool was maintained for)
    st->pool_hint = st->pool_hint_next = 1;
}

}  // namespace BPE_G
}  // namespace bpe
