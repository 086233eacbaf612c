// k_slots2.hip -- K3 in the training loop, second form: in-place slotted merge for a != b
// (dense and sparse passes), the a == b pass, the inverted slot index, re-packing.
// Part of bpe_kernels.hip, which includes the parts in order.
// (no include guard: bpe_kernels.hip includes this part once per geometry, namespace BPE_G)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../bpe_device.h"

namespace bpe {
namespace BPE_G {

// ---------------------------------------------------------------------------
// The stream is a sequence of T slots of TILE2 = 1024 words; slot t holds `len` ids at the start
// of its home in buffer 0 or 1 (SlotHdr.meta = len | buffer << 31).  Stream order is slot order,
// so first-occurrence order (F3) is that of slot-space positions t * TILE2 + offset.
//
// ONE WAVE owns a slot: lane l holds ids [4l, 4l+4) of each of four 256-id stripes (every global
// access a full 1 KiB wave access), all cross-lane traffic is DPP / readlane, and there is no
// barrier anywhere in an a != b pass.
//
// A merge of (a, b), a != b, touches a slot only where `a` is directly followed by `b`:
//   * the slot that owns the `a` of a site gets the new token there, and OWES the whole
//     pair-table update of that site (delta format B below), whichever slots the words around
//     the site live in -- two words of left context and three of right context come from the
//     neighbours' headers;
//   * a slot whose first word is the `b` of a site that started in the previous slot drops it.
// Every other slot is untouched and owes nothing.  A changed slot is compacted IN PLACE: the
// wave holds all 1024 words in registers before it stores, the output is staged in LDS and
// written back with 16-byte stores, from the first changed word on; nobody else reads a slot's
// words during an a != b pass (neighbours read HEADERS, which are double-buffered or staged).
//   * dense pass (k_merge_ab_dense): one wave per slot, headers ping-pong between two arrays;
//   * sparse pass (k_merge_ab_sparse): a resident grid whose waves work through the candidate
//     list k_select made from the inverted index; new headers go to a staging area that the
//     table-update kernel commits (the header array must stay intact while neighbours read it).
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


__global__ void __launch_bounds__(256)
k_slot2_init(SlotHdr *__restrict__ hdr, uint64_t T, DevState *st, int par, uint32_t which,
             const uint32_t *__restrict__ ids) {
    const uint64_t n = st->n[par];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->gap = 0;
        st->tlive = (uint32_t)min(T, (n + TILE2 - 1) / TILE2);
    }
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += stride) {
        const uint64_t b0 = t * TILE2;
        const uint32_t len = b0 >= n ? 0u : (uint32_t)min((uint64_t)TILE2, n - b0);
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
// empty neighbours; rare); also which slots they come from
__device__ __forceinline__ void slot_context_walk(const SlotHdr *__restrict__ hdr, uint32_t t, uint32_t T,
                                                  uint32_t *ctx) {
    // (no indexed local arrays: they would live in scratch memory, and every kernel that inlines this would carry
    // a private segment for a path that runs once in a blue moon)
    uint32_t h0 = INVALID_WORD, h1 = INVALID_WORD, h2 = INVALID_WORD;
    uint32_t tn = 0xFFFFFFFFu;  // the next non-empty slot
    int got = 0;
    for (uint32_t u = t + 1; u < T && got < 3; u++) {
        const SlotHdr hh = hdr[u];
        const uint32_t lu = hh.meta & 0x7FFFFFFFu;
        if (lu && tn == 0xFFFFFFFFu) tn = u;
#pragma unroll
        for (uint32_t i = 0; i < 3; i++) {
            if (i < lu && got < 3) {
                const uint32_t w = i == 0 ? hh.w0 : (i == 1 ? hh.w1 : hh.w2);
                if (got == 0) h0 = w;
                else if (got == 1) h1 = w;
                else h2 = w;
                got++;
            }
        }
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
    ctx[0] = h0;
    ctx[1] = h1;
    ctx[2] = h2;
    ctx[3] = p2;
    ctx[4] = p1;
    ctx[5] = tp;
    ctx[6] = tn;
}

__device__ __forceinline__ uint32_t bcast(uint32_t v, int srclane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, srclane);
}

// One slot of an a != b pass, by ONE WAVE.  `out` = this wave's 1024 words of LDS staging.
// INDEXED: the inverted slot index is live and learns the pairs this pass creates (costs ~10
// VGPRs; the early dense passes, which run before the index exists, use the variant without).
// LDSD: every id is below LDSD_CAP and the delta goes into the workgroup's LDS tables `sd`
// (SL[LDSD_CAP] | SR[LDSD_CAP] | adj | removed), flushed by the kernel when its slots are done.
template <bool SPARSE, bool INDEXED, bool LDSD, bool THROUGH = false>
__device__ __forceinline__ void merge_ab_wave(uint32_t *__restrict__ out, uint32_t *__restrict__ sd, const uint32_t t,
                                              const AbArgs &A, const uint32_t a, const uint32_t b,
                                              const uint32_t tl_in = 0xFFFFFFFFu) {
    const int lane = lane_id();
    // no slot from Tl on holds anything (tl_in: the caller read st->tlive once for all its slots)
    const uint32_t Tl = tl_in != 0xFFFFFFFFu ? tl_in : min(A.T, A.st->tlive);
    // ---- (1) every load that does not depend on another one -------------------------------
    // the slot itself, speculatively from buffer 0 (a slot lives in buffer 1 only between an
    // a == b pass that rewrote it and the next re-packing), all TILE2 words whatever its length;
    // the headers of slots t-1, t, t+1: six 16-byte pieces, lanes 0..5
    const uint32_t *src = A.b0 + (size_t)t * TILE2;
    uint4 rv[MJ];
#pragma unroll
    for (int j = 0; j < MJ; j++) rv[j] = *reinterpret_cast<const uint4 *>(src + j * 256 + lane * 4);
    uint4 hv = (lane & 1) ? make_uint4(INVALID_WORD, INVALID_WORD, 0u, 0u)
                          : make_uint4(INVALID_WORD, INVALID_WORD, INVALID_WORD, 0u);
    {
        const long long hi = 2ll * (long long)t - 2 + lane;
        if (lane < 6 && hi >= 0 && hi < 2ll * (long long)A.T) hv = reinterpret_cast<const uint4 *>(A.hdr_in)[hi];
    }
    const uint32_t meta = bcast(hv.w, 2);
    const uint32_t len = meta & 0x7FFFFFFFu, buf = meta >> 31;
    auto keep_header = [&]() {  // dense: the slot stays as it is (lanes 2 and 3 hold its header)
        if (!SPARSE && (lane == 2 || lane == 3)) reinterpret_cast<uint4 *>(A.hdr_out)[2 * (size_t)t + (lane - 2)] = hv;
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
    uint32_t tprev = t - 1;  // the slot that owns the word before mine (t == 0: none, and prev1 is invalid)
    uint32_t tnext = t + 1;  // ... and the words after mine
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
    uint32_t tail[3];  // positions TILE2 .. TILE2+2: beyond any slot's own words
#pragma unroll
    for (int i = 0; i < 3; i++) tail[i] = at(TILE2 + i, 0u);
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
        const int nb = (int)len - (j * 256 + lane * 4);
        valid[j] = nb >= 4 ? 0xFu : (nb <= 0 ? 0u : ((1u << nb) - 1u));
        rb[j] = r;
        anyr |= r & valid[j];
    }
    // carry: my first word is the `b` of a site that starts at the previous slot's last word
    const uint32_t s = (uint32_t)((prev1 != INVALID_WORD) & ((prev1 & IDMASK) == a) & ((first & NWMASK) == b));
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
        // r of the position before my group: the previous lane's bit 3, the previous stripe's
        // last lane, or (first group of the slot) the carry
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
    // (everything before the first site keeps its place and value: only the rest is stored)
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
                out[o++] = ((mb[j] >> k) & 1u) ? (A.newid | (w & (FLAG | WMASK))) : w;
            }
        }
    }
    // (the wave's LDS operations complete in issue order: its own reads below see these writes)
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
        if (!SPARSE) {
            reinterpret_cast<uint4 *>(A.hdr_out)[2 * (size_t)t] = make_uint4(h[0], h[1], h[2], h[3]);
            reinterpret_cast<uint4 *>(A.hdr_out)[2 * (size_t)t + 1] = make_uint4(h[4], h[5], 0u, 0u);
        } else {
            stage_put<THROUGH>(A.stage + t, t, h);
            atomicOr(&A.smask[t >> 5], 1u << (t & 31));
        }
        if (LDSD) atomicAdd(&sd[2 * LDSD_CAP + 1], len - total);
        else if (A.removed) atomicAdd(&A.removed[(t & 255u) * REMOVED_STRIDE], len - total);  // (nullptr: a chain step of an unweighted
                                                                                              // stream -- the ids removed ARE the pair's count)
        if (total < 3 && t + 1 < Tl) A.st->gap = 1;
    }
    // ---- (7) pair-table delta of my sites (format B); their new pairs enter the index -----------
    if (!sites) return;  // carry only: the site belongs to the previous slot
    if (!A.delta) return;  // (experiment "exp_no_delta": time the pass without its table bookkeeping; results are wrong)
    const uint32_t nrep = 1u << (A.vcap >> 24);
    const uint32_t vc = A.vcap & 0xFFFFFFu;
    uint32_t *dl = A.delta + delta_rep_off(t & (nrep - 1), vc);  // SL
    uint32_t *dr = dl + vc;                                      // SR
    uint32_t adj = 0;
#pragma unroll
    for (int j = 0; j < MJ; j++) {
        if (!__any(mb[j] != 0)) continue;  // (uniform) no site in this stripe
        // two words before my group, three after it
        uint32_t upm2, upm1, dn0, dn1, dn2;
        if (j > 0) {
            upm2 = lane_last(x[(j + MJ - 1) % MJ][2]);
            upm1 = lane_last(x[(j + MJ - 1) % MJ][3]);
        } else {
            upm2 = prev2;
            upm1 = prev1;
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
        W[0] = (uint32_t)dpp_mov<0x138>((int)upm2, (int)x[j][2]);  // (lane 0 keeps upm2 / upm1)
        W[1] = (uint32_t)dpp_mov<0x138>((int)upm1, (int)x[j][3]);
        W[2] = x[j][0];
        W[3] = x[j][1];
        W[4] = x[j][2];
        W[5] = x[j][3];
        W[6] = lane_next(x[j][0], dn0);
        W[7] = lane_next(x[j][1], dn1);
        W[8] = lane_next(x[j][2], dn2);
        if (mb[j] == 0) continue;
        // (a lane's four words hold at most two sites: the lanes take their first site together, then their second --
        // merge_chain_wave, k_chain.hip)
        uint32_t todo = mb[j];
#pragma unroll
        for (int si = 0; si < 2; si++) {
            if (todo == 0) continue;
            const uint32_t k = (uint32_t)__ffs((int)todo) - 1u;
            todo &= todo - 1u;
            const int q = j * 256 + lane * 4 + (int)k;
            const bool k1 = (k & 1u) != 0, k2 = (k & 2u) != 0;
            auto wsel = [&](int o) -> uint32_t { return k2 ? (k1 ? W[o + 3] : W[o + 2]) : (k1 ? W[o + 1] : W[o]); };
            const uint32_t wa = wsel(2);
            const uint32_t wt = word_weight(wa);
            const uint32_t Lw = wsel(1), LL = wsel(0);
            if (!(wa & FLAG) && Lw != INVALID_WORD) {
                const bool ltail = ((LL & IDMASK) == a) & ((Lw & NWMASK) == b);
                if (!ltail) {
                    if (LDSD) atomicAdd(&sd[Lw & IDMASK], wt);
                    else atomicAdd(&dl[Lw & IDMASK], wt);
                    // the new pair (L, Z) enters the filter of the slot that holds L, and mine too if
                    // that is another slot (a boundary pair is known to both slots it touches: the
                    // one that owns its site and the one that drops the site's second word)
                    if (INDEXED) {
                        index_add(A.idx, A.istride, t, Lw & IDMASK, A.newid);
                        if (q == 0) index_add(A.idx, A.istride, tprev, Lw & IDMASK, A.newid);
                    }
                }
            }
            const uint32_t R = wsel(4), RR = wsel(5);
            if (!(R & FLAG)) {  // (INVALID_WORD has the flag bit set: end of stream)
                const bool rsite = ((R & IDMASK) == a) & ((RR & NWMASK) == b);
                if (rsite) adj += wt;
                else if (LDSD) atomicAdd(&sd[LDSD_CAP + (R & IDMASK)], wt);
                else atomicAdd(&dr[R & IDMASK], wt);
                if (INDEXED) {
                    const uint32_t y = rsite ? A.newid : (R & IDMASK);
                    index_add(A.idx, A.istride, t, A.newid, y);
                    if (q + 2 >= (int)len) index_add(A.idx, A.istride, tnext, A.newid, y);
                }
            }
        }
    }
    if (__any(adj != 0)) {
        adj = wave_sum_u32(adj);
        if (lane == 0) {
            if (LDSD) atomicAdd(&sd[2 * LDSD_CAP], adj);
            else atomicAdd(&A.st->adj, adj);
        }
    }
}

// LDSD kernels: clear the tables / add them to this workgroup's replica of the delta vectors
__device__ __forceinline__ void ldsd_clear(uint32_t *sd) {
    for (uint32_t i = threadIdx.x; i < 2 * LDSD_CAP + 2; i += MT) sd[i] = 0;
    __syncthreads();
}
__device__ __forceinline__ void ldsd_flush(const uint32_t *sd, const AbArgs &A) {
    __syncthreads();
    const uint32_t nrep = 1u << (A.vcap >> 24);
    const uint32_t vc = A.vcap & 0xFFFFFFu;
    if (A.delta) {
        uint32_t *g = A.delta + delta_rep_off(blockIdx.x & (nrep - 1), vc);
        const uint32_t lim = min(vc, (uint32_t)LDSD_CAP);
        for (uint32_t i = threadIdx.x; i < lim; i += MT) {
            const uint32_t l = sd[i], r = sd[LDSD_CAP + i];
            if (l) atomicAdd(&g[i], l);
            if (r) atomicAdd(&g[vc + i], r);
        }
    }
    if (threadIdx.x == 0) {
        const uint32_t adj = sd[2 * LDSD_CAP], rem = sd[2 * LDSD_CAP + 1];
        if (adj) atomicAdd(&A.st->adj, adj);
        if (rem && A.removed) atomicAdd(&A.removed[(blockIdx.x & 255u) * REMOVED_STRIDE], rem);
    }
}

// dense a != b pass: one wave per slot (four per workgroup)
template <bool INDEXED>
__global__ void __launch_bounds__(MT, INDEXED ? 5 : 7)
k_merge_ab_dense(AbArgs A) {
    __shared__ __attribute__((aligned(16))) uint32_t s_out[MT / 64][TILE2];
    const DevState *st = A.st;
    if (blockIdx.x == 0 && threadIdx.x == 0) *A.dirty_n = 0;
    const uint32_t t = blockIdx.x * (MT / 64) + wave_id();
    if (t >= A.T || st->status) return;
    if (!st->found) {
        if (t == 0 && lane_id() == 0) A.st->status = ST_INTERNAL;
        return;
    }
    const uint32_t a = (uint32_t)st->a, b = (uint32_t)st->b;
    if (a == b) return;  // k_merge_aa's pass
    merge_ab_wave<false, INDEXED, false>(s_out[wave_id()], nullptr, t, A, a, b);
}
// ... while every id is below LDSD_CAP: a resident grid (five workgroups per CU), wave w of the
// grid takes slots w, w + waves, ...; the delta goes through LDS
template <bool INDEXED>
__global__ void __launch_bounds__(MT, 5)
k_merge_ab_dense_early(AbArgs A) {
    __shared__ __attribute__((aligned(16))) uint32_t s_out[MT / 64][TILE2];
    __shared__ uint32_t s_delta[2 * LDSD_CAP + 2];
    const DevState *st = A.st;
    if (blockIdx.x == 0 && threadIdx.x == 0) *A.dirty_n = 0;
    if (st->status) return;
    if (!st->found) {
        if (blockIdx.x == 0 && threadIdx.x == 0) A.st->status = ST_INTERNAL;
        return;
    }
    const uint32_t a = (uint32_t)st->a, b = (uint32_t)st->b;
    if (a == b) return;
    ldsd_clear(s_delta);
    const uint32_t nw = gridDim.x * (MT / 64);
    const uint32_t Tl = min(A.T, st->tlive);
    for (uint32_t t = blockIdx.x * (MT / 64) + wave_id(); t < A.T; t += nw)
        merge_ab_wave<false, INDEXED, true>(s_out[wave_id()], s_delta, t, A, a, b, Tl);
    ldsd_flush(s_delta, A);
}

// sparse a != b pass: a resident grid whose waves work through the candidate list that the
// deciding block of k_select made from the index (st->ncand slots in A.cand: every slot whose
// filter admits the pair, plus the slots an a == b pass rewrote since the index was built; all
// slots while st->gap is up) -- evenly dealt, wave by wave.
#ifndef AB_SPARSE_WAVES
#define AB_SPARSE_WAVES 4
#endif
template <bool LDSD>
__global__ void __launch_bounds__(MT, AB_SPARSE_WAVES)
k_merge_ab_sparse(AbArgs A) {
    __shared__ __attribute__((aligned(16))) uint32_t s_out[MT / 64][TILE2];
    __shared__ uint32_t s_delta[LDSD ? 2 * LDSD_CAP + 2 : 1];
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
    const uint32_t nw = gridDim.x * (MT / 64);
    if (LDSD) {
        if (blockIdx.x * (MT / 64) >= n) return;  // (uniform) none of my waves has a slot
        ldsd_clear(s_delta);
    }
    const uint32_t Tl = min(A.T, st->tlive);
    for (uint32_t i = blockIdx.x * (MT / 64) + wave_id(); i < n; i += nw)
        merge_ab_wave<true, true, LDSD>(s_out[wave_id()], s_delta, A.cand[i], A, a, b, Tl);
    if (LDSD) ldsd_flush(s_delta, A);
}

// ---------------------------------------------------------------------------
// a == b pass (runs only when the decided pair has a == b; a resident grid of single-wave
// workgroups striding over ALL slots).  The first form's algorithm (k_slots.hip: run-parity carry
// found by walking the run back through the previous slots, out-of-place rewrite, format A delta,
// every pair charged to its left element) on the 32-byte headers and one-wave slots: the tile
// helpers of k_merge.hip with a single wave (its span IS the slot).

__device__ __forceinline__ void merge_aa_tile(const uint32_t t, const AaArgs &A, const uint32_t a, const uint32_t Tl) {
    __shared__ int s_wave[MT / 64];
    __shared__ uint32_t s_wsum[MT / 64];
    __shared__ uint32_t s_ctx[8];   // halo0..2, prev2, prev1, carry
    __shared__ uint32_t s_hdr[8];   // my header (copied, then rewritten by the tile)
    const uint32_t b = a;
    // the headers of slots t - 1, t, t + 1 as six 16-byte pieces on lanes 0..5, ONE round trip (merge_ab_wave's scheme: the
    // pass used to read its own header, then the state's tlive, then walk to both neighbours with thread 0 -- four
    // dependent round trips per slot before the slot's words were asked for)
    uint4 hv6 = (lane_id() & 1) ? make_uint4(INVALID_WORD, INVALID_WORD, 0u, 0u) : make_uint4(INVALID_WORD, INVALID_WORD, INVALID_WORD, 0u);
    {
        const long long hi = 2ll * (long long)t - 2 + lane_id();
        if (lane_id() < 6 && hi >= 0 && hi < 2ll * (long long)A.T) hv6 = reinterpret_cast<const uint4 *>(A.hdr_in)[hi];
    }
    SlotHdr hme;
    hme.w0 = bcast(hv6.x, 2);
    hme.w1 = bcast(hv6.y, 2);
    hme.w2 = bcast(hv6.z, 2);
    hme.meta = bcast(hv6.w, 2);
    hme.l0 = bcast(hv6.x, 3);
    hme.l1 = bcast(hv6.y, 3);
    hme.pad0 = hme.pad1 = 0;
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
    const uint32_t *src = (cur ? A.b1 : A.b0) + (size_t)t * TILE2;
    const uint32_t nlen = bcast(hv6.w, 4) & 0x7FFFFFFFu, plen = bcast(hv6.w, 0) & 0x7FFFFFFFu;
    const uint32_t nh0 = bcast(hv6.x, 4), nh1 = bcast(hv6.y, 4), nh2 = bcast(hv6.z, 4), pp2 = bcast(hv6.x, 1), pp1 = bcast(hv6.y, 1);
    if (threadIdx.x == 0) {
        if ((t + 1 < Tl && nlen < 3) || (t > 0 && plen < 2)) {  // (rare) short or empty neighbours: walk the headers
            slot_context_walk(A.hdr_in, t, Tl, s_ctx);
        } else {
            s_ctx[0] = nh0;
            s_ctx[1] = nh1;
            s_ctx[2] = nh2;
            s_ctx[3] = pp2;
            s_ctx[4] = pp1;
            s_ctx[5] = t - 1;
            s_ctx[6] = t + 1;
        }
        s_hdr[0] = hme.w0;
        s_hdr[1] = hme.w1;
        s_hdr[2] = hme.w2;
        s_hdr[3] = hme.meta;
        s_hdr[4] = hme.l0;
        s_hdr[5] = hme.l1;
    }
    // (the tile helpers exchange per-wave values of a 4-wave workgroup through these: one wave here)
    if (threadIdx.x < MT / 64) {
        s_wave[threadIdx.x] = -1;
        s_wsum[threadIdx.x] = 0;
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
                const uint32_t *pu = ((mu >> 31) ? A.b1 : A.b0) + (size_t)u * TILE2;
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
        const int qw = lane_id() * 4;
#pragma unroll
        for (int j = 0; j < MJ; j++) {
            const int q0 = qw + j * 256;
            const int nb = len + 1 - q0;  // keep the bits of positions q <= len
            const uint32_t keep = nb >= 4 ? 0xFu : (nb <= 0 ? 0u : ((1u << nb) - 1u));
            anyr |= tl.rb[j] & keep;
        }
        // a full slot: the word after it is the wave's tail, not one of my registers
        if (len == TILE2) anyr |= (uint32_t)(((tl.tail[0] & IDMASK) == a) & ((tl.tail[1] & NWMASK) == b));
        if (!__syncthreads_or((int)(anyr != 0))) {
            if (threadIdx.x == 0) keep_header();
            return;
        }
    }
    tile_lzscan(tl, s_wave);
    uint32_t kept = 0;
    bool changed = false;
    uint32_t *dst = (cur ? A.w0 : A.w1) + (size_t)t * TILE2;  // the OTHER buffer
    tile_rewrite<true, true, 1>(tl, s, a, b, A.newid, dst, s_wsum, A.delta, A.vcap, len, &kept, &changed, s_hdr, A.idx,
                                A.istride, t, s_ctx[6]);
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
            atomicAdd(&A.removed[(t & 255u) * REMOVED_STRIDE], (uint32_t)len - kept);
            if (kept < 3 && t + 1 < Tl) A.st->gap = 1;
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

__global__ void __launch_bounds__(64)
k_merge_aa(AaArgs A) {
    const DevState *st = A.st;
    // (one wave per workgroup: every thread reads the same values, and barriers are wave-local)
    if (st->status || !st->found) return;  // (a missing decision is reported by the a != b kernel)
    const uint32_t a = (uint32_t)st->a;
    if (a != (uint32_t)st->b) return;
    const uint32_t Tl = min(A.T, st->tlive);  // (read once: the pass stores to *st, a load inside the loop is repeated per slot)
    if (A.cand) {  // (ascending, like the slots themselves: a wait for a predecessor's carry ends)
        const uint32_t n = st->ncand;
        for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
            merge_aa_tile(A.cand[i], A, a, Tl);
            __syncthreads();
        }
        return;
    }
    for (uint32_t t = blockIdx.x; t < A.T; t += gridDim.x) {
        merge_aa_tile(t, A, a, Tl);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// Index build: one workgroup per group of 32 slots, the group's IDX_H x 32-bit filter in LDS
// (64 KiB), three ds_or per pair, then one coalesced write of the group's row into a
// group-major scratch image; k_index_transpose turns that into the bucket-major index.  No global
// atomics.  Also clears the group's `dirty` word.
__global__ void __launch_bounds__(1024)
k_index_build(const uint32_t *__restrict__ b0, const uint32_t *__restrict__ b1, const SlotHdr *__restrict__ hdr,
              uint32_t T, uint32_t *__restrict__ scratch, uint32_t *__restrict__ dirty,
              const DevState *__restrict__ st) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_mask[];
    const uint32_t g = blockIdx.x;
    const uint32_t Tl = min(T, st->tlive);  // (the walk below must not run down an empty tail)
    for (uint32_t i = threadIdx.x; i < IDX_H; i += 1024) s_mask[i] = 0;
    if (threadIdx.x == 0) dirty[g] = 0;
    __syncthreads();
    // 1024 threads x 4 words = four slots per round, eight rounds
    for (uint32_t it = 0; it < 8; it++) {
        const uint32_t sl = it * 4 + (threadIdx.x >> 8);
        const uint32_t t = g * 32 + sl;
        if (t >= T) continue;
        const uint32_t m = hdr[t].meta;
        const uint32_t len = m & 0x7FFFFFFFu;
        const uint32_t *src = ((m >> 31) ? b1 : b0) + (size_t)t * TILE2;
        const uint32_t q = (threadIdx.x & 255u) * 4;
        if (q >= len) continue;
        const uint4 v = *reinterpret_cast<const uint4 *>(src + q);
        uint32_t x[5] = {v.x, v.y, v.z, v.w, INVALID_WORD};
        if (q + 4 < len) {
            x[4] = src[q + 4];
        } else {
            // the word after the slot: first word of the next non-empty slot (headers)
            uint32_t nxt = INVALID_WORD;
            for (uint32_t u = t + 1; u < Tl; u++) {
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
    uint32_t *row = scratch + (size_t)g * IDX_H;
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
// (one wave per slot, four slots per workgroup)
__global__ void __launch_bounds__(256)
k_slot2_compact(const uint32_t *__restrict__ b0, const uint32_t *__restrict__ b1,
                const SlotHdr *__restrict__ hdr, uint64_t T, const unsigned long long *__restrict__ off,
                uint32_t *__restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * 4 + wave_id();
    if (t >= T) return;
    const uint32_t m = hdr[t].meta;
    const uint32_t len = m & 0x7FFFFFFFu;
    const uint32_t *src = ((m >> 31) ? b1 : b0) + t * TILE2;
    uint32_t *dst = out + off[t];
    for (uint32_t i = lane_id(); i < len; i += 64) dst[i] = src[i];
}

}  // namespace BPE_G
}  // namespace bpe
