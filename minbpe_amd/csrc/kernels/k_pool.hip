// k_pool.hip -- K2 of a chain step, third form: THE POOL.
//
// Rounds 4-5's selection (k_chain_sel, retired in round 6) kept the pairs tied at the maximum as a list and needed a FULL
// selection -- every flagged row re-scanned, the row maxima read, the tied pairs located through the index -- once per count
// level, ~25 us against ~5 us for a step that took its pairs off the list; and only a FULL selection could walk below its
// level.  Late in training a level holds three or four pairs, so most steps were FULL ones and most batches were cut short by
// the end of their level, not by a shared token.
//
// The pool is the list generalised to EVERY pair that counts at least a threshold theta (tests/test_pool_model.py: the
// CPU model of exactly this, against the reference semantics of base.py:13-41, basic.py:31-42, regex.py:49-63):
//   invariant   every pair with count >= theta is an entry, with its exact count;
//   order key   epoch << 40 | first-occurrence position: positions taken in one launch (one epoch) are a valid relative
//               order for as long as the entries stay untouched; keys of different epochs are never compared;
//   a step      maintain (below) -> sort by (count descending, key) -> the levels the walk can reach that hold several
//               entries and are not CLEAN (all entries of one epoch) are located afresh through the index, all at once
//               -> the batch = the longest prefix with a != b in which no pair could chain onto a site of another -- (x, y)
//               with x a second token or y a first token of a pair before it -- and no first token comes twice; a SECOND
//               token may (round 6: sites of (a, b) and (c, b) never overlap, neither merge moves the other's count, and
//               what a merge lowers or creates -- (L, a), (b, R) and their heirs -- chains onto the batch and stops the walk
//               where it stands; tests/test_level_model.py: walking the levels from the top, inside a level in order of first
//               occurrence, under this rule is what the reference merges) -> the rest of the entries is the pool of the next step;
//   maintain    after a batch, an entry (x, y) with x the SECOND token of a batch pair (-> Zx) or y the FIRST token of
//               one (-> Zy) has its occurrences spread over (x, y), (Zx, y), (x, Zy), (Zx, Zy): each of the four that
//               counts >= theta in the updated table is an entry; the one whose count EQUALS the entry's old count
//               took over every occurrence and stands where the entry stood (it inherits the key), the others have no
//               order.  (Several batch pairs p may end in x: a variant (Z_p, y), (Z_p, Zy) for each.)  Every other entry is
//               untouched.  No pair from outside the pool can reach theta: a created pair
//               (L, Z) / (Z, R) / (Zi, Zj) counts at most what (L, a) / (b, R) / (bi, aj) counted before;
//   rebuild     pool (nearly) empty: the flagged rows are re-scanned (workgroups 1..), the deciding workgroup picks a new
//               theta from the row maxima -- the deepest of eight candidates M - (M >> s) that at most PL_ROWS rows
//               reach --, hands those rows to workgroups 1.., which gather every entry >= theta of their rows.
// A step that takes its pairs off the pool is one workgroup and two dependent round trips (entries, then the table
// words of the touched ones) plus the index look-ups of the levels that need an order; a rebuild is announced one step
// ahead (st->pool_hint, set when fewer untouched entries than `hint_below` are left) so that the scanning workgroups
// know at launch whether to stay.
// Part of bpe_kernels.hip, which includes the parts in order.
// (no include guard: bpe_kernels.hip includes this part once per geometry, namespace BPE_G)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../bpe_device.h"

namespace bpe {
namespace BPE_G {

constexpr uint32_t PL_CAP = 256;      // entries the pool holds
constexpr uint32_t PL_ROWS = 128;      // rows a rebuild hands to the scanning workgroups
constexpr uint32_t PL_GATHER = 1024;   // pairs a rebuild may gather (and 4 x PL_CAP: what maintain can make of a full pool)
static_assert(PL_GATHER >= 4 * PL_CAP, "maintain: four variants per entry");
// request / answer words of a rebuild (all self-validating: tag << 32 | value, the launch tag never repeats):
// [0] = rows to scan (0: the scanning workgroups are dismissed), [1 + j] = row j, [1 + PL_ROWS] = theta,
// [2 + PL_ROWS + w] = workgroup w has gathered its rows
constexpr uint32_t PL_REQ_WORDS = 2 + PL_ROWS + 256;

// Scratch of the helpers below.  They are written for LATENCY: a selection is one workgroup's serial work in the middle of
// every step (20 of a late step's 58 us while its loops ran one entry per thread over a level or over the whole pool, each
// iteration waiting for its LDS read, and thread 0 walked the levels alone: profiles/r6_step_stamps_*.json) -- so every
// "count the entries that rank before mine" is split over several threads per entry, and what thread 0 did in loops is a
// ballot or a prefix sum.
struct PoolScratch {
    uint32_t dl[PL_CAP];       // [level start] the level lacks an order
    uint32_t rank[PL_GATHER];  // partial ranks, summed by LDS atomics
    uint32_t w[8];             // per wave: last / first level start
    uint32_t wtot[PL_CAP / 64];
    uint32_t fv, nl, k;
};
// entries ranked by count (descending; equal counts by pair): out[rank] = in[i].  n <= PL_GATHER, every thread
// calls.  The order INSIDE a level is made later, from the keys, by a loop over the level alone.
__device__ __forceinline__ void pool_sort(const uint32_t *ixy, const uint32_t *ic, const unsigned long long *ikey,
                                          uint32_t *oxy, uint32_t *oc, unsigned long long *okey, uint32_t n, PoolScratch &X) {
    const uint32_t tid = threadIdx.x;
    uint32_t sh = 0;  // log2(threads per entry)
    while (sh < 4 && (n << (sh + 1)) <= blockDim.x) sh++;
    const uint32_t tpe = 1u << sh, e = tid >> sh, q = tid & (tpe - 1u);
    if (tid < n) X.rank[tid] = 0;
    __syncthreads();
    if (e < n) {
        const uint32_t c = ic[e], x = ixy[e];
        uint32_t r = 0;
#pragma unroll 4
        for (uint32_t j = q; j < n; j += tpe) {
            const uint32_t cj = ic[j];
            r += (cj > c) | ((cj == c) & (ixy[j] < x));  // (equal counts by pair: the same order on every rank of a sharded job)
        }
        if (r) atomicAdd(&X.rank[e], r);
    }
    __syncthreads();
    if (tid < n) {
        const uint32_t r = X.rank[tid];
        oxy[r] = ixy[tid];
        oc[r] = ic[tid];
        okey[r] = ikey[tid];
    }
    __syncthreads();
}
// (ksh: where the epoch sits in a key -- 40 on one GPU, PL_KSH_DP in a sharded job, whose positions carry the rank)
constexpr int PL_KSH = 40, PL_KSH_DP = 43, PL_POS_DP = 33;  // sharded: epoch << 43 | rank << 33 | local position (< 2^33)
// the bounds of every entry's level in a pool sorted by count (n <= PL_CAP; every thread calls; ends with a barrier): a
// level starts where the count changes -- one ballot per wave, the nearest start below / above a lane from its bits
__device__ __forceinline__ void pool_level_bounds(const uint32_t *c, uint32_t n, uint32_t *ls, uint32_t *le, PoolScratch &X) {
    const uint32_t tid = threadIdx.x, w = (uint32_t)wave_id(), lane = (uint32_t)lane_id();
    const bool in = tid < n;
    const bool start = in && (tid == 0 || c[tid - 1] != c[tid]);
    const unsigned long long bal = __ballot(start);
    if (tid < PL_CAP && lane == 0) {
        X.w[w] = bal ? w * 64u + 63u - (uint32_t)__clzll((long long)bal) : 0xFFFFFFFFu;        // the wave's last start
        X.w[4 + w] = bal ? w * 64u + (uint32_t)__ffsll((long long)bal) - 1u : 0xFFFFFFFFu;     // ... and its first
    }
    __syncthreads();
    if (in) {
        const unsigned long long upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
        const unsigned long long below = bal & upto, above = bal & ~upto;
        uint32_t lo = 0, hi = n;
        if (below) {
            lo = w * 64u + 63u - (uint32_t)__clzll((long long)below);
        } else {
            for (int v = (int)w - 1; v >= 0; v--)
                if (X.w[v] != 0xFFFFFFFFu) {
                    lo = X.w[v];
                    break;
                }
        }
        if (above) {
            hi = w * 64u + (uint32_t)__ffsll((long long)above) - 1u;
        } else {
            for (uint32_t v = w + 1; v < PL_CAP / 64; v++)
                if (X.w[4 + v] != 0xFFFFFFFFu) {
                    hi = X.w[4 + v];
                    break;
                }
        }
        ls[tid] = lo;
        le[tid] = hi;
    }
    __syncthreads();
}
// dirty[i] = the level of entry i (bounds ls / le) needs an order it does not have: several entries, not all of one epoch.
// Every thread calls; ends with a barrier.
__device__ __forceinline__ void pool_levels_dirty(const unsigned long long *key, uint32_t n, const uint32_t *ls, const uint32_t *le,
                                                  uint32_t *dirty, int ksh, PoolScratch &X) {
    const uint32_t tid = threadIdx.x;
    if (tid < n) X.dl[tid] = 0;
    __syncthreads();
    if (tid < n) {
        const uint32_t lo = ls[tid], hi = le[tid];
        const unsigned long long e0 = key[lo] >> ksh;
        if (hi - lo > 1 && (e0 == 0 || (key[tid] >> ksh) != e0)) X.dl[lo] = 1;  // (everybody who writes writes 1)
    }
    __syncthreads();
    if (tid < n) dirty[tid] = X.dl[ls[tid]];
    __syncthreads();
}
// The end of a selection, once the keys are what they are going to be (b_*: the pool sorted by count, s_ls / s_le its
// level bounds): the order inside every level, the batch -- the longest prefix with a != b, no pair chaining onto another, no first token twice, every level
// it enters in a known order --, the step's state, and the rest of the entries as the next step's pool.  Every thread of
// the deciding workgroup calls (blockDim.x >= PL_CAP).
__device__ __forceinline__ void pool_finish(DevState *st, PoolEnt *__restrict__ pool, const uint32_t *b_xy, const uint32_t *b_c,
                                            const unsigned long long *b_key, uint32_t *a_xy, uint32_t *a_c,
                                            unsigned long long *a_key, const uint32_t *s_ls, const uint32_t *s_le,
                                            uint32_t *s_dirty, uint32_t *s_clash, uint32_t *s_k, uint32_t *s_unt, uint32_t n,
                                            uint32_t kmax, uint32_t iter, uint32_t theta, unsigned long long epoch, bool rebuilt,
                                            uint32_t hint_below, int ksh, PoolScratch &X) {
    const uint32_t tid = threadIdx.x;
    const uint32_t nwalk = min(n, kmax);
    // ---- the order inside every level: by key (a level whose keys are not of one epoch stays without an order) ----------
    const uint32_t sh = blockDim.x >= 4 * PL_CAP ? 2u : 0u;  // (threads per entry: four in the 1024-thread selection)
    const uint32_t tpe = 1u << sh, e = tid >> sh, q = tid & (tpe - 1u);
    if (tid < n) {
        X.rank[tid] = 0;
        X.dl[tid] = 0;
    }
    if (tid == 0) *s_unt = 0;
    __syncthreads();
    if (e < n) {
        const uint32_t lo = s_ls[e], hi = s_le[e];
        const unsigned long long k = b_key[e];
        uint32_t r = 0;
#pragma unroll 4
        for (uint32_t j = lo + q; j < hi; j += tpe) {
            const unsigned long long kj = b_key[j];
            r += (kj < k) | ((kj == k) & (j < e));
        }
        if (r) atomicAdd(&X.rank[e], r);
        if (q == 0) {
            const unsigned long long e0 = b_key[lo] >> ksh;
            if (hi - lo > 1 && (e0 == 0 || (k >> ksh) != e0)) X.dl[lo] = 1;
        }
    }
    __syncthreads();
    if (tid < n) {
        const uint32_t lo = s_ls[tid];
        const uint32_t r = lo + X.rank[tid];
        a_xy[r] = b_xy[tid];
        a_c[r] = b_c[tid];
        a_key[r] = b_key[tid];
        s_dirty[r] = X.dl[lo];
    }
    __syncthreads();
    // ---- the batch: the longest prefix with a != b, no chain, no first token twice, every level entered in a known order ----
    if (tid < 64) {  // (nwalk <= CH_KSWEEP < 64: one wave)
        uint32_t bad = 0;
        if (tid < nwalk) {
            const uint32_t x = a_xy[tid] >> 16, y = a_xy[tid] & 0xFFFFu;
            bad = (x == y) | s_dirty[tid];
#pragma unroll
            for (uint32_t j = 0; j < (uint32_t)CH_KSWEEP - 1u; j++) {
                if (j < tid) {
                    const uint32_t xj = a_xy[j] >> 16, yj = a_xy[j] & 0xFFFFu;
                    // (a SECOND token may be shared: sites of (a, b) and (c, b) never overlap and neither merge moves the
                    // other's count -- only a pair that could chain onto a site of the batch, x a second token or y a first
                    // one, stops the walk; first tokens stay distinct: the merge pass looks a pair up by its first token)
                    bad |= (xj == x) | (xj == y) | (yj == x);
                }
            }
        }
        const unsigned long long bb = __ballot(bad != 0);
        if (tid == 0) *s_k = bb ? min(nwalk, (uint32_t)__ffsll((long long)bb) - 1u) : nwalk;
    }
    __syncthreads();
    const uint32_t K = *s_k;
    if (tid == 0) {
        st_agent(&st->adj, 0u);  // (written through: the merge pass adds to it with device atomics -- in k_step in this very launch)
        st->count = a_c[0];
        st->ntied = s_le[0];
        st->firstpos = NOPOS;
        st->sel_tie = 0;
        st->a = (int32_t)(a_xy[0] >> 16);
        st->b = (int32_t)(a_xy[0] & 0xFFFFu);
        st->fin_a = (int32_t)(a_xy[0] >> 16);
        st->fin_b = (int32_t)(a_xy[0] & 0xFFFFu);
        st->bk = K;
        st->bz0 = 256u + iter;
        st->tl_n = st->tl_skip = 0;
        st->dp_wait = 0;
        st->sel_mode = rebuilt ? CH_FULL : CH_LIST;  // (statistics: what kind of step this was)
        if (K == 0) {
            st->found = 0;
            // a == b at the head: the general path's merge | a level the step cannot order: the general path's selection
            st->defer = ((a_xy[0] >> 16) == (a_xy[0] & 0xFFFFu) && !s_dirty[0]) ? 1u : 2u;
            st->pool_n = 0;
            st->pool_hint_next = 1;
        } else {
            st->found = 1;
        }
    }
    if (K == 0) return;
    if (tid < K) X.rank[tid] = a_xy[tid] >> 16;  // (the batch's first tokens, for chain_hash_find below)
    __syncthreads();
    if (tid < 64) {  // the batch itself, a pair per lane
        uint32_t c = 0;
        if (tid < K) {
            c = a_c[tid];
            st->ba[tid] = (int32_t)(a_xy[tid] >> 16);
            st->bb[tid] = (int32_t)(a_xy[tid] & 0xFFFFu);
            st_agent(&st->badj[tid], 0u);
            st->bcnt[tid] = c;
        }
        const uint32_t cmax = wave_umax_dpp(c);
        // the multiplier of the merge pass's first-token look-up table, found here once (chain_hash_find, k_chain.hip)
        // instead of by every workgroup of the pass; the key says which batch it belongs to
        const uint32_t hm = chain_hash_find(reinterpret_cast<const uint32_t *>(X.rank), K);
        if (tid == 0) {
            st->brep = cmax > CH_REP_COUNT ? (uint32_t)CH_RSTRIDE : (uint32_t)CH_REP;
            st->bhm = hm;
            st->bhm_key = ((256u + iter) << 8) | K;
        }
    }
    // ---- the rest is the next step's pool ----------------------------------------------------------------------------------
    bool unt = false;
    if (tid >= K && tid < n) {
        PoolEnt e2;
        e2.xy = a_xy[tid];
        e2.c = a_c[tid];
        e2.key = a_key[tid];
        pool[tid - K] = e2;
        const uint32_t x = e2.xy >> 16, y = e2.xy & 0xFFFFu;
        bool touched = false;
#pragma unroll
        for (uint32_t p = 0; p < (uint32_t)CH_KSWEEP; p++)
            if (p < K) touched |= ((a_xy[p] & 0xFFFFu) == x) | ((a_xy[p] >> 16) == y);
        unt = !touched;
    }
    if (tid < PL_CAP) {
        const unsigned long long ub = __ballot(unt);
        if (lane_id() == 0 && ub) atomicAdd(s_unt, (uint32_t)__popcll(ub));
    }
    __syncthreads();
    if (tid == 0) {
        st->pool_n = n - K;
        st->pool_theta = theta;
        st->pool_epoch = epoch;
        st->pool_hint_next = *s_unt < hint_below ? 1u : 0u;
    }
}

// The LDS of a selection (one struct, so that k_step can overlay it with its merge pass's).
struct PoolLds {
    unsigned long long s_red[32];
    uint32_t s_words[DBITS_WORDS], s_pref[DBITS_WORDS + 1];
    uint32_t s_exrow[CH_EX_CAP], s_exm[CH_EX_CAP], s_exarg[CH_EX_CAP];
    uint32_t a_xy[PL_GATHER], a_c[PL_GATHER], b_xy[PL_GATHER], b_c[PL_GATHER];
    unsigned long long a_key[PL_GATHER], b_key[PL_GATHER];
    uint32_t s_ls[PL_CAP], s_le[PL_CAP], s_dirty[PL_CAP], s_clash[PL_CAP];
    int32_t s_tied[2 * TIE_CAP];
    unsigned long long s_pos[TIE_CAP];
    uint32_t s_lidx[TIE_CAP];
    uint32_t s_rows[PL_ROWS], s_cnt[8], s_r16[16], s_wtot[PL_CAP / 64];
    uint32_t s_fail, s_n, s_theta, s_nrows, s_nl, s_reach, s_k, s_unt, s_x;
    PoolScratch X;
    uint32_t s_sw[64];  // the words of st a selection needs (fetched at once)
};
// The selection: workgroup `blk` of `nblk` (0 decides, 1 .. nblk - 1 do a rebuild's row work); every thread calls.  Shared
// by k_pool_sel (its own launch) and k_step (k_step.hip: the head of the step's one launch).
__device__ __forceinline__ void
pool_sel_body(uint32_t *__restrict__ rowmax, uint32_t *__restrict__ mat, uint32_t stride, DevState *st, const SlotRefH &ref,
              const CandArgs &C, uint32_t *__restrict__ dbits, unsigned long long *__restrict__ res, uint32_t tag,
              unsigned long long *__restrict__ req, uint32_t kcap, PoolEnt *__restrict__ pool, uint32_t *__restrict__ gather,
              uint32_t hint_below, long long *__restrict__ dpkey, unsigned long long dprank, PoolEnt *__restrict__ mid,
              PoolLds &L, const uint32_t blk, const uint32_t nblk, unsigned long long *dbg = nullptr) {
    auto dstamp = [&](int i) {  // (debug, BPE_STEP_STAMPS: where a selection's time goes)
        if (dbg && threadIdx.x == 0 && blk == 0) dbg[i] = wall_clock64();
    };
    auto &s_red = L.s_red;
    auto &s_words = L.s_words;
    auto &s_pref = L.s_pref;
    auto &s_exrow = L.s_exrow;
    auto &s_exm = L.s_exm;
    auto &s_exarg = L.s_exarg;
    auto &a_xy = L.a_xy;
    auto &a_c = L.a_c;
    auto &b_xy = L.b_xy;
    auto &b_c = L.b_c;
    auto &a_key = L.a_key;
    auto &b_key = L.b_key;
    auto &s_ls = L.s_ls;
    auto &s_le = L.s_le;
    auto &s_dirty = L.s_dirty;
    auto &s_clash = L.s_clash;
    auto &s_tied = L.s_tied;
    auto &s_pos = L.s_pos;
    auto &s_lidx = L.s_lidx;
    auto &s_rows = L.s_rows;
    auto &s_cnt = L.s_cnt;
    auto &s_r16 = L.s_r16;
    auto &s_wtot = L.s_wtot;
    uint32_t &s_fail = L.s_fail, &s_n = L.s_n, &s_theta = L.s_theta, &s_nrows = L.s_nrows, &s_nl = L.s_nl, &s_k = L.s_k, &s_reach = L.s_reach,
             &s_unt = L.s_unt, &s_x = L.s_x;
    const uint32_t tid = threadIdx.x;
    // everything the selection needs from st, and its own pool entry, in ONE round trip (the fields used to be read where
    // the code came to them: three dependent round trips before the first table word was asked for)
    enum { PW_STATUS = 32, PW_DEFER, PW_GAP, PW_ITER, PW_NM, PW_HINT, PW_THETA, PW_POOLN, PW_BK, PW_BZ0, PW_EPOCH_LO, PW_EPOCH_HI, PW_N };
    PoolEnt my_ent;
    my_ent.xy = my_ent.c = 0;
    my_ent.key = 0;
    if (blk == 0 && tid < PL_CAP) my_ent = pool[tid];
    if (tid < (uint32_t)PW_N) {
        const uint32_t *base = reinterpret_cast<const uint32_t *>(st);
        uint32_t off;
        if (tid < 16) off = (uint32_t)offsetof(DevState, ba) / 4 + tid;
        else if (tid < 32) off = (uint32_t)offsetof(DevState, bb) / 4 + (tid - 16);
        else {
            constexpr uint32_t o[PW_N - 32] = {
                (uint32_t)offsetof(DevState, status) / 4,     (uint32_t)offsetof(DevState, defer) / 4,      (uint32_t)offsetof(DevState, gap) / 4,
                (uint32_t)offsetof(DevState, iter) / 4,       (uint32_t)offsetof(DevState, num_merges) / 4, (uint32_t)offsetof(DevState, pool_hint) / 4,
                (uint32_t)offsetof(DevState, pool_theta) / 4, (uint32_t)offsetof(DevState, pool_n) / 4,     (uint32_t)offsetof(DevState, bk) / 4,
                (uint32_t)offsetof(DevState, bz0) / 4,        (uint32_t)offsetof(DevState, pool_epoch) / 4, (uint32_t)offsetof(DevState, pool_epoch) / 4 + 1};
            off = o[0];
#pragma unroll
            for (int k = 1; k < PW_N - 32; k++) off = (tid == 32u + (uint32_t)k) ? o[k] : off;
        }
        L.s_sw[tid] = base[off];
    }
    __syncthreads();
    const uint32_t *s_sw = L.s_sw;
    const uint32_t status = s_sw[PW_STATUS], defer = s_sw[PW_DEFER], gap = s_sw[PW_GAP];
    const uint32_t iter = s_sw[PW_ITER], nm = s_sw[PW_NM], hint = s_sw[PW_HINT];
    // Sharded training (dpkey != nullptr): the pool is a replica of GLOBAL state, so every rank maintains, gathers and
    // sorts alike -- but a first occurrence is a rank-local fact.  This launch leaves the MIN all-reduce payload ([0] =
    // -status, [1] = -1 if this rank cannot order its share (short slots about), [2 + l] = rank << 33 | first local
    // position of the l-th entry to locate, INT64_MAX = no occurrence here) and the sorted pool in `mid`;
    // k_pool_sel_dp finishes the selection from the reduced words.  Every word is first written with its neutral value.
    if (dpkey && blk == 0 && tid < (uint32_t)DP_KEY_WORDS)
        dpkey[tid] = tid == 0 ? -(long long)status : (tid == 1 ? 0ll : 0x7FFFFFFFFFFFFFFFll);
    if (status || defer) return;
    if (iter >= nm) {  // training is over: this step and the ones behind it do nothing
        if (blk == 0 && tid == 0) st->bk = 0;
        return;
    }
    const uint32_t vcur = 256u + iter;
    const DirtyView D{s_words, s_pref};
    // ================= workgroups 1..: a rebuild's row work (only in a launch that was told to expect one) ===========
    if (blk != 0) {
        if (!hint) return;
        const uint32_t nd = dirty_view_build(dbits, D);
        if (nd) lean_scan_rows(mat, stride, rowmax, vcur, NOROW, NOROW, NOROW, D, 3 + nd, blk - 1, nblk - 1, res, tag,
                               true, s_red, 3);
        if (tid == 0) {
            uint32_t n = 0, th = 0;
            if (!granule_get(req, tag, n)) n = 0;  // (never asked: the deciding workgroup reports its own failures)
            if (n && !granule_get(req + 1 + PL_ROWS, tag, th)) n = 0;
            s_n = n;
            s_theta = th;
        }
        __syncthreads();
        const uint32_t n = s_n, theta = s_theta;
        if (blk - 1 >= n) return;
        for (uint32_t j = blk - 1; j < n; j += nblk - 1) {
            if (tid == 0) {
                uint32_t x = 0;
                s_x = granule_get(req + 1 + j, tag, x) ? x : 0xFFFFFFFFu;
            }
            __syncthreads();
            const uint32_t x = s_x;
            if (x != 0xFFFFFFFFu) {
                const uint32_t *row = mat + (size_t)x * stride;
                const uint32_t n4 = (vcur + 3) & ~3u;
                constexpr int U = 8;
                for (uint32_t base = 0; base < n4; base += U * 4096) {
                    uint4 q[U];
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const uint32_t y = base + ((uint32_t)u * 1024u + tid) * 4u;
                        q[u] = (y < n4) ? *reinterpret_cast<const uint4 *>(row + y) : make_uint4(0u, 0u, 0u, 0u);
                    }
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const uint32_t y = base + ((uint32_t)u * 1024u + tid) * 4u;
                        const uint32_t v[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            if (v[k] >= theta && y + k < vcur) {
                                const uint32_t s = atomicAdd(&gather[0], 1u);
                                if (s < PL_GATHER) {
                                    __hip_atomic_store(&gather[2 + 2 * s], (x << 16) | (y + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    __hip_atomic_store(&gather[3 + 2 * s], v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                }
                            }
                        }
                    }
                }
            }
            __syncthreads();  // (s_x is rewritten by the next round)
        }
        // (what was gathered went out as agent-scope stores: it needs no cache maintenance, only to have been
        // acknowledged before the deciding workgroup is told -- a __threadfence() here made every one of the 63 x 1024
        // threads write back and invalidate its L2, tens of microseconds per rebuild: tools/atomic_peak.hip)
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0) granule_put(req + 2 + PL_ROWS + blk, tag, 1u);
        return;
    }
    // ================= the deciding workgroup ========================================================================
    auto dismiss = [&]() {
        if (hint && tid == 0) granule_put(req, tag, 0u);
    };
    uint32_t rm[SEL_RPT];
    if (hint) select_load(rowmax, vcur, rm);  // (a rebuild is likely: the row maxima travel while the pool is looked at)
    uint32_t theta = s_sw[PW_THETA];
    unsigned long long epoch = (unsigned long long)s_sw[PW_EPOCH_LO] | ((unsigned long long)s_sw[PW_EPOCH_HI] << 32);
    const uint32_t n0 = min(s_sw[PW_POOLN], PL_CAP);
    if (tid == 0) {
        s_fail = 0;
        s_nrows = 0;
        s_nl = 0;
        s_unt = 0;
    }
    if (tid < 8) s_cnt[tid] = 0;
    dstamp(8);
    // ---- maintain: what the last batch made of my entry ------------------------------------------------------------
    uint32_t vx[4] = {0, 0, 0, 0}, vy[4] = {0, 0, 0, 0}, cv[4] = {0, 0, 0, 0}, keepm = 0, ec = 0;
    unsigned long long ekey = 0;
    // (an entry (x, y) whose x is the second token of SEVERAL batch pairs p has a variant (Z_p, y) -- and (Z_p, Zy) -- for
    // each of them: counted here, looked up again when they are written; rare, and the four-variant path stays as it was)
    uint32_t mxm = 0, nmulti = 0;
    bool multi = false;
    auto multi_variants = [&](auto &&emit) {
        const uint32_t zp = s_sw[PW_BZ0];
        const uint32_t x = vx[0], y = vy[0];
        const bool hy = vy[2] != y;
        for (uint32_t m = mxm | 0x80000000u; m; m &= m - 1u) {
            const uint32_t p = (uint32_t)__ffs((int)m) - 1u;
            const uint32_t lx = p == 31u ? x : zp + p;
            for (uint32_t r = 0; r < (hy ? 2u : 1u); r++) {
                const uint32_t ry = r ? vy[2] : y;
                const uint32_t c = mat[(size_t)lx * stride + ry];
                if (c >= theta) emit(lx, ry, c);
            }
        }
    };
    if (tid < n0) {
        const uint32_t Kp = min(s_sw[PW_BK], (uint32_t)CH_KMAX), zp = s_sw[PW_BZ0];
        const PoolEnt e = my_ent;
        const uint32_t x = e.xy >> 16, y = e.xy & 0xFFFFu;
        ec = e.c;
        ekey = e.key;
        int32_t zx = -1, zy = -1;
        for (uint32_t p = 0; p < Kp; p++) {
            if (s_sw[16 + p] == x) {
                zx = (int32_t)(zp + p);
                mxm |= 1u << p;
            }
            if (s_sw[p] == y) zy = (int32_t)(zp + p);
        }
        multi = (mxm & (mxm - 1u)) != 0;  // several pairs of the batch end in x (a batch may share second tokens)
        vx[0] = vx[2] = x;
        vx[1] = vx[3] = zx >= 0 ? (uint32_t)zx : x;
        vy[0] = vy[1] = y;
        vy[2] = vy[3] = zy >= 0 ? (uint32_t)zy : y;
        if (zx < 0 && zy < 0) {  // shares no token that matters: untouched
            cv[0] = e.c;
            keepm = 1u;
        } else if (multi) {
            multi_variants([&](uint32_t, uint32_t, uint32_t) { nmulti++; });
        } else {
            const bool on[4] = {true, zx >= 0, zy >= 0, zx >= 0 && zy >= 0};
#pragma unroll
            for (int v = 0; v < 4; v++) cv[v] = on[v] ? mat[(size_t)vx[v] * stride + vy[v]] : 0u;
#pragma unroll
            for (int v = 0; v < 4; v++) keepm |= (on[v] && cv[v] >= theta) ? 1u << v : 0u;
        }
    }
    __syncthreads();
    {
        const uint32_t no = multi ? nmulti : (uint32_t)__popc(keepm);
        const uint32_t inc = wave_iscan_add(no);
        if (tid < PL_CAP && lane_id() == 63) s_wtot[wave_id()] = inc;
        __syncthreads();
        if (tid < PL_CAP) {
            uint32_t o = inc - no;
            for (int w = 0; w < wave_id(); w++) o += s_wtot[w];
            if (multi) {
                multi_variants([&](uint32_t lx, uint32_t ry, uint32_t c) {
                    if (o < PL_GATHER) {
                        a_xy[o] = (lx << 16) | ry;
                        a_c[o] = c;
                        a_key[o] = c == ec ? ekey : 0ull;
                    }
                    o++;
                });
            }
#pragma unroll
            for (int v = 0; v < 4; v++) {
                if ((keepm >> v) & 1u) {
                    if (o < PL_GATHER) {  // (always, unless entries with several variants came before: n1 > PL_GATHER below)
                        a_xy[o] = (vx[v] << 16) | vy[v];
                        a_c[o] = cv[v];
                        a_key[o] = cv[v] == ec ? ekey : 0ull;  // took over every occurrence: stands where the entry stood
                    }
                    o++;
                }
            }
        }
        __syncthreads();
    }
    dstamp(9);
    uint32_t n1 = 0;
    if (n0) for (uint32_t w = 0; w < PL_CAP / 64; w++) n1 += s_wtot[w];
    if (n1 > PL_GATHER) n1 = 0;  // (more variants than the arrays hold -- shared second tokens only: the pool is gathered afresh)
    bool rebuilt = false;
    // (a hinted launch -- the scanning workgroups stayed -- whose pool could not fill a batch any more gathers a fresh,
    // deeper one instead of merging the last few entries in small batches: the old entries are in it, without their keys)
    if (hint && n1 < min(kcap, nm - iter)) n1 = 0;
    if (n1 == 0) {
        if (!hint) {  // nobody stayed to scan rows: this step merges nothing, the next one rebuilds
            if (tid == 0) {
                st->bk = 0;
                st->found = 0;
                st->pool_n = 0;
                st->pool_hint_next = 1;
                st->sel_mode = CH_LIST;
                st->tl_n = st->tl_skip = 0;
            }
            return;
        }
        // ---- rebuild ---------------------------------------------------------------------------------------------
        rebuilt = true;
        const uint32_t nd = dirty_view_build(dbits, D);
        if (tid == 0) st->sel_ran = 1;  // (this launch re-scans every flagged row)
        if (nd > CH_EX_CAP) {  // (the other workgroups re-scan them all the same; the general path selects)
            if (tid == 0) {
                st->found = 0;
                st->bk = 0;
                st->defer = 2;
            }
            dismiss();
            return;
        }
        for (uint32_t i = tid; i < nd; i += 1024) {
            const uint32_t x = dirty_view_row(D, i);
            uint32_t m = 0, arg = 0;
            const bool ok = granule_get(res + 2 * (size_t)i, tag, m) && granule_get(res + 2 * (size_t)i + 1, tag, arg);
            if (!ok) s_fail = 1;
            s_exrow[i] = x;
            s_exm[i] = m;
            s_exarg[i] = arg;
        }
        __syncthreads();
        if (s_fail) {  // a row never arrived: never decide on a stale maximum
            if (tid == 0) atomicExch(&st->status, ST_LOOKBACK);
            dismiss();
            return;
        }
        const uint2 *__restrict__ rowma = reinterpret_cast<const uint2 *>(rowmax);
        auto stale = [&](uint32_t x) -> bool { return (s_words[x >> 5] >> (x & 31)) & 1u; };
        uint32_t m = 0;
#pragma unroll
        for (int i = 0; i < SEL_RPT; i++) {
            const uint32_t x = tid + 1024u * (uint32_t)i;
            if (x >= vcur || stale(x)) rm[i] = 0u;
            m = max(m, rm[i]);
        }
        for (uint32_t x = tid + 1024u * SEL_RPT; x < vcur; x += 1024)
            if (!stale(x)) m = max(m, rowma[x].x);
        for (uint32_t i = tid; i < nd; i += 1024) m = max(m, s_exm[i]);
        m = wave_umax_dpp(m);
        if (lane_id() == 0) s_r16[wave_id()] = m;
        __syncthreads();
        uint32_t M = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) M = max(M, s_r16[w]);
        if (M == 0) {  // stats is empty: max() raises ValueError in the reference (F6)
            if (tid == 0) {
                st->status = ST_EMPTY;
                st->count = 0;
                st->found = 0;
                st->bk = 0;
                st->sel_tie = 0;
            }
            dismiss();
            return;
        }
        // eight candidate thresholds, deepest first: M - M/4, M - M/8, ..., M - M/256, M; rows that reach each
        uint32_t th[8], cn[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            th[k] = k < 7 ? M - (M >> (2 + k)) : M;
            cn[k] = 0;
        }
        auto tally = [&](uint32_t v) {
#pragma unroll
            for (int k = 0; k < 8; k++) cn[k] += v >= th[k];
        };
#pragma unroll
        for (int i = 0; i < SEL_RPT; i++) tally(rm[i]);
        for (uint32_t x = tid + 1024u * SEL_RPT; x < vcur; x += 1024)
            if (!stale(x)) tally(rowma[x].x);
        for (uint32_t i = tid; i < nd; i += 1024) tally(s_exm[i]);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t s = wave_sum_u32(cn[k]);
            if (lane_id() == 0 && s) atomicAdd(&s_cnt[k], s);
        }
        __syncthreads();
        int ks = -1;
#pragma unroll
        for (int k = 7; k >= 0; k--)
            if (s_cnt[k] <= PL_ROWS) ks = k;  // (the deepest threshold that few enough rows reach)
        if (ks < 0) {  // more rows at the maximum itself than a rebuild scans: the general path decides
            if (tid == 0) {
                st->count = M;
                st->found = 0;
                st->bk = 0;
                st->defer = 2;
            }
            dismiss();
            return;
        }
        theta = th[0];
#pragma unroll
        for (int k = 1; k < 8; k++) theta = (k == ks) ? th[k] : theta;
        if (ks == 0) theta = th[0];
        auto row_in = [&](uint32_t x, uint32_t v) {
            if (v >= theta) {
                const uint32_t s = atomicAdd(&s_nrows, 1u);
                if (s < PL_ROWS) s_rows[s] = x;
            }
        };
#pragma unroll
        for (int i = 0; i < SEL_RPT; i++) row_in(tid + 1024u * (uint32_t)i, rm[i]);
        for (uint32_t x = tid + 1024u * SEL_RPT; x < vcur; x += 1024)
            if (!stale(x)) row_in(x, rowma[x].x);
        for (uint32_t i = tid; i < nd; i += 1024) row_in(s_exrow[i], s_exm[i]);
        if (tid == 0) __hip_atomic_store(&gather[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);  // (the counter's reset is in memory before the rows are handed out; no fence: see above)
        __syncthreads();
        const uint32_t nrows = min(s_nrows, PL_ROWS);
        if (tid < nrows) granule_put(req + 1 + tid, tag, s_rows[tid]);
        if (tid == 0) {
            granule_put(req + 1 + PL_ROWS, tag, theta);
            granule_put(req, tag, nrows);
        }
        const uint32_t nh = min(nblk - 1, nrows);
        if (tid < nh) {
            uint32_t d = 0;
            if (!granule_get(req + 2 + PL_ROWS + (tid + 1), tag, d)) s_fail = 1;
        }
        __syncthreads();
        if (s_fail) {
            if (tid == 0) atomicExch(&st->status, ST_LOOKBACK);
            return;
        }
        const uint32_t ng = __hip_atomic_load(&gather[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ng > PL_GATHER || ng == 0) {  // (more pairs at theta or above than a rebuild looks at: the general path decides this merge)
            if (tid == 0) {
                st->count = M;
                st->found = 0;
                st->bk = 0;
                if (ng) st->defer = 2; else atomicExch(&st->status, ST_INTERNAL);
            }
            return;
        }
        if (tid < ng) {
            a_xy[tid] = __hip_atomic_load(&gather[2 + 2 * tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a_c[tid] = __hip_atomic_load(&gather[3 + 2 * tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a_key[tid] = 0ull;
        }
        n1 = ng;
        __syncthreads();
    } else {
        dismiss();  // (a hinted launch whose pool still holds entries: the row scans were not needed)
        if (hint && tid == 0) st->sel_ran = 1;  // (... but workgroups 1.. do re-scan every flagged row)
    }
    // ---- sort by count; more entries than the pool holds: whole levels leave from the bottom, theta rises above them ----
    dstamp(10);
    pool_sort(a_xy, a_c, a_key, b_xy, b_c, b_key, n1, L.X);
    dstamp(11);
    uint32_t n = n1;
    if (n1 > PL_CAP) {
        if (tid == 0) {
            uint32_t cut = PL_CAP;
            while (cut > 0 && b_c[cut - 1] == b_c[cut]) cut--;
            s_n = cut;
        }
        __syncthreads();
        n = s_n;
        if (n == 0) {  // one level larger than the pool: the general path decides
            if (tid == 0) {
                st->count = b_c[0];
                st->found = 0;
                st->bk = 0;
                st->defer = 2;
                st->pool_n = 0;
                st->pool_hint_next = 1;
            }
            return;
        }
        theta = b_c[n] + 1;
        __syncthreads();
    }
    // ---- the levels the walk can reach; the ones that need an order are located through the index ----------------------
    const uint32_t kmax = min(kcap, nm - iter);
    const uint32_t nwalk = min(n, kmax);  // the batch is a prefix of at most this many entries
    const int ksh = dpkey ? PL_KSH_DP : PL_KSH;
    PoolScratch &X = L.X;
    pool_level_bounds(b_c, n, s_ls, s_le, X);
    pool_levels_dirty(b_key, n, s_ls, s_le, s_dirty, ksh, X);
    // does entry i < nwalk end a batch that reaches its level (a == b, or a token shared with anything above its level's
    // end)?  Entry j looks at every such i (nwalk <= 15 broadcast reads) instead of entry i looking at every j.
    if (tid < 64) s_clash[tid] = 0;
    if (tid == 0) {
        X.fv = 0xFFFFFFFFu;
        X.nl = 0;
    }
    __syncthreads();
    if (tid < n) {
        const uint32_t x = b_xy[tid] >> 16, y = b_xy[tid] & 0xFFFFu;
#pragma unroll
        for (uint32_t i = 0; i < (uint32_t)CH_KSWEEP; i++) {
            if (i < nwalk) {
                const uint32_t xi = b_xy[i] >> 16, yi = b_xy[i] & 0xFFFFu;
                if (tid != i && tid < s_le[i] && ((xi == x) | (xi == y) | (yi == x))) s_clash[i] = 1;
            }
        }
        if (tid < nwalk && x == y) s_clash[tid] = 1;
    }
    __syncthreads();
    if (tid < 64) {
        const unsigned long long cb = __ballot(tid < nwalk && s_clash[tid] != 0);
        const uint32_t lim = cb ? (uint32_t)__ffsll((long long)cb) - 1u : nwalk - 1u;
        if (tid == 0) s_reach = s_le[lim];
    }
    __syncthreads();
    {
        const uint32_t reach = s_reach;
        // (sharded: what is located must not depend on this rank's own slots -- a rank with short slots about objects, a
        // rank whose shard is empty or has no index (C.T == 0) finds no occurrence: tie_by_index then looks at nothing --
        // but it names the same levels, so that the epoch and the keys k_pool_sel_dp makes are the same on every rank)
        const bool locate = dpkey || (C.T != 0 && gap == 0);
        // every level the walk can reach that lacks an order -- and, while the round of sixteen waves has room, the next
        // ones below (the same latency now, a clean level when the walk gets there).  Going down the levels that start
        // before scan_end: a level that lacks an order is taken whole, until one does not fit -- below `reach` the round
        // of sixteen, anywhere TIE_CAP -- and nothing after that one.  With P = the entries of such levels before a level,
        // the first level that does not fit is the first with P + size over its limit: a prefix sum, no walk.
        const uint32_t scan_end = min(n, reach + 48u);
        const bool cons = locate && tid < n && s_ls[tid] < scan_end && s_dirty[tid] != 0;
        const uint32_t one = cons ? 1u : 0u;
        const uint32_t inc = wave_iscan_add(one);
        if (tid < PL_CAP && lane_id() == 63) X.wtot[wave_id()] = inc;
        __syncthreads();
        uint32_t P = inc - one;
        if (tid < PL_CAP)
            for (int v = 0; v < wave_id(); v++) P += X.wtot[v];
        if (cons && s_ls[tid] == tid) {  // (a level's first entry: P = the entries taken before this level)
            const uint32_t size = s_le[tid] - tid;
            if ((tid >= reach && P + size > 16u) || P + size > (uint32_t)TIE_CAP) atomicMin(&X.fv, tid);
        }
        __syncthreads();
        const bool take = cons && s_ls[tid] < X.fv;
        if (take) {
            s_tied[2 * P] = (int32_t)(b_xy[tid] >> 16);
            s_tied[2 * P + 1] = (int32_t)(b_xy[tid] & 0xFFFFu);
            s_lidx[P] = tid;
        }
        if (tid < PL_CAP) {
            const unsigned long long tb = __ballot(take);
            if (lane_id() == 0 && tb) atomicAdd(&X.nl, (uint32_t)__popcll(tb));
        }
        __syncthreads();
        if (tid == 0) s_nl = X.nl;
        __syncthreads();
    }
    const uint32_t nl = s_nl;
    dstamp(12);
    if (dpkey) {
        // ---- sharded: my first occurrences into the payload, the sorted pool into `mid`; k_pool_sel_dp goes on ----------
        const bool objection = nl != 0 && gap != 0;
        if (nl && !objection) (void)tie_by_index(ref, C, s_tied, nl, s_pos);
        __syncthreads();
        if (tid == 1 && objection) dpkey[1] = -1ll;
        if (tid < nl && !objection && s_pos[tid] != NOPOS)
            dpkey[2 + tid] = (long long)((dprank << PL_POS_DP) | (s_pos[tid] & ((1ull << PL_POS_DP) - 1ull)));
        if (tid < n) {
            PoolEnt e;
            e.xy = b_xy[tid];
            e.c = b_c[tid];
            e.key = b_key[tid];
            mid[1 + tid] = e;
        }
        if (tid < nl) mid[1 + PL_CAP + tid].xy = s_lidx[tid];
        if (tid == 0) {
            PoolEnt h;  // header: n | nl << 16, theta, epoch | rebuilt << 63
            h.xy = n | (nl << 16);
            h.c = theta;
            h.key = epoch | ((unsigned long long)(rebuilt ? 1u : 0u) << 63);
            mid[0] = h;
            st->dp_wait = 1;
            st->bk = 0;
            st->found = 0;
        }
        return;
    }
    if (nl) {
        (void)tie_by_index(ref, C, s_tied, nl, s_pos);
        __syncthreads();
        epoch++;
        if (tid < nl) b_key[s_lidx[tid]] = s_pos[tid] != NOPOS ? (epoch << PL_KSH) | s_pos[tid] : 0ull;
        __syncthreads();
    }
    dstamp(13);
    pool_finish(st, pool, b_xy, b_c, b_key, a_xy, a_c, a_key, s_ls, s_le, s_dirty, s_clash, &s_k, &s_unt, n, kmax, iter, theta,
                epoch, rebuilt, hint_below, PL_KSH, X);
    dstamp(14);
}

__global__ void __launch_bounds__(1024)
k_pool_sel(uint32_t *__restrict__ rowmax, uint32_t *__restrict__ mat, uint32_t stride, DevState *st, SlotRefH ref,
           CandArgs C, uint32_t *__restrict__ dbits, unsigned long long *__restrict__ res, uint32_t tag,
           unsigned long long *__restrict__ req, uint32_t kcap, PoolEnt *__restrict__ pool, uint32_t *__restrict__ gather,
           uint32_t hint_below, long long *__restrict__ dpkey, unsigned long long dprank, PoolEnt *__restrict__ mid) {
    __shared__ PoolLds L;
    pool_sel_body(rowmax, mat, stride, st, ref, C, dbits, res, tag, req, kcap, pool, gather, hint_below, dpkey, dprank, mid, L,
                  blockIdx.x, gridDim.x);
}

// k_pool_sel_dp: the second half of a sharded selection, after the MIN all-reduce of the first occurrences (one workgroup
// of PL_CAP threads).  The reduced words order the located entries -- lowest (rank, local position) = earliest in the
// global stream (F3 / F5) -- on every rank alike.
__global__ void __launch_bounds__(PL_CAP)
k_pool_sel_dp(DevState *st, const long long *__restrict__ key, uint32_t kcap, PoolEnt *__restrict__ pool,
              const PoolEnt *__restrict__ mid, uint32_t hint_below) {
    __shared__ uint32_t a_xy[PL_CAP], a_c[PL_CAP], b_xy[PL_CAP], b_c[PL_CAP];
    __shared__ unsigned long long a_key[PL_CAP], b_key[PL_CAP];
    __shared__ uint32_t s_ls[PL_CAP], s_le[PL_CAP], s_dirty[PL_CAP], s_clash[PL_CAP];
    __shared__ uint32_t s_k, s_unt;
    __shared__ PoolScratch X;
    const uint32_t tid = threadIdx.x;
    const uint32_t status = st->status;
    if (key[0] < 0 && status == 0) {  // some rank failed: every rank stops at this merge
        if (tid == 0) st->status = ST_INTERNAL;
        return;
    }
    if (status || st->defer || !st->dp_wait) return;
    const PoolEnt h = mid[0];
    const uint32_t n = h.xy & 0xFFFFu, nl = h.xy >> 16, theta = h.c;
    unsigned long long epoch = h.key & 0x7FFFFFFFFFFFFFFFull;
    const bool rebuilt = (h.key >> 63) != 0;
    const uint32_t iter = st->iter, nm = st->num_merges;
    if (tid < n) {
        const PoolEnt e = mid[1 + tid];
        b_xy[tid] = e.xy;
        b_c[tid] = e.c;
        b_key[tid] = e.key;
    }
    __syncthreads();
    if (nl) {
        epoch++;
        const bool objection = key[1] < 0;  // (some rank has short slots about: nobody orders these levels)
        if (tid < nl) {
            const long long p = key[2 + tid];
            b_key[mid[1 + PL_CAP + tid].xy] =
                (!objection && p != 0x7FFFFFFFFFFFFFFFll) ? (epoch << PL_KSH_DP) | (unsigned long long)p : 0ull;
        }
        __syncthreads();
    }
    pool_level_bounds(b_c, n, s_ls, s_le, X);
    pool_finish(st, pool, b_xy, b_c, b_key, a_xy, a_c, a_key, s_ls, s_le, s_dirty, s_clash, &s_k, &s_unt, n, min(kcap, nm - iter),
                iter, theta, epoch, rebuilt, hint_below, PL_KSH_DP, X);
}

}  // namespace BPE_G
}  // namespace bpe
