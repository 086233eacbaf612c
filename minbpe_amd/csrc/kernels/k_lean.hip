// k_lean.hip -- the training iteration once a merged pair has few sites ("lean" iterations: 31,445
// of the 31,744 merges of a 1 GB / vocab-32000 run; for most of them a pass rewrites a few hundred
// slots and nothing is bound by bytes any more, only by the number of launches and of dependent
// memory round trips inside them -- a launch costs ~2 us plus ~1.7 us per round trip).
// Three launches instead of five, each a short chain:
//   k_sel_lean       the pair of this merge from a TWO-LEVEL view of the row maxima: the table update
//                    of the previous merge left one record per 64 rows (their maximum, who attains
//                    it) and the new maxima of rows a, b, Z as per-wave partial results, so the
//                    deciding workgroup reads ~30 KB instead of the 256 KB row-maxima array and waits
//                    for nobody -- unless that update flagged rows for re-scanning (one iteration in
//                    five): workgroups 1.. re-scan those and hand each result over as two tagged
//                    8-byte words (no fence, no flag: the data is the flag).  Ties through the index
//                    (tie_by_index, k_select.hip); nothing of the decision goes through memory until
//                    it is final.  When the tie is among few pairs it lines ALL of them up in the
//                    order of their first occurrences (the chain, DevState::chain): the reference
//                    merges them in that order for as long as their counts stand, so the launches
//                    of the following iterations only take the next pair off the chain.
//   k_rowsel_lean    the same without the records (the first lean iteration after one of the general
//                    path): workgroups 1.. re-scan rows a, b, Z and the flagged rows, workgroup 0
//                    reads the whole row-maxima array meanwhile and leaves those rows out.
//   k_merge_ab_lean  every wave finds its own candidate slots in the inverted index (the three
//                    filter words of the pair per 32 slots: no candidate list, no single block
//                    building one) and rewrites them (merge_ab_wave, k_slots2.hip; delta format B)
//   k_apply_lean     folds the delta into the pair table, one token per thread, every load in
//                    flight at once, no returning atomic: a row is flagged for re-scanning when the
//                    entry that lost pairs attained its maximum.  Thread t also holds entry t of rows
//                    a, b and Z as they will stand, so every wave leaves the partial maxima of those
//                    rows and the record of its 64 rows for k_sel_lean
// What workgroup 0 cannot settle alone (more than TIE_CAP tied pairs, short slots about, a tied pair
// the index does not lead to) and every pair with a == b is DEFERRED: the iteration reports
// ST_DEFER, everything enqueued behind it is a no-op that only carries the stream length forward,
// and the host re-runs the iteration through the general path.  75 of 31,744 merges have a == b;
// the general path pays a k_merge_aa launch in every iteration for them.
// Part of bpe_kernels.hip, which includes the parts in order.
// (no include guard: bpe_kernels.hip includes this part once per geometry, namespace BPE_G)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../bpe_device.h"

namespace bpe {
namespace BPE_G {

constexpr uint32_t NOROW = 0xFFFFFFFFu;
constexpr int DBITS_WORDS = 2048;     // one bit per row (vocab <= 65536)
constexpr uint32_t LEAN_EX_CAP = 1024;  // rows one selection launch (k_sel_lean, k_rowsel_lean) hands to its deciding workgroup
// Per-wave records of k_apply_lean (wave w of the token workgroups = tokens [64w, 64w + 64)), read by
// k_sel_lean: four arrays of LEAN_SUM_CAP uint4,
//   [0][w] = {maximum of the row maxima of rows 64w.. (rows a, b, Z and the rows flagged by this update
//             left out), the first row that attains it, that row's rowarg, how many rows attain it}
//   [1 + r][w], r = 0, 1, 2 for rows a, b, Z as they stand after the update: {maximum over columns
//             64w.., first column that attains it, last column that attains it, 0}
constexpr uint32_t LEAN_SUM_CAP = 1024;  // waves of the token workgroups at vocab 65536
constexpr uint32_t LEAN_CHAIN_TIES = 48;  // more tied pairs than this: no chain (every pair's first occurrence would be looked up)
constexpr uint32_t LEAN_BLK_CAP = 128;   // 64-row groups with several rows at the maximum that one selection looks into

// ---------------------------------------------------------------------------
// merge pass.  use_index == 0: no index (small streams) -- every live slot is visited.
// One 1024-thread workgroup per CU (sixteen waves = sixteen slots in flight).  A workgroup owns a
// contiguous range of the candidate mask -- the three filter words of the pair AND-ed per 32 slots,
// plus the slots an a == b pass rewrote since the index was built; thread k loads word k, all in one
// round trip -- compacts its candidates into a list in LDS and deals them to its waves one by one.
// (Measured: letting every wave work through the candidates of its own mask words costs 2x in the
// pass's duration -- one wave in a few hundred draws five or six slots and everybody waits for it.)
constexpr int LEAN_MT = 1024;
constexpr uint32_t LEAN_SUB = 128;  // mask words per round of a workgroup: at most 4096 candidates listed
// static LDS of k_merge_ab_lean: one slot of staging per wave + the candidate list.  More than the 64 KB a
// workgroup gets on other parts: the build is gfx950-only (160 KB per CU), and bpe_create turns the lean
// iterations off when the device reports less than this per workgroup.
constexpr int LEAN_LDS_BYTES = (LEAN_MT / 64) * TILE2 * 4 + (int)LEAN_SUB * 32 * 4 + 8;
static_assert(LEAN_LDS_BYTES <= 160 * 1024, "k_merge_ab_lean: staging + candidate list must fit the CU's LDS");
template <bool INDEXED>
__global__ void __launch_bounds__(LEAN_MT)
k_merge_ab_lean(AbArgs A, const uint32_t *__restrict__ idx_dirty, uint32_t use_index, uint32_t *__restrict__ dbits) {
    __shared__ __attribute__((aligned(16))) uint32_t s_out[LEAN_MT / 64][TILE2];
    __shared__ uint32_t s_list[LEAN_SUB * 32];
    __shared__ uint32_t s_tot[2];
    DevState *st = A.st;
    // the rows flagged by the table updates so far were re-scanned by the launch before this one -- unless
    // that was a chained iteration, whose selection launch only takes the next pair of the chain
    // (k_sel_lean): the flags then stay and pile up until a selection does re-scan them
    const uint32_t ran = st->sel_ran;
    if (blockIdx.x == 0 && ran) {
        for (uint32_t i = threadIdx.x; i < DBITS_WORDS; i += LEAN_MT) dbits[i] = 0;
        __syncthreads();  // (every thread has read the word above)
        if (threadIdx.x == 0) st->sel_ran = 0;
    }
    if (st->status || st->defer) return;
    if (!st->found) {
        if (blockIdx.x == 0 && threadIdx.x == 0) st->status = ST_INTERNAL;
        return;
    }
    const uint32_t a = (uint32_t)st->a, b = (uint32_t)st->b;
    if (a == b) {  // the general path's pass (k_merge_aa): the host re-runs this iteration there
        if (blockIdx.x == 0 && threadIdx.x == 0) st->defer = 1;
        return;
    }
    const uint32_t Tl = min(A.T, st->tlive);
    constexpr uint32_t NWV = LEAN_MT / 64;
    // short slots about: adjacency in slot numbers means nothing, visit everything (k_index.hip)
    if (!use_index || st->gap != 0) {
        const uint32_t nw = gridDim.x * NWV;
        for (uint32_t t = blockIdx.x * NWV + wave_id(); t < Tl; t += nw)
            merge_ab_wave<true, INDEXED, false>(s_out[wave_id()], nullptr, t, A, a, b, Tl);
        return;
    }
    const uint32_t nwords = (Tl + 31) / 32;
    const uint32_t per = (nwords + gridDim.x - 1) / gridDim.x;  // mask words of one workgroup
    const uint32_t wlo = blockIdx.x * per, whi = min(nwords, wlo + per);
    uint32_t h1, h2, h3;
    pair_hash(a, b, h1, h2, h3);
    for (uint32_t sub = wlo; sub < whi; sub += LEAN_SUB) {
        uint32_t mk = 0;
        const uint32_t w = sub + threadIdx.x;
        if (threadIdx.x < LEAN_SUB && w < whi) {
            mk = (A.idx[(size_t)h1 * A.istride + w] & A.idx[(size_t)h2 * A.istride + w] &
                  A.idx[(size_t)h3 * A.istride + w]) | idx_dirty[w];
            const uint32_t left = Tl - w * 32;
            if (left < 32) mk &= (1u << left) - 1u;
        }
        // (the words live in waves 0 and 1: two wave scans, no more)
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
        for (uint32_t i = wave_id(); i < n; i += NWV)
            merge_ab_wave<true, INDEXED, false>(s_out[wave_id()], nullptr, s_list[i], A, a, b, Tl);
        __syncthreads();  // (the list is rewritten by the next round)
    }
}

// ---------------------------------------------------------------------------
// table update, a != b (delta format B: vector 0 = SL, vector 1 = SR, st->adj; see k_slots2.hip).
// Workgroups [0, na): one token per thread.  Workgroups [na, grid): commit the staged headers; the
// first of them also makes the new stream length and the iteration's record.
__global__ void __launch_bounds__(256)
k_apply_lean(uint32_t *__restrict__ mat, uint32_t stride, uint32_t *__restrict__ delta, uint32_t vcap,
             const uint32_t *__restrict__ rowmax, DevState *st, uint32_t Z, uint32_t *__restrict__ dbits, int par,
             IterRec *rec, int iter, uint32_t na, SlotHdr *__restrict__ hdr_cur, const StageRec *__restrict__ stage,
             uint32_t *__restrict__ removed, uint32_t *__restrict__ smask, uint32_t nwords, uint4 *__restrict__ sums) {
    const uint32_t status = st->status, defer = st->defer;
    if (blockIdx.x < na) {
        const uint32_t t = blockIdx.x * 256u + threadIdx.x;
        const bool live = t <= Z;  // (dead lanes stay for the wave reductions below)
        const uint32_t nrep = 1u << (vcap >> 24);
        const uint32_t vc = vcap & 0xFFFFFFu;
        // the first sixteen replicas of this token's delta: on their way before anything is known
        // about the iteration (the state words above are scalar loads; these do not wait for them)
        uint32_t x[16][2];
        auto load_batch = [&](uint32_t r0) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t r = r0 + k;
                x[k][0] = (live && r < nrep) ? delta[delta_rep_off(r, vc) + t] : 0u;
                x[k][1] = (live && r < nrep) ? delta[delta_rep_off(r, vc) + vc + t] : 0u;
            }
        };
        load_batch(0);
        if (status || defer) return;
        const uint32_t a = (uint32_t)st->fin_a, b = (uint32_t)st->fin_b, adj = st->adj;
        // (t,a) as it stands: only this thread touches it in this launch ((b,a) also takes thread a's
        // update, and row b is re-scanned when that happens).  Loaded for every token, with everything
        // else -- a column walk, one 64-byte sector per token, instead of a dependent round trip
        // later -- and so are entry t of row a and of row b (two coalesced row reads)
        uint2 rm = make_uint2(0u, 0u);
        uint32_t old_ta = 0, old_at = 0, old_bt = 0, prevflag = 0;
        const uint32_t tied_count = st->count;
        if (live) {
            prevflag = (dbits[t >> 5] >> (t & 31)) & 1u;  // (flagged by an earlier update of a chain, not re-scanned yet)
            rm = reinterpret_cast<const uint2 *>(rowmax)[t];
            old_ta = mat[(size_t)t * stride + a];
            old_at = mat[(size_t)a * stride + t];
            old_bt = mat[(size_t)b * stride + t];
        }
        uint32_t sl = 0, sr = 0;
        for (uint32_t r0 = 0;;) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if (x[k][0]) delta[delta_rep_off(r0 + k, vc) + t] = 0;
                if (x[k][1]) delta[delta_rep_off(r0 + k, vc) + vc + t] = 0;
                sl += x[k][0];
                sr += x[k][1];
            }
            r0 += 16;
            if (r0 >= nrep) break;
            load_batch(r0);
        }
        // format B -> the four table updates of token t: column a and the new column Z of row t,
        // entries t of row b and of the new row Z
        const uint32_t dr = sr + (t == a ? adj : 0u), ir = sr + (t == Z ? adj : 0u);
        const bool abz = (t == a) | (t == b) | (t == Z);
        bool flagged = false;
        if (sl) {
            atomicSub(&mat[(size_t)t * stride + a], sl);
            atomicAdd(&mat[(size_t)t * stride + Z], sl);
            // Row t lost pairs in column a only and gained (t,Z) = sl <= what (t,a) lost: its maximum
            // moves only if (t,a) attained it.  Rows a and b: sl != 0 means (a,Z) resp. (b,a) and (b,Z)
            // change by an amount only this thread knows, while threads Z and a account for those
            // entries below -- re-scan the row instead (rare: a site preceded by a resp. by a lone b)
            flagged = abz ? true : (old_ta == rm.x);
            if (flagged) atomicOr(&dbits[t >> 5], 1u << (t & 31));
        }
        if (dr) atomicSub(&mat[(size_t)b * stride + t], dr);
        if (ir) atomicAdd(&mat[(size_t)Z * stride + t], ir);
        if (live && t == b) mat[(size_t)a * stride + b] = 0;  // no (a,b) survives the merge (F2)
        // ---- what the next selection needs, per wave -------------------------------------------
        // entry t of rows a, b, Z after this update (exact unless the row was flagged just above)
        const uint32_t va = (!live || t == b || t == Z) ? 0u : (t == a ? old_at - sl : old_at);
        const uint32_t vb = live ? old_bt - dr : 0u;
        const uint32_t vz = live ? ir : 0u;
        const uint32_t vs = (!live || abz || flagged || prevflag) ? 0u : rm.x;
        const uint32_t gw = blockIdx.x * 4u + wave_id(), base = gw * 64u;
        const int lane = lane_id();
        // a pair this merge created -- (t,Z) = sl, (Z,t) = ir -- that reaches the count the chain's pairs
        // are tied at: the next pair of the chain is no longer known to be the reference's next merge
        if (wave_umax_dpp(max(sl, ir)) >= tied_count && lane == 0) st->chain_cut = 1;
        {
            const uint32_t m = wave_umax_dpp(vs);
            const unsigned long long bal = __ballot(m != 0 && vs == m);
            const int fl = bal ? __ffsll((long long)bal) - 1 : 0;
            const uint32_t arg = (uint32_t)__shfl((int)rm.y, fl);
            if (lane == 0) sums[gw] = make_uint4(m, base + (uint32_t)fl, arg, (uint32_t)__popcll(bal));
        }
        const uint32_t v3[3] = {va, vb, vz};
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const uint32_t m = wave_umax_dpp(v3[r]);
            const unsigned long long bal = __ballot(m != 0 && v3[r] == m);
            const int fl = bal ? __ffsll((long long)bal) - 1 : 0, ll = bal ? 63 - __clzll((long long)bal) : 0;
            if (lane == 0) sums[(size_t)(1 + r) * LEAN_SUM_CAP + gw] = make_uint4(m, base + (uint32_t)fl, base + (uint32_t)ll, 0u);
        }
        return;
    }
    if (blockIdx.x == na && threadIdx.x < 64) {
        // ids removed by the merge pass: 256 counters, one per 256-byte line (see DELTA_SKEW)
        uint32_t v = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t x = removed[(threadIdx.x * 4 + i) * REMOVED_STRIDE];
            if (x) removed[(threadIdx.x * 4 + i) * REMOVED_STRIDE] = 0;
            v += x;
        }
        v = wave_sum_u32(v);
        if (threadIdx.x == 0) {
            const unsigned long long n = st->n[par];
            unsigned long long nn = n;
            if (status == 0) {
                nn = n - v;  // (a deferred iteration removed nothing: the length is carried forward)
                st->n[par ^ 1] = nn;
                if (!defer) {
                    st->scan_a = (uint32_t)st->fin_a;
                    st->scan_b = (uint32_t)st->fin_b;
                    st->scan_z = Z;
                }
            }
            st->removed = 0;
            rec[iter].a = status == 0 ? st->fin_a : st->a;
            rec[iter].b = status == 0 ? st->fin_b : st->b;
            rec[iter].count = st->count;
            rec[iter].status = (status == 0 && defer) ? ST_DEFER : status;
            rec[iter].new_len = nn;
            __threadfence_system();
            rec[iter].seq = (unsigned long long)iter + 1;
        }
    }
    if (status || defer) return;
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

// ---------------------------------------------------------------------------
// One row of the table by a 1024-thread workgroup, every load of the row in flight at once
// (vocab 32000: 8 x 16 bytes per thread).  zero_col >= 0: that entry is retired on the way.
// Returns the result to thread 0.
__device__ __forceinline__ void row_scan_wide(uint32_t *__restrict__ row, uint32_t ncols, int zero_col,
                                              unsigned long long *s_red, uint32_t &m_out, uint32_t &arg_out) {
    unsigned long long kf = 0, kl = 0;  // count << 32 | ~column  and  count << 32 | column
    const uint32_t n4 = (ncols + 3) & ~3u;
    constexpr int U = 8;
    for (uint32_t base = 0; base < n4; base += U * 4096) {
        uint4 q[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t y = base + ((uint32_t)u * 1024u + threadIdx.x) * 4u;
            q[u] = (y < n4) ? *reinterpret_cast<const uint4 *>(row + y) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t y = base + ((uint32_t)u * 1024u + threadIdx.x) * 4u;
            uint32_t v[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
            if (zero_col >= 0 && (uint32_t)zero_col - y < 4u && y < n4) {
                const uint32_t k = (uint32_t)zero_col - y;
                if (k == 0) v[0] = 0; else if (k == 1) v[1] = 0; else if (k == 2) v[2] = 0; else v[3] = 0;
                row[zero_col] = 0;
            }
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
        for (int w = 1; w < 16; w++) {
            kf = s_red[2 * w] > kf ? s_red[2 * w] : kf;
            kl = s_red[2 * w + 1] > kl ? s_red[2 * w + 1] : kl;
        }
        m_out = (uint32_t)(kf >> 32);
        const uint32_t cf = 0xFFFFFFFFu - (uint32_t)kf, cl = (uint32_t)kl;
        arg_out = (m_out == 0) ? 0u : (cf == cl ? cf : ROWARG_MULTI);
    }
}

// The rows to re-scan after a lean table update, as an indexable set: items 0, 1, 2 = rows scan_a,
// scan_b, scan_z (NOROW before the first lean update), item 3 + j = the j-th flagged row in ascending
// order.  Every workgroup builds the same view: the flag words and their exclusive popcount prefix
// in LDS (1024 threads).  Returns the number of flagged rows.
struct DirtyView {
    uint32_t *words;  // [DBITS_WORDS]
    uint32_t *pref;   // [DBITS_WORDS + 1]
};
__device__ __forceinline__ uint32_t dirty_view_build(const uint32_t *__restrict__ dbits, const DirtyView &D) {
    __shared__ uint32_t s_wtot[16];
    const uint32_t w0 = dbits[2 * threadIdx.x], w1 = dbits[2 * threadIdx.x + 1];
    D.words[2 * threadIdx.x] = w0;
    D.words[2 * threadIdx.x + 1] = w1;
    const uint32_t c0 = (uint32_t)__popc(w0), c = c0 + (uint32_t)__popc(w1);
    const uint32_t inc = wave_iscan_add(c);
    if (lane_id() == 63) s_wtot[wave_id()] = inc;
    __syncthreads();
    uint32_t off = inc - c, tot = 0;
    for (int v = 0; v < 16; v++) {
        const uint32_t x = s_wtot[v];
        if (v < wave_id()) off += x;
        tot += x;
    }
    D.pref[2 * threadIdx.x] = off;
    D.pref[2 * threadIdx.x + 1] = off + c0;
    if (threadIdx.x == 0) D.pref[DBITS_WORDS] = tot;
    __syncthreads();
    return tot;
}
__device__ __forceinline__ uint32_t dirty_view_row(const DirtyView &D, uint32_t j) {  // j < number of flagged rows
    uint32_t lo = 0, hi = DBITS_WORDS;  // the word w with pref[w] <= j < pref[w + 1]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (D.pref[mid] <= j) lo = mid; else hi = mid;
    }
    uint32_t m = D.words[lo];
    for (uint32_t k = j - D.pref[lo]; k > 0; k--) m &= m - 1u;
    return lo * 32 + (uint32_t)__ffs((int)m) - 1u;
}

// a row maximum on its way to the deciding workgroup: two 8-byte words {tag, value}, written and
// polled with agent-scope relaxed accesses (the tag never repeats: a launch counter)
typedef unsigned long long __attribute__((address_space(1))) gu64;
__device__ __forceinline__ void granule_put(unsigned long long *p, uint32_t tag, uint32_t v) {
    __hip_atomic_store((gu64 *)p, ((unsigned long long)tag << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool granule_get(unsigned long long *p, uint32_t tag, uint32_t &v) {
    for (uint32_t spins = 0; spins < LOOKBACK_SPINS; spins++) {
        const unsigned long long g = __hip_atomic_load((gu64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(g >> 32) == tag) {
            v = (uint32_t)g;
            return true;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    return false;
}

// Re-scan the rows in the set (items i = lo + first, lo + first + step, ...; lo = 3 leaves rows a, b, Z
// out); publish == true hands every result to the deciding workgroup as well, as item i - lo.
__device__ __forceinline__ void lean_scan_rows(uint32_t *__restrict__ mat, uint32_t stride, uint32_t *__restrict__ rowmax,
                                               uint32_t ncols, const uint32_t sa, const uint32_t sb, const uint32_t sz,
                                               const DirtyView &D, uint32_t n_items, uint32_t first, uint32_t step,
                                               unsigned long long *res, uint32_t tag, bool publish,
                                               unsigned long long *s_red, uint32_t lo = 0) {
    for (uint32_t i = lo + first; i < n_items; i += step) {
        const uint32_t x = i == 0 ? sa : (i == 1 ? sb : (i == 2 ? sz : dirty_view_row(D, i - 3)));
        uint32_t m = 0, arg = 0;
        if (x != NOROW) row_scan_wide(mat + (size_t)x * stride, ncols, x == sa ? (int)sb : -1, s_red, m, arg);
        if (threadIdx.x == 0) {
            // (written through: in k_step the table update of the same launch reads it from another workgroup)
            if (x != NOROW) st_agent64(reinterpret_cast<unsigned long long *>(rowmax) + x, (unsigned long long)m | ((unsigned long long)arg << 32));
            if (publish) {
                granule_put(res + 2 * (size_t)(i - lo), tag, m);
                granule_put(res + 2 * (size_t)(i - lo) + 1, tag, arg);
            }
        }
    }
}

// Row maxima alone (no selection): before a general k_select that follows a lean table update, and
// when training ends.
__global__ void __launch_bounds__(1024)
k_rowmax_lean(uint32_t *__restrict__ mat, uint32_t stride, uint32_t *__restrict__ rowmax, DevState *__restrict__ st,
              uint32_t ncols, const uint32_t *__restrict__ dbits) {
    __shared__ unsigned long long s_red[32];
    __shared__ uint32_t s_words[DBITS_WORDS], s_pref[DBITS_WORDS + 1];
    const uint32_t status = st->status, defer = st->defer;
    const uint32_t sa = st->scan_a, sb = st->scan_b, sz = st->scan_z;
    const DirtyView D{s_words, s_pref};
    const uint32_t nd = dirty_view_build(dbits, D);
    if (status || defer) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->sel_ran = 1;  // (every flagged row is re-scanned here)
        st->chain_n = 0;
    }
    lean_scan_rows(mat, stride, rowmax, ncols, sa, sb, sz, D, 3 + nd, blockIdx.x, gridDim.x, nullptr, 0u, false, s_red);
}

// K2 of a lean iteration with the index live, fused with the row maxima the previous table update
// left to do.  ncols = vcur = the ids in use (the previous merge's new token included).
__global__ void __launch_bounds__(1024)
k_rowsel_lean(uint32_t *__restrict__ rowmax, uint32_t *__restrict__ mat, uint32_t stride, uint32_t vcur, DevState *st,
              SlotRefH ref, CandArgs C, const uint32_t *__restrict__ dbits, unsigned long long *__restrict__ res,
              uint32_t tag) {
    __shared__ unsigned long long s_red[32];
    __shared__ uint32_t s_words[DBITS_WORDS], s_pref[DBITS_WORDS + 1];
    __shared__ int32_t s_tied[2 * TIE_CAP];
    __shared__ uint32_t s_bits[2048];
    __shared__ uint32_t s_exrow[LEAN_EX_CAP], s_exm[LEAN_EX_CAP], s_exarg[LEAN_EX_CAP];
    __shared__ uint32_t s_fail;
    const uint32_t status = st->status, defer = st->defer, gap = st->gap;
    const uint32_t sa = st->scan_a, sb = st->scan_b, sz = st->scan_z;
    const DirtyView D{s_words, s_pref};
    if (blockIdx.x != 0) {
        const uint32_t nd = dirty_view_build(dbits, D);
        if (status || defer) return;
        lean_scan_rows(mat, stride, rowmax, vcur, sa, sb, sz, D, 3 + nd, blockIdx.x - 1, gridDim.x - 1, res, tag, true,
                       s_red);
        return;
    }
    // ---- the deciding workgroup ---------------------------------------------------------------
    uint32_t rm[SEL_RPT];
    select_load(rowmax, vcur, rm);  // (in flight while the set of rows to leave out is put together)
    for (uint32_t i = threadIdx.x; i < 2048; i += 1024) s_bits[i] = 0;
    if (threadIdx.x == 0) s_fail = 0;
    const uint32_t nd = dirty_view_build(dbits, D);
    if (status || defer) return;
    if (threadIdx.x == 0) {
        st->sel_ran = 1;  // (this launch re-scans every flagged row)
        st->chain_n = 0;
    }
    const uint32_t n_items = 3 + nd;
    if (n_items > LEAN_EX_CAP) {  // (the other workgroups re-scan them all the same; the general path selects)
        if (threadIdx.x == 0) {
            st->found = 0;
            st->defer = 2;
        }
        return;
    }
    if (threadIdx.x < n_items) {
        const uint32_t i = threadIdx.x;
        const uint32_t x = i == 0 ? sa : (i == 1 ? sb : (i == 2 ? sz : dirty_view_row(D, i - 3)));
        uint32_t m = 0, arg = 0;
        const bool ok = granule_get(res + 2 * (size_t)i, tag, m) && granule_get(res + 2 * (size_t)i + 1, tag, arg);
        if (!ok) s_fail = 1;
        s_exrow[i] = x == NOROW ? 0u : x;
        s_exm[i] = x == NOROW ? 0u : m;
        s_exarg[i] = arg;
    }
    __syncthreads();
    // s_words becomes the bitmap of rows whose entry in the row-maxima array is stale (only now: the
    // searches above count the bits of the flagged rows inside a word)
    if (threadIdx.x < 3) {
        const uint32_t x = threadIdx.x == 0 ? sa : (threadIdx.x == 1 ? sb : sz);
        if (x != NOROW) atomicOr(&s_words[x >> 5], 1u << (x & 31));
    }
    __syncthreads();
    if (s_fail) {  // a row never arrived: never decide on a stale maximum
        if (threadIdx.x == 0) atomicExch(&st->status, ST_LOOKBACK);
        return;
    }
    uint32_t M, nt;
    select_core(rowmax, mat, stride, vcur, s_tied, s_bits, M, nt, rm, SelExtra{s_words, n_items, s_exrow, s_exm, s_exarg});
    if (M == 0) {
        if (threadIdx.x == 0) {
            st->status = ST_EMPTY;
            st->count = 0;
            st->found = 0;
            st->sel_tie = 0;
        }
        return;
    }
    uint32_t pi = 0;
    unsigned long long pos = NOPOS;
    bool decided = (nt == 1);
    if (!decided && nt <= TIE_CAP && gap == 0) {
        const unsigned long long key = tie_by_index(ref, C, s_tied, nt);
        if (key != NOPOS) {
            pi = (uint32_t)(key & 127u);
            pos = key >> 7;
            decided = true;
        }
    }
    if (threadIdx.x == 0) {
        st->adj = 0;  // (the previous pass's format-B "adjacent sites" count was folded into the table)
        st->count = M;
        st->ntied = nt;
        st->firstpos = pos;
        st->sel_tie = 0;
        if (decided) {
            st->a = s_tied[2 * pi];
            st->b = s_tied[2 * pi + 1];
            st->fin_a = s_tied[2 * pi];
            st->fin_b = s_tied[2 * pi + 1];
            st->found = 1;
        } else {
            st->found = 0;
            st->defer = 2;
        }
    }
}

// K2 of a lean iteration that follows a lean table update (index live): the deciding workgroup works
// from k_apply_lean's per-wave records (LEAN_SUM_CAP above) -- nwv of them, one per thread -- instead
// of the row-maxima array.  Rows a, b, Z of the previous merge come from their partial maxima and
// their entries in the row-maxima array are written here; the rows that update flagged are re-scanned
// by workgroups 1.. (items 0.. of the hand-off) and override a partial result for the same row.
__global__ void __launch_bounds__(1024)
k_sel_lean(uint32_t *__restrict__ rowmax, uint32_t *__restrict__ mat, uint32_t stride, uint32_t vcur, DevState *st,
           SlotRefH ref, CandArgs C, uint32_t *__restrict__ dbits, unsigned long long *__restrict__ res, uint32_t tag,
           const uint4 *__restrict__ sums, uint32_t nwv, uint32_t chain_on) {
    __shared__ unsigned long long s_red[32];
    __shared__ uint32_t s_words[DBITS_WORDS], s_pref[DBITS_WORDS + 1];
    __shared__ int32_t s_tied[2 * TIE_CAP];
    __shared__ uint32_t s_exrow[LEAN_EX_CAP + 3], s_exm[LEAN_EX_CAP + 3], s_exarg[LEAN_EX_CAP + 3];
    __shared__ uint32_t s_pr[16][3][3];
    __shared__ uint32_t s_blk[LEAN_BLK_CAP], s_rows[ARGMAX_ROWS];
    __shared__ uint32_t s_wm[16];
    __shared__ uint32_t s_fail, s_M, s_nt, s_nrows, s_nblk;
    __shared__ unsigned long long s_pos[TIE_CAP];
    __shared__ uint32_t s_order[TIE_CAP], s_cut;
    const uint32_t status = st->status, defer = st->defer, gap = st->gap;
    const uint32_t sa = st->scan_a, sb = st->scan_b, sz = st->scan_z;
    // ---- a chained iteration: the pair was lined up by an earlier selection ---------------------------
    // (DevState::chain; valid as long as no merge of the chain created a pair at the tied count.)  Nothing
    // is re-scanned and nothing selected: rows a, b, Z of the merge before this one join the flagged rows
    // (their partial maxima are about to be overwritten by the next table update).
    const uint32_t ch_pos = st->chain_pos;
    if (!status && !defer && chain_on && ch_pos < st->chain_n && !st->chain_cut) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (sa != NOROW) atomicOr(&dbits[sa >> 5], 1u << (sa & 31));
            if (sb != NOROW) atomicOr(&dbits[sb >> 5], 1u << (sb & 31));
            if (sz != NOROW) atomicOr(&dbits[sz >> 5], 1u << (sz & 31));
            const int32_t a = st->chain[2 * ch_pos], b = st->chain[2 * ch_pos + 1];
            st->a = a;
            st->b = b;
            st->fin_a = a;
            st->fin_b = b;
            st->found = 1;
            st->adj = 0;
            st->sel_tie = 0;
            st->firstpos = NOPOS;
            st->chain_pos = ch_pos + 1;
            st->chain_taken++;
        }
        return;
    }
    const DirtyView D{s_words, s_pref};
    if (blockIdx.x != 0) {
        const uint32_t nd = dirty_view_build(dbits, D);
        if (status || defer || nd == 0) return;
        lean_scan_rows(mat, stride, rowmax, vcur, sa, sb, sz, D, 3 + nd, blockIdx.x - 1, gridDim.x - 1, res, tag, true,
                       s_red, 3);
        return;
    }
    // ---- the deciding workgroup ---------------------------------------------------------------
    const uint32_t tid = threadIdx.x;
    const int lane = lane_id(), wv = wave_id();
    uint4 su = make_uint4(0u, 0u, 0u, 0u), pr[3];
#pragma unroll
    for (int r = 0; r < 3; r++) pr[r] = make_uint4(0u, 0u, 0u, 0u);
    if (tid < nwv) {  // (all in flight with the flag words below)
        su = sums[tid];
#pragma unroll
        for (int r = 0; r < 3; r++) pr[r] = sums[(size_t)(1 + r) * LEAN_SUM_CAP + tid];
    }
    if (tid == 0) {
        s_fail = 0;
        s_nt = 0;
        s_nrows = 0;
        s_nblk = 0;
    }
    const uint32_t nd = dirty_view_build(dbits, D);
    if (status || defer) return;
    if (tid == 0) {
        st->sel_ran = 1;  // (this launch re-scans every flagged row)
        st->chain_n = 0;
        st->chain_cut = 0;
    }
    auto flagged = [&](uint32_t x) -> bool { return (s_words[x >> 5] >> (x & 31)) & 1u; };
    // rows a, b, Z: maximum, first and last column that attains it, over the waves' partial results
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const uint32_t m = wave_umax_dpp(pr[r].x);
        const bool at = pr[r].x == m && m != 0;
        const uint32_t f = wave_umax_dpp(at ? 0xFFFFFFFFu - pr[r].y : 0u), l = wave_umax_dpp(at ? pr[r].z : 0u);
        if (lane == 0) {
            s_pr[wv][r][0] = m;
            s_pr[wv][r][1] = f;
            s_pr[wv][r][2] = l;
        }
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const uint32_t pm = lane < 16 ? s_pr[lane][r][0] : 0u;
            const uint32_t m = wave_umax_dpp(pm);
            const bool at = lane < 16 && pm == m && m != 0;
            const uint32_t f = wave_umax_dpp(at ? s_pr[lane][r][1] : 0u), l = wave_umax_dpp(at ? s_pr[lane][r][2] : 0u);
            if (lane == 0) {
                const uint32_t x = r == 0 ? sa : (r == 1 ? sb : sz);
                const uint32_t cf = 0xFFFFFFFFu - f;
                const uint32_t arg = (m == 0) ? 0u : (cf == l ? cf : ROWARG_MULTI);
                // (a flagged row is re-scanned by another workgroup, which also writes its entry)
                const bool use = x != NOROW && !flagged(x);
                if (use) reinterpret_cast<uint2 *>(rowmax)[x] = make_uint2(m, arg);
                s_exrow[r] = use ? x : 0u;
                s_exm[r] = use ? m : 0u;
                s_exarg[r] = arg;
            }
        }
    }
    const uint32_t n_items = 3 + nd;
    if (n_items > LEAN_EX_CAP) {  // (the other workgroups re-scan them all the same; the general path selects)
        if (tid == 0) {
            st->found = 0;
            st->defer = 2;
        }
        return;
    }
    if (tid < nd) {
        const uint32_t x = dirty_view_row(D, tid);
        uint32_t m = 0, arg = 0;
        const bool ok = granule_get(res + 2 * (size_t)tid, tag, m) && granule_get(res + 2 * (size_t)tid + 1, tag, arg);
        if (!ok) s_fail = 1;
        s_exrow[3 + tid] = x;
        s_exm[3 + tid] = m;
        s_exarg[3 + tid] = arg;
    }
    __syncthreads();
    if (s_fail) {  // a row never arrived: never decide on a stale maximum
        if (tid == 0) atomicExch(&st->status, ST_LOOKBACK);
        return;
    }
    // ---- the maximum ------------------------------------------------------------------------------
    {
        uint32_t m = su.x;  // (0 beyond nwv)
        if (tid < n_items) m = max(m, s_exm[tid]);
        m = wave_umax_dpp(m);
        if (lane == 0) s_wm[wv] = m;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t M = 0;
        for (int i = 0; i < 16; i++) M = max(M, s_wm[i]);
        s_M = M;
    }
    __syncthreads();
    const uint32_t M = s_M;
    if (M == 0) {  // stats is empty: max() raises ValueError in the reference (F6)
        if (tid == 0) {
            st->status = ST_EMPTY;
            st->count = 0;
            st->found = 0;
            st->sel_tie = 0;
        }
        return;
    }
    // ---- every pair that attains it ---------------------------------------------------------------
    auto row_at_max = [&](uint32_t x, uint32_t y) {
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
    if (su.x == M) {
        if (su.w == 1) {
            row_at_max(su.y, su.z);
        } else {  // several rows of this group of 64 attain it: look into the group
            const uint32_t s = atomicAdd(&s_nblk, 1u);
            if (s < LEAN_BLK_CAP) s_blk[s] = tid;
        }
    }
    if (tid < n_items && s_exm[tid] == M) row_at_max(s_exrow[tid], s_exarg[tid]);
    __syncthreads();
    const uint32_t nblk = s_nblk;
    if (nblk <= LEAN_BLK_CAP) {
        const uint2 *__restrict__ rowma = reinterpret_cast<const uint2 *>(rowmax);
        for (uint32_t i = wv; i < nblk; i += 16) {
            const uint32_t x = s_blk[i] * 64u + (uint32_t)lane;
            // (rows a, b, Z and the flagged rows were left out of the record: they came in above)
            if (x < vcur && x != sa && x != sb && x != sz && !flagged(x)) {
                const uint2 v = rowma[x];
                if (v.x == M) row_at_max(x, v.y);
            }
        }
    }
    __syncthreads();
    const uint32_t nrows = s_nrows;
    if (nblk <= LEAN_BLK_CAP && nrows <= ARGMAX_ROWS && s_nt <= TIE_CAP) {
        for (uint32_t r = 0; r < nrows; r++) {
            const uint32_t x = s_rows[r];
            const uint32_t *row = mat + (size_t)x * stride;
            for (uint32_t y = tid; y < vcur; y += 1024) {
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
    // more than TIE_CAP pairs (or groups / rows left unexamined): the general path decides
    const uint32_t nt = (nblk > LEAN_BLK_CAP || nrows > ARGMAX_ROWS) ? (TIE_CAP + 1) : min(s_nt, (uint32_t)TIE_CAP + 1);
    uint32_t pi = 0;
    unsigned long long pos = NOPOS;
    bool decided = (nt == 1);
    // Up to LEAN_CHAIN_TIES tied pairs: find EVERY pair's first occurrence -- the reference merges them in
    // that order for as long as their counts stay what they are (the pairs share no token: a merge leaves
    // every occurrence of the others in place, and their order with it) and nothing new reaches the tied
    // count (k_apply_lean watches that): the iterations that follow take the pairs off this list
    const bool chain = chain_on && nt > 1 && nt <= LEAN_CHAIN_TIES && gap == 0;
    if (!decided && nt <= TIE_CAP && gap == 0) {
        const unsigned long long key = tie_by_index(ref, C, s_tied, nt, chain ? s_pos : nullptr);
        if (key != NOPOS) {
            pi = (uint32_t)(key & 127u);
            pos = key >> 7;
            decided = true;
        }
    }
    uint32_t chain_len = 0;
    if (chain && decided) {
        // order of first occurrences (positions are distinct); a pair the index did not lead to ends the list
        if (tid == 0) s_cut = nt;
        __syncthreads();
        if (tid < nt) {
            const unsigned long long me = s_pos[tid];
            uint32_t rank = 0;
            for (uint32_t q = 0; q < nt; q++) rank += (s_pos[q] < me) | (s_pos[q] == me && q < tid);
            s_order[rank] = tid;
        }
        __syncthreads();
        // the list ends before the first pair that shares a token with an earlier one, has a == b (the
        // general path's merge) or was not found
        for (uint32_t k = tid; k < nt * nt; k += 1024) {
            const uint32_t j = k / nt, i = k % nt;
            if (j == 0 || i > j) continue;
            const uint32_t pj = s_order[j], pi2 = s_order[i];
            const int32_t aj = s_tied[2 * pj], bj = s_tied[2 * pj + 1];
            bool stop;
            if (i == j) {
                stop = (aj == bj) || s_pos[pj] == NOPOS;
            } else {
                const int32_t ai = s_tied[2 * pi2], bi = s_tied[2 * pi2 + 1];
                stop = (ai == aj) | (ai == bj) | (bi == aj) | (bi == bj);
            }
            if (stop) atomicMin(&s_cut, j);
        }
        __syncthreads();
        chain_len = s_cut;  // (>= 1: entry 0 is this iteration's pair)
        if (tid < chain_len) {
            st->chain[2 * tid] = s_tied[2 * s_order[tid]];
            st->chain[2 * tid + 1] = s_tied[2 * s_order[tid] + 1];
        }
    }
    if (tid == 0) {
        st->adj = 0;  // (the previous pass's format-B "adjacent sites" count was folded into the table)
        st->count = M;
        st->ntied = nt;
        st->firstpos = pos;
        st->sel_tie = 0;
        if (decided) {
            st->a = s_tied[2 * pi];
            st->b = s_tied[2 * pi + 1];
            st->fin_a = s_tied[2 * pi];
            st->fin_b = s_tied[2 * pi + 1];
            st->found = 1;
            st->chain_n = chain_len;
            st->chain_pos = 1;
        } else {
            st->found = 0;
            st->defer = 2;
        }
    }
}

// host: a deferred iteration is about to be re-run through the general path
__global__ void k_clear_defer(DevState *st) {
    st->defer = 0;
    st->chain_n = 0;  // (a chain of merges lined up by k_sel_lean ends here)
}

}  // namespace BPE_G
}  // namespace bpe
