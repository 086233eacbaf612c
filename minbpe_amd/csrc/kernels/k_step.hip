// k_step.hip -- a sparse chain step as ONE launch: selection -> merge pass -> table update.
//
// A chain step (k_chain.hip, k_pool.hip) used to be three launches -- k_pool_sel, k_merge_chain, k_apply_chain -- and late
// in training, where a batch of ~7 merges has a few thousand candidate slots, a step was nothing but their hand-overs:
// every launch pays its ramp and its drain (4.5 us for a grid of 256 x 1024 threads that does nothing), and every
// workgroup of every launch starts by fetching the step's state (60-68 us per step from merge 8000 on, 9 us per merge,
// round after round: profiles/r5_notes.md).  Here the three parts are phases of one resident grid (one 1024-thread
// workgroup per CU at most, all of them co-resident):
//
//   S  workgroup 0 selects (pool_sel_body, k_pool.hip) while workgroups 1 .. nscan do a rebuild's row work exactly as in
//      k_pool_sel; everybody else waits for the PUBLISHED LINE: 32 self-validating 8-byte words (launch tag << 32 | value)
//      that carry the batch itself -- K, the new ids, the pairs, "the flagged rows were re-scanned" -- so a workgroup that
//      sees the line needs no further round trip to start its merge pass (the data is the flag);
//   M  every workgroup below `gm` works through its share of the candidate mask (merge_chain_body, k_chain.hip);
//   B  one grid barrier WITHOUT cache maintenance (3.8 us; with the L2s cleaned by fences 14-120 us, tools/atomic_peak.hip):
//      the batch's delta vectors and the staged headers are complete -- what crosses it is written and read at agent
//      scope (k_common.hip);
//   A  the table update (apply_chain_tokens / _records / _commit, k_chain.hip) by the whole grid.  The flag words of the
//      rows to re-scan are cleared HERE (the wave that owns a word stores it whole, starting from zero when the selection
//      re-scanned every flagged row) -- the three-launch form cleared them at the head of the merge pass, which in one
//      launch would race with a scanning workgroup that has not copied them yet.
//
// Nothing is decided from state another workgroup writes in the same launch: workgroups other than 0 take the batch from
// the line, the hint from st->pool_hint (written only by phase A: pool_hint_next), everything else after the barrier.
// Every wait is bounded (LOOKBACK_SPINS): a workgroup that never sees the line or the barrier raises ST_LOOKBACK, and the
// host fails the training instead of hanging.  The barrier counter only ever grows (the host hands every launch its
// own target), so a launch cut short cannot strand the next one on a stale count.
// Sharded training keeps the three launches (its collectives sit between them), and so do the dense steps.
// Part of bpe_kernels.hip, which includes the parts in order.
// (no include guard: bpe_kernels.hip includes this part once per geometry, namespace BPE_G)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../bpe_device.h"

namespace bpe {
namespace BPE_G {

// the published line: [0] = K | brep << 8 | ran << 16 | noop << 17 | next hint << 18 | defer << 19 | mode << 21, [1] = 256 + merges done so far
// (= the batch's first new id), [2 + p] = a_p << 16 | b_p, [17 + p] = the count of pair p
constexpr uint32_t STEP_PUB_WORDS = 32;
static_assert(2 + 2 * CH_KSWEEP == STEP_PUB_WORDS, "the line holds the pairs and their counts");

union StepLds {
    PoolLds p;
    MergeLds m;
};

// All workgroups of the grid have arrived; false: timed out.  NO cache maintenance (see k_common.hip: a fence that cleans
// the L2s costs 14-120 us per barrier): every thread waits for its own memory operations to be acknowledged, one thread
// per workgroup counts in and polls.  What crosses the barrier is therefore written with device atomics or agent-scope
// stores and read with agent-scope loads -- the delta words, adj counts, staged headers and their mask words, removal
// counters, the re-scanned rows' maxima, the status word; everything else a phase reads was written by an earlier LAUNCH
// or by its own workgroup.
__device__ __forceinline__ bool step_grid_barrier(uint32_t *ctr, uint32_t target, uint32_t *s_flag) {
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t ok = 0;
        for (uint32_t spins = 0; spins < LOOKBACK_SPINS; spins++) {
            if ((int32_t)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) {
                ok = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        *s_flag = ok;
    }
    __syncthreads();
    return *s_flag != 0;
}

__global__ void __launch_bounds__(LEAN_MT)
k_step(StepArgs S) {
    __shared__ StepLds U;
    __shared__ uint32_t s_b[STEP_PUB_WORDS], s_adj[CH_KMAX];
    __shared__ uint32_t s_flag;
    DevState *st = S.st;
    const uint32_t tid = threadIdx.x;
    // (debug, BPE_STEP_STAMPS: the 100 MHz clock at the phase boundaries of workgroups 0, gm / 2 and the last one)
    const int swho = blockIdx.x == 0 ? 0 : (blockIdx.x == S.gm / 2 ? 1 : (blockIdx.x == gridDim.x - 1 ? 2 : -1));
    auto stamp = [&](int i) {
        if (S.stamps && tid == 0 && swho >= 0) S.stamps[(((size_t)(S.step % STEP_STAMP_RING)) * 3 + (size_t)swho) * 16 + (size_t)i] = wall_clock64();
    };
    unsigned long long *const dbg = (S.stamps && swho >= 0) ? S.stamps + (((size_t)(S.step % STEP_STAMP_RING)) * 3 + (size_t)swho) * 16 : nullptr;
    stamp(0);
    // ================= S: the selection (workgroup 0 decides, 1 .. nscan scan rows when told to) ==========================
    if (blockIdx.x == 0) {
        pool_sel_body(S.rowmax, S.mat, S.stride, st, S.ref, S.C, S.dbits, S.res, S.tag, S.req, S.kcap, S.pool, S.gather,
                      S.hint_below, nullptr, 0ull, S.pool + PL_CAP, U.p, 0u, 1u + S.nscan, dbg);
        __builtin_amdgcn_s_waitcnt(0);  // (the adj words it zeroed are in memory before anybody is told to add to them)
        __syncthreads();
        stamp(1);
        if (tid < STEP_PUB_WORDS) {
            // (what thread 0 of this workgroup left in st: the same addresses read back through the same L2)
            const uint32_t status = st->status, defer = st->defer;
            const uint32_t noop = (status || defer) ? 1u : 0u;
            const uint32_t K = noop ? 0u : min(st->bk, (uint32_t)CH_KSWEEP);
            uint32_t v = 0;
            if (tid == 0)
                v = K | (st->brep << 8) | ((st->sel_ran ? 1u : 0u) << 16) | (noop << 17) | ((st->pool_hint_next ? 1u : 0u) << 18) |
                    ((defer & 3u) << 19) | ((st->sel_mode & 1u) << 21);
            else if (tid == 1) v = 256u + st->iter;
            else if (tid - 2 < K) v = ((uint32_t)st->ba[tid - 2] << 16) | ((uint32_t)st->bb[tid - 2] & 0xFFFFu);
            else if (tid >= 2 + CH_KSWEEP && tid - (2 + CH_KSWEEP) < K) v = st->bcnt[tid - (2 + CH_KSWEEP)];
            s_b[tid] = v;
            granule_put(S.pub + tid, S.tag, v);
        }
        if (tid == 0) s_flag = 1;
    } else {
        if (blockIdx.x <= S.nscan)
            pool_sel_body(S.rowmax, S.mat, S.stride, st, S.ref, S.C, S.dbits, S.res, S.tag, S.req, S.kcap, S.pool, S.gather,
                          S.hint_below, nullptr, 0ull, S.pool + PL_CAP, U.p, blockIdx.x, 1u + S.nscan);
        __syncthreads();
        if (tid == 0) s_flag = 1;
        __syncthreads();
        if (tid < STEP_PUB_WORDS) {
            uint32_t v = 0;
            if (!granule_get(S.pub + tid, S.tag, v)) s_flag = 0;
            s_b[tid] = v;
        }
    }
    __syncthreads();
    if (!s_flag) {  // the line never came: never merge on a guess (the barrier below still counts this workgroup)
        if (tid == 0) atomicExch(&st->status, ST_LOOKBACK);
        if (tid < STEP_PUB_WORDS) s_b[tid] = tid == 0 ? (1u << 17) : 0u;
    }
    __syncthreads();
    stamp(2);
    if (S.stamps && tid == 0 && blockIdx.x < 256)
        S.stamps[(size_t)STEP_STAMP_RING * 3 * 16 + ((size_t)(S.step % STEP_STAMP_RING) * 256 + blockIdx.x) * 2] = wall_clock64();
    const uint32_t K = s_b[0] & 0xFFu, brep = (s_b[0] >> 8) & 0xFFu, ran = (s_b[0] >> 16) & 1u, z0 = s_b[1];
    const bool sel_noop = ((s_b[0] >> 17) & 1u) != 0 || K == 0;
    // ================= M: the merge pass =======================================================================================
    if (!sel_noop && blockIdx.x < S.gm) {
        MergeLds &L = U.m;
        if (tid < CH_KMAX) {
            const uint32_t w = tid < K ? s_b[2 + tid] : 0u;
            L.s_pa[tid] = tid < K ? (w >> 16) : 0xFFFFFFFFu;
            L.s_pb[tid] = tid < K ? (w & 0xFFFFu) : 0xFFFFFFFFu;
            L.s_pb1[tid + 1] = tid < K ? (w & 0xFFFFu) : 0xFFFFFFFFu;
            if (tid == 0) L.s_pb1[0] = 0xFFFFFFFFu;  // (a masked word has its weight bits clear: never equal)
        }
        __syncthreads();
        merge_chain_body<true>(S.A, S.idx_dirty, S.use_index, L, K, z0, brep, blockIdx.x, S.gm, blockIdx.x == 0 ? nullptr : dbg);
    }
    // ================= B: every delta word, staged header and flag of this step is in memory ==================================
    stamp(3);
    // (debug: when every workgroup got the line and when it reached the barrier -- who is the last one?)
    if (S.stamps && tid == 0 && blockIdx.x < 256) {
        unsigned long long *all = S.stamps + (size_t)STEP_STAMP_RING * 3 * 16 + ((size_t)(S.step % STEP_STAMP_RING) * 256 + blockIdx.x) * 2;
        all[1] = wall_clock64();
    }
    const bool arrived = step_grid_barrier(S.bar, S.bar_target, &s_flag);
    stamp(4);
    if (!arrived && tid == 0) atomicExch(&st->status, ST_LOOKBACK);
    // ================= A: the table update ====================================================================================
    const uint32_t status = arrived ? ld_agent(&st->status) : (uint32_t)ST_LOOKBACK;
    const bool noop = status || ((s_b[0] >> 19) & 3u) || K == 0;  // (defer: from the line)
    if (!noop) {
        // a token per thread, 64 tokens per wave, dealt over ALL workgroups (a wave or two each: the update is a chain of
        // round trips per wave, not a matter of bandwidth)
        const uint32_t nwv4 = (((z0 + K - 1u) >> 8) + 1u) * 4u;  // waves of tokens, in blocks of 256 tokens
        const uint32_t wpw = (nwv4 + gridDim.x - 1u) / gridDim.x;
        const uint32_t tw = blockIdx.x * wpw + (uint32_t)wave_id();
        // (the adj counts: device atomics of this launch's merge pass -- agent-scope loads, into LDS for everybody)
        if (tid < CH_KMAX) s_adj[tid] = K == 1 ? (tid == 0 ? ld_agent(&st->adj) : 0u) : (tid < K ? ld_agent(&st->badj[tid]) : 0u);
        __syncthreads();
        if ((uint32_t)wave_id() < wpw && tw < nwv4)
            apply_chain_tokens<true>(tw * 64u + (uint32_t)lane_id(), S.mat, S.stride, S.delta, S.dl, S.rowmax, S.dbits, S.sums,
                                     nullptr, 0u, nullptr, K, z0, ran, s_b + 2, s_adj, brep);
    } else if (ran && arrived) {  // (nothing merged, but the selection did re-scan the flagged rows: nobody sets a flag in this phase)
        for (uint32_t i = blockIdx.x * (uint32_t)LEAN_MT + tid; i < (uint32_t)DBITS_WORDS; i += gridDim.x * (uint32_t)LEAN_MT) S.dbits[i] = 0;
    }
    stamp(5);
    // ---- the step's records.  To the HOST (pinned memory, PCIe round trips: ~5 us) by the last workgroup, which has no
    // tokens to update in a full-size grid -- everything it writes it has from the line.  The state words in st by workgroup
    // 0 alone: it wrote their neighbours during the selection, and one line of st must not sit dirty in two L2s.
    if ((blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && wave_id() == 0) {
        const uint32_t lane = (uint32_t)lane_id();
        const uint32_t defer2 = (s_b[0] >> 19) & 3u, mode_used = (s_b[0] >> 21) & 1u, hint_next = (s_b[0] >> 18) & 1u;
        // ids removed by the merge pass: CH_RMV counters per pair (lane l: counters 4l .. 4l + 3, all of pair l / 4), read at
        // agent scope (device atomics of this launch) -- or, an unweighted stream, nobody counted: a merge of a != b removes
        // exactly as many ids as the pair counts (base.py:25-41)
        uint32_t v = 0;
        if (S.removed) {
#pragma unroll
            for (int i = 0; i < 4; i++) v += ld_agent(&S.removed[(lane * 4 + i) * REMOVED_STRIDE]);
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
        if (!S.removed) {
#pragma unroll
            for (int p = 0; p < CH_KMAX; p++) rem[p] = p < CH_KSWEEP ? s_b[2 + CH_KSWEEP + p] : 0u;
        }
        const uint32_t iter = z0 - 256u;
        unsigned long long nn = st->n[S.par];  // (an earlier launch's: workgroup 0 writes the OTHER parity below)
        if (!noop)
            for (uint32_t p = 0; p < K; p++) nn -= rem[p];
        if (blockIdx.x == 0) {
            if (S.removed && lane < 64) {  // (only now: the last workgroup reads the counters from memory, where they still stand)
#pragma unroll
                for (int i = 0; i < 4; i++) S.removed[(lane * 4 + i) * REMOVED_STRIDE] = 0;
            }
            if (lane == 0) {
                if (!noop) st->iter = iter + K;
                if (status == 0) st->n[S.par ^ 1] = nn;  // (a step that merged nothing carries the length forward)
                st->removed = 0;
                st->pool_hint = hint_next;  // (k_pool.hip: what this step's selection announced for the next one)
                if (arrived) st->sel_ran = 0;
            }
        }
        if (blockIdx.x == gridDim.x - 1 && lane == 0) {
            if (!noop) {
                unsigned long long m = st->n[S.par];
                for (uint32_t p = 0; p < K; p++) {
                    m -= rem[p];
                    iter_rec_put(S.rec + iter + p, (int32_t)(s_b[2 + p] >> 16), (int32_t)(s_b[2 + p] & 0xFFFFu), s_b[2 + CH_KSWEEP + p], ST_OK, m);
                }
                __builtin_amdgcn_s_waitcnt(0);
                for (uint32_t p = 0; p < K; p++) iter_rec_seal(S.rec + iter + p, (unsigned long long)(iter + p) + 1);
                __builtin_amdgcn_s_waitcnt(0);  // (the step record below tells the host that these are final)
            }
            StepRec *sr = S.srec + (S.step % STEP_RING);
            // (pad: the step's mode | defer << 8 -- 1 = a == b heads the pool, 2 = a tie the step could not settle)
            step_rec_put(sr, iter, noop ? 0u : K, (status == 0 && defer2) ? (uint32_t)ST_DEFER : status, mode_used | (defer2 << 8), nn);
            __builtin_amdgcn_s_waitcnt(0);
            step_rec_seal(sr, (unsigned long long)S.step + 1);
        }
    }
    stamp(6);
    if (noop) return;
    apply_chain_commit<true>(blockIdx.x * (uint32_t)LEAN_MT + tid, gridDim.x * (uint32_t)LEAN_MT, S.nwords, S.smask, S.stage, S.hdr_cur);
    stamp(7);
}

}  // namespace BPE_G
}  // namespace bpe
