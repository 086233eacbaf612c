// k_dp.hip -- data-parallel (sharded chunks) helpers.
// Part of bpe_kernels.hip, which includes the parts in order.
// (no include guard: bpe_kernels.hip includes this part once per geometry, namespace BPE_G)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../bpe_device.h"

namespace bpe {
namespace BPE_G {

// ---------------------------------------------------------------------------
// Data-parallel training over sharded chunks: every rank holds a
// contiguous range of chunks and a replica of the GLOBAL pair table.  Per
// iteration the ranks exchange (1) two 64-bit words that decide the tie-break
// and (2) the four delta vectors -- never ids, never the table.
//
// Tie-break across ranks: global first occurrence = lowest (rank, local
// position).  w0 = key<<16 | a, w1 = key<<16 | b with key = rank<<32 | pos: the
// element-wise MIN all-reduce of (w0, w1) returns the pair of the winning rank,
// because the keys are distinct per rank.  No tie: every rank sends key 0 and
// the same pair.  No local occurrence: INT64_MAX.
template <class Ref>
__global__ void k_dp_key(Ref ref, int par, const DevState *__restrict__ st,
                         unsigned long long rank, long long *__restrict__ key) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    long long w0 = 0x7FFFFFFFFFFFFFFFll, w1 = 0x7FFFFFFFFFFFFFFFll;
    if (st->status == 0) {
        if (st->found) {
            w0 = (long long)(uint32_t)st->a;
            w1 = (long long)(uint32_t)st->b;
        } else if (st->firstpos != NOPOS) {
            // (slot-space positions are < 2^32 too: the slot area never exceeds the original stream)
            const unsigned long long k = ((rank << 32) | st->firstpos) + 1;  // > 0: a tie never ties with "no tie"
            uint32_t x0 = 0, x1 = 0;
            slot_get(ref, st->n[par], st->firstpos, x0);
            slot_next(ref, st->n[par], st->firstpos, x1);
            w0 = (long long)((k << 16) | (x0 & IDMASK));
            w1 = (long long)((k << 16) | (x1 & IDMASK));
        }
    }
    key[0] = w0;
    key[1] = w1;
    // third word: this rank's status, negated.  The MIN all-reduce hands every rank the worst one,
    // so a rank-local failure (a bounded wait that timed out, a consistency check) stops ALL ranks
    // at this same merge instead of leaving the peers to merge on without this shard.
    key[2] = -(long long)st->status;
}
__global__ void k_dp_resolve(DevState *st, const long long *__restrict__ key) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (st->status) return;
    if (key[2] < 0) {  // some rank failed (ST_EMPTY is replica-identical and never gets here)
        st->status = ST_INTERNAL;
        return;
    }
    if (key[0] == 0x7FFFFFFFFFFFFFFFll) {
        st->status = ST_INTERNAL;  // a tie at the maximum, yet no rank holds a tied pair
        return;
    }
    st->a = (int32_t)(key[0] & 0xFFFF);
    st->b = (int32_t)(key[1] & 0xFFFF);
    st->fin_a = st->a;
    st->fin_b = st->b;
    st->found = 1;
}
// second slotted form, sharded: the pair is known only now (after the MIN all-reduce), so the
// candidate list of a sparse pass is made here instead of inside k_select (one 1024-thread block)
__global__ void __launch_bounds__(1024)
k_dp_cand(DevState *st, CandArgs C) {
    if (st->status || !st->found || (st->a == st->b && !C.aa)) return;
    build_cand_list(C, st, (uint32_t)st->a, (uint32_t)st->b);  // (a == b: the slots that may hold (a,a) and the slot before each)
}
// ... and the fold of its replicated delta vectors (stride / replica count packed in `dl` as for
// the merge kernels) into the all-reduce payload [4][vcap] + adj (format B fills vectors 0 and 1)
__global__ void __launch_bounds__(256)
k_dp_fold2(uint32_t *__restrict__ delta, uint32_t dl, uint32_t Z, uint32_t *__restrict__ folded, uint32_t vcap,
           const DevState *__restrict__ st) {
    const uint32_t nrep = 1u << (dl >> 24), ds = dl & 0xFFFFFFu;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        folded[4 * (size_t)vcap] = st->status ? 0u : st->adj;
        // (summed over the ranks: a failure inside any rank's merge pass is known to all before the table update of
        // this same merge -- k_dp_after_sum, k_chain.hip)
        folded[4 * (size_t)vcap + 8] = st->status ? 1u : 0u;
    }
    if (t >= vcap) return;
#pragma unroll
    for (int v = 0; v < 4; v++) {
        uint32_t acc = 0;
        if (t <= Z && t < ds) {
            for (uint32_t r = 0; r < nrep; r++) {
                const uint32_t x = delta[delta_rep_off(r, ds) + (size_t)v * ds + t];
                if (x) delta[delta_rep_off(r, ds) + (size_t)v * ds + t] = 0;
                acc += x;
            }
        }
        folded[(size_t)v * vcap + t] = acc;
    }
}
// fold the replicated delta vectors into one compact 4 x vcap buffer (the SUM all-reduce payload)
__global__ void __launch_bounds__(256)
k_dp_fold(uint32_t *__restrict__ delta, uint32_t vcap, uint32_t Z, uint32_t *__restrict__ folded) {
    const uint32_t nrep = 1u << (vcap >> 24);
    vcap &= 0xFFFFFFu;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= vcap) return;
#pragma unroll
    for (int v = 0; v < 4; v++) {
        uint32_t acc = 0;
        if (t <= Z) {
            for (uint32_t r = 0; r < nrep; r++) {
                const uint32_t x = delta[delta_rep_off(r, vcap) + (size_t)v * vcap + t];
                if (x) delta[delta_rep_off(r, vcap) + (size_t)v * vcap + t] = 0;
                acc += x;
            }
        }
        folded[(size_t)v * vcap + t] = acc;
    }
}

// The first all-reduce payload of a sharded job: the 256 x 256 block of byte-pair counts as 16-bit LIMBS -- out[i] = the
// low half of this shard's count i, out[65536 + i] = its high half -- so that the INT32 sum over up to 1024 ranks cannot
// wrap (each limb sum stays below 2^26) and every rank sees the exact GLOBAL count: k_dp_table_join writes
// the replica's table from the summed limbs and raises `over` when a pair occurs 2^32 times or more in the whole job
// (table entries, and every later count, are 32-bit: every later count is bounded by the largest of these).
__global__ void __launch_bounds__(256)
k_dp_table_split(const uint32_t *__restrict__ mat, uint32_t stride, uint32_t *__restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;  // row blockIdx.x, column threadIdx.x
    const uint32_t v = mat[(size_t)blockIdx.x * stride + threadIdx.x];
    out[i] = v & 0xFFFFu;
    out[65536u + i] = v >> 16;
}
__global__ void __launch_bounds__(256)
k_dp_table_join(const uint32_t *__restrict__ in, uint32_t *__restrict__ mat, uint32_t stride, unsigned long long *__restrict__ over) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const unsigned long long g = ((unsigned long long)in[65536u + i] << 16) + in[i];
    if (g >> 32) atomicMax(over, g);
    mat[(size_t)blockIdx.x * stride + threadIdx.x] = (uint32_t)g;
}

}  // namespace BPE_G
}  // namespace bpe
