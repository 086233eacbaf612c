// k_slots.hip -- K3 in the training loop: slotted merge, re-packing.
// Part of bpe_kernels.hip, which includes the parts in order.
// (no include guard: bpe_kernels.hip includes this part once per geometry, namespace BPE_G)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../bpe_device.h"

namespace bpe {
namespace BPE_G {

// ---------------------------------------------------------------------------
// Slotted merge (the training loop's default for a != b).
//
// The contiguous form moves every id every iteration (8N + 4N' bytes with the
// count pass) although late in training a merge touches a few ids per
// thousand.  Here the stream is a sequence of TILE-sized slots, each holding
// `len` ids at its start; a merge rewrites a slot only if one of its ids
// changes, into the same slot of the other buffer, and flips that slot's
// buffer bit.  No prefix sum, no second pass: one read of the ids (4N) plus the
// slots that actually change.  Stream order is slot order, so first-occurrence
// order (F3) is preserved; k_slot_compact restores a contiguous stream when the
// slots run low or when a == b needs the cross-tile pairing of the scan path.
//
// For a != b the carry into a slot is local knowledge: the previous slot's
// last id is a and my first word is b.

__global__ void __launch_bounds__(256)
k_slot_init(uint32_t *__restrict__ meta, uint4 *__restrict__ hdr, uint64_t T,
            const DevState *__restrict__ st, int par, uint32_t which, const uint32_t *__restrict__ ids) {
    const uint64_t n = st->n[par];
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += stride) {
        const uint64_t b0 = t * TILE;
        const uint32_t len = b0 >= n ? 0u : (uint32_t)min((uint64_t)TILE, n - b0);
        meta[t] = len | (which << 31);
        uint4 h = make_uint4(INVALID_WORD, INVALID_WORD, INVALID_WORD, INVALID_WORD);
        if (len > 0) h.x = ids[b0];
        if (len > 1) h.y = ids[b0 + 1];
        if (len > 2) h.z = ids[b0 + 2];
        if (len > 0) h.w = ids[b0 + len - 1];
        hdr[t] = h;
    }
}

template <bool DELTA>
__device__ __forceinline__ void merge_slot_tile(
    uint64_t t, const uint32_t *__restrict__ b0, const uint32_t *__restrict__ b1,
    uint32_t *__restrict__ w0, uint32_t *__restrict__ w1, const uint32_t *__restrict__ meta_in,
    uint32_t *__restrict__ meta_out, uint64_t T, DevState *st, int par, uint32_t newid,
    uint32_t *__restrict__ delta, uint32_t vcap, unsigned long long *__restrict__ sdesc, uint32_t epoch,
    const uint4 *__restrict__ hdr_in, uint4 *__restrict__ hdr_out) {
    __shared__ int s_wave[MT / 64];
    __shared__ uint32_t s_wsum[MT / 64];
    __shared__ uint32_t s_ctx[9];  // halo[0..2], previous last word, my header x, carry (a == b), my header y z w
    SlotRef ref;
    ref.b0 = b0;
    ref.b1 = b1;
    ref.meta = meta_in;
    ref.T = T;
    uint32_t a, b;
    if (!resolved_pair(st, ref, 0, a, b)) {
        if (t == 0 && threadIdx.x == 0) st->status = ST_INTERNAL;
        return;
    }
    const uint32_t mi = meta_in[t];
    const int len = (int)(mi & 0x7FFFFFFFu);
    if (t == 0 && threadIdx.x == 0) {
        st->fin_a = (int32_t)a;
        st->fin_b = (int32_t)b;
    }
    if (len == 0) {
        if (threadIdx.x == 0) {
            meta_out[t] = mi;
            hdr_out[t] = hdr_in[t];
        }
        return;
    }
    const uint32_t cur = mi >> 31;
    const uint32_t *src = (cur ? b1 : b0) + t * TILE;
    if (threadIdx.x == 0) {
        // The three words after my slot and the word before it, in stream order.  Every slot
        // keeps {first three words, last word} in a header array, so in the common case these
        // are independent loads that fly together with the slot's own; only a neighbour with
        // fewer than 3 ids sends us walking.
        const uint4 hme = hdr_in[t];
        uint32_t mn = 0, mp = 0;
        uint4 hn = make_uint4(INVALID_WORD, INVALID_WORD, INVALID_WORD, INVALID_WORD), hp = hn;
        if (t + 1 < T) {
            mn = meta_in[t + 1];
            hn = hdr_in[t + 1];
        }
        if (t > 0) {
            mp = meta_in[t - 1];
            hp = hdr_in[t - 1];
        }
        uint32_t h0 = hn.x, h1 = hn.y, h2 = hn.z;
        if (t + 1 < T && (mn & 0x7FFFFFFFu) < 3) {  // rare: gather across short / empty slots
            h0 = h1 = h2 = INVALID_WORD;
            int got = 0;
            for (uint64_t u = t + 1; u < T && got < 3; u++) {
                const uint32_t mu = meta_in[u];
                const uint32_t lu = mu & 0x7FFFFFFFu;
                const uint32_t *pu = ((mu >> 31) ? b1 : b0) + u * TILE;
                for (uint32_t i = 0; i < lu && got < 3; i++) {
                    const uint32_t w = pu[i];
                    if (got == 0) h0 = w; else if (got == 1) h1 = w; else h2 = w;
                    got++;
                }
            }
        }
        uint32_t prev = hp.w;
        if (t > 0 && (mp & 0x7FFFFFFFu) == 0) {  // rare: previous slot is empty
            prev = INVALID_WORD;
            for (uint64_t u = t; u-- > 0;) {
                const uint32_t mu = meta_in[u];
                const uint32_t lu = mu & 0x7FFFFFFFu;
                if (lu) {
                    prev = (((mu >> 31) ? b1 : b0) + u * TILE)[lu - 1];
                    break;
                }
            }
        }
        s_ctx[0] = h0;
        s_ctx[1] = h1;
        s_ctx[2] = h2;
        s_ctx[3] = prev;
        s_ctx[4] = hme.x;  // my first word (len > 0)
        s_ctx[6] = hme.y;
        s_ctx[7] = hme.z;
        s_ctx[8] = hme.w;
    }
    SlotRaw raw;
    slot_raw_load(raw, src, len);
    __syncthreads();
    const uint32_t halo[3] = {s_ctx[0], s_ctx[1], s_ctx[2]};
    const uint32_t prev = s_ctx[3];
    const uint32_t s_first_word = s_ctx[4];
    Tile tl;
    tile_from_slot(tl, raw, len, halo);
    tile_rbits(tl, a, b);
    // carry: the previous slot ended with a site start iff its last id is a and my first word is b
    // (thread 0 stored my first word next to the neighbours' in s_ctx)
    uint32_t s = (uint32_t)((prev != INVALID_WORD) & ((prev & IDMASK) == a) & ((s_first_word & NWMASK) == b));
    if (a == b) {
        // a == b: the carry is the PARITY of the run of a's that ends at the previous slot's
        // last id (F2).  Walk that run backwards, 64 ids per step; only if it swallows the whole
        // previous slot does this tile need that slot's own carry (published below by every
        // tile; tiles are dispatched in order, so the wait is on a running or finished tile).
        const unsigned long long tag = ((unsigned long long)(epoch & EPOCH_MASK)) << 42;
        if (wave_id() == 0) {
            const int lane = lane_id();
            uint32_t sc = 0;
            bool failed = false;
            if (s) {  // the boundary pair matches: r[last of previous slot] = 1
                uint64_t u = t;
                uint32_t mu = 0;
                while (u-- > 0) {
                    mu = meta_in[u];
                    if (mu & 0x7FFFFFFFu) break;
                }
                const int lu = (int)(mu & 0x7FFFFFFFu);
                const uint32_t *pu = ((mu >> 31) ? b1 : b0) + u * TILE;
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
                        const unsigned long long d = desc_load(&sdesc[u]);
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
                desc_store(&sdesc[t], tag | (1ull << 62) | sc);
                if (failed) atomicExch(&st->status, ST_LOOKBACK);
            }
        }
        __syncthreads();
        s = s_ctx[5];
    }
    // Fast path: no match at any owned position, none at the first word after the slot, no
    // carry -> nothing in this slot changes and it owes no pair-table update.  Late in training
    // this is most slots; they skip the scans and the rewrite altogether.
    {
        uint32_t anyr = s;
        const int qw = wave_id() * WAVE_SPAN + lane_id() * 4;
#pragma unroll
        for (int j = 0; j < MJ; j++) {
            const int q0 = qw + j * 256;
            // keep the bits of positions q <= len
            const int nb = len + 1 - q0;
            const uint32_t keep = nb >= 4 ? 0xFu : (nb <= 0 ? 0u : ((1u << nb) - 1u));
            anyr |= tl.rb[j] & keep;
        }
        // a full slot: the word after it is the last wave's tail, not one of my registers
        if (len == TILE && wave_id() == MT / 64 - 1)
            anyr |= (uint32_t)(((tl.tail[0] & IDMASK) == a) & ((tl.tail[1] & NWMASK) == b));
        if (!__syncthreads_or((int)(anyr != 0))) {
            if (threadIdx.x == 0) {
                meta_out[t] = mi;
                hdr_out[t] = make_uint4(s_ctx[4], s_ctx[6], s_ctx[7], s_ctx[8]);
            }
            return;
        }
    }
    tile_lzscan(tl, s_wave);
    uint32_t kept = 0;
    bool changed = false;
    uint32_t *dst = (cur ? w0 : w1) + t * TILE;  // the OTHER buffer
    uint32_t *my_hdr = reinterpret_cast<uint32_t *>(hdr_out + t);
    tile_rewrite<DELTA, true>(tl, s, a, b, newid, dst, s_wsum, delta, vcap, len, &kept, &changed, my_hdr);
    if (threadIdx.x == 0) {
        if (changed) {
            meta_out[t] = kept | ((cur ^ 1u) << 31);
            atomicAdd(&st->removed, (unsigned long long)((uint32_t)len - kept));
            for (uint32_t i = kept; i < 3; i++) my_hdr[i] = INVALID_WORD;  // fewer than 3 ids left
            if (kept == 0) my_hdr[3] = INVALID_WORD;
        } else {
            meta_out[t] = mi;
            hdr_out[t] = make_uint4(s_ctx[4], s_ctx[6], s_ctx[7], s_ctx[8]);
        }
    }
}

// One workgroup per slot.  (A resident grid striding over the slots was tried: the loop
// costs 55 more VGPRs -- occupancy 6 -> 3 -- and measured 20 % slower.)
template <bool DELTA>
__global__ void __launch_bounds__(MT)
k_merge_slot(const uint32_t *__restrict__ b0, const uint32_t *__restrict__ b1, uint32_t *__restrict__ w0,
             uint32_t *__restrict__ w1, const uint32_t *__restrict__ meta_in,
             uint32_t *__restrict__ meta_out, uint64_t T, DevState *st, int par, uint32_t newid,
             uint32_t *__restrict__ delta, uint32_t vcap, uint32_t *dirty_n,
             unsigned long long *__restrict__ sdesc, uint32_t epoch, const uint4 *__restrict__ hdr_in,
             uint4 *__restrict__ hdr_out) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && dirty_n) *dirty_n = 0;
    if (blockIdx.x >= T || st->status) return;
    merge_slot_tile<DELTA>(blockIdx.x, b0, b1, w0, w1, meta_in, meta_out, T, st, par, newid, delta, vcap,
                           sdesc, epoch, hdr_in, hdr_out);
}

// slots -> contiguous: tile t's ids go to out[off[t] ...]; the stream length is left in st->n[par]
__global__ void __launch_bounds__(256)
k_slot_lens(const uint32_t *__restrict__ meta, uint64_t T, uint32_t *__restrict__ lens) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += stride)
        lens[t] = meta[t] & 0x7FFFFFFFu;
}
__global__ void __launch_bounds__(256)
k_slot_compact(const uint32_t *__restrict__ b0, const uint32_t *__restrict__ b1,
               const uint32_t *__restrict__ meta, const unsigned long long *__restrict__ off,
               uint32_t *__restrict__ out) {
    const uint64_t t = blockIdx.x;
    const uint32_t m = meta[t];
    const uint32_t len = m & 0x7FFFFFFFu;
    const uint32_t *src = ((m >> 31) ? b1 : b0) + t * TILE;
    uint32_t *dst = out + off[t];
    for (uint32_t i = threadIdx.x; i < len; i += 256) dst[i] = src[i];
}
__global__ void k_set_status(DevState *st, uint32_t status) { st->status = status; }
__global__ void k_move_n(DevState *st, int from, int to) { st->n[to] = st->n[from]; }

}  // namespace BPE_G
}  // namespace bpe
