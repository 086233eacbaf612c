// k_encode.hip -- K4: batch encode.
// Part of bpe_kernels.hip, which includes the parts in order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../bpe_device.h"
#include "k_common.hip"

namespace bpe {

// ---------------------------------------------------------------------------
// K4: encode  (_encode_chunk regex.py:92-109 == basic.py:57-74, batched)
//
// The reference repeatedly merges the lowest-rank pair present in a chunk.
// Merging the LEFTMOST lowest-rank pair, one occurrence at a time, is the same
// computation (pairs created by a merge of rank r involve token 256+r and so
// have rank > r; left-to-right order reproduces the greedy a==b pairing).
//
// Ranks live in an open-addressing hash table (key = a<<32|b, value = rank),
// a few hundred KB, L2-resident.  Short chunks -- virtually all of them under a
// GPT-style split pattern (mean ~4 bytes) -- are encoded one chunk per lane
// with the token list in lane-private LDS columns.  Chunks longer than
// ENC_LMAX tokens are queued and encoded by stream-wide rounds (bpe_api.hip).

__device__ __forceinline__ uint32_t rank_lookup(const unsigned long long *__restrict__ keys,
                                                const uint32_t *__restrict__ vals, uint32_t mask,
                                                uint32_t a, uint32_t b) {
    const unsigned long long key = ((unsigned long long)a << 32) | b;
    uint32_t h = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 40) & mask;
    for (;;) {
        const unsigned long long k = keys[h];
        if (k == key) return vals[h];
        if (k == ~0ull) return 0xFFFFFFFFu;
        h = (h + 1) & mask;
    }
}

// The merge loop of one chunk, by one lane: tokens tok[i * ENC_THREADS] (lane-private LDS column),
// i < L, already filled with the chunk's bytes; rk = the rank column.  Returns the new length.
template <typename TT>
__device__ __forceinline__ uint32_t encode_lane(TT *tok, TT *rk, uint32_t L, const unsigned long long *__restrict__ keys,
                                                const uint32_t *__restrict__ vals, uint32_t mask,
                                                const int32_t *__restrict__ merge_ids) {
    constexpr uint32_t NONE = (uint32_t)(TT)0xFFFFFFFFu;  // "no rank" in storage
    for (uint32_t i = 0; i + 1 < L; i++)
        rk[i * ENC_THREADS] = (TT)rank_lookup(keys, vals, mask, tok[i * ENC_THREADS], tok[(i + 1) * ENC_THREADS]);
    while (L >= 2) {
        uint32_t best = NONE, bi = 0;
        for (uint32_t i = 0; i + 1 < L; i++) {
            const uint32_t r = rk[i * ENC_THREADS];
            if (r < best) {  // strict: leftmost occurrence of the lowest rank
                best = r;
                bi = i;
            }
        }
        if (best == NONE) break;  // nothing else can be merged
        tok[bi * ENC_THREADS] = (TT)(merge_ids ? (uint32_t)merge_ids[best] : 256u + best);
        for (uint32_t i = bi + 1; i + 1 < L; i++) {
            tok[i * ENC_THREADS] = tok[(i + 1) * ENC_THREADS];
            rk[i * ENC_THREADS] = rk[(i + 1) * ENC_THREADS];
        }
        L--;
        if (bi > 0)
            rk[(bi - 1) * ENC_THREADS] =
                (TT)rank_lookup(keys, vals, mask, tok[(bi - 1) * ENC_THREADS], tok[bi * ENC_THREADS]);
        if (bi + 1 < L)
            rk[bi * ENC_THREADS] =
                (TT)rank_lookup(keys, vals, mask, tok[bi * ENC_THREADS], tok[(bi + 1) * ENC_THREADS]);
    }
    return L;
}

// TT: storage type of the lane-private token / rank columns.  uint16_t when every id and rank
// fits (vocabularies up to 65535: 32 KiB of LDS per workgroup, 5 workgroups per CU -- the kernel
// is bound by the latency of the rank look-ups, so resident waves are what counts); uint32_t
// otherwise (cl100k-sized rank tables: 64 KiB, 2 per CU).
template <typename TT>
__global__ void __launch_bounds__(ENC_THREADS)
k_encode_short(const uint8_t *__restrict__ bytes, const uint64_t *__restrict__ off, uint64_t n_chunks,
               uint64_t n, const unsigned long long *__restrict__ keys,
               const uint32_t *__restrict__ vals, uint32_t mask, const int32_t *__restrict__ merge_ids,
               uint32_t *__restrict__ tmp, uint32_t *__restrict__ outlen,
               unsigned long long *__restrict__ long_list, unsigned long long *__restrict__ n_long) {
    __shared__ TT s_tok[ENC_LMAX * ENC_THREADS];
    __shared__ TT s_rk[ENC_LMAX * ENC_THREADS];
    const uint64_t c = (uint64_t)blockIdx.x * ENC_THREADS + threadIdx.x;
    if (c >= n_chunks) return;
    const uint64_t s0 = off[c];
    const uint64_t e0 = (c + 1 < n_chunks) ? off[c + 1] : n;
    uint32_t L = (uint32_t)min(e0 - s0, (uint64_t)0xFFFFFFFFu);
    if (L == 0) {
        outlen[c] = 0;
        return;
    }
    if (L > ENC_LMAX) {
        outlen[c] = 0;
        long_list[atomicAdd(n_long, 1ull)] = c;
        return;
    }
    TT *tok = s_tok + threadIdx.x;  // element i at tok[i * ENC_THREADS]
    TT *rk = s_rk + threadIdx.x;
    for (uint32_t i = 0; i < L; i++) tok[i * ENC_THREADS] = (TT)bytes[s0 + i];
    L = encode_lane<TT>(tok, rk, L, keys, vals, mask, merge_ids);
    for (uint32_t i = 0; i < L; i++) tmp[s0 + i] = tok[i * ENC_THREADS];
    outlen[c] = L;
}

// ---------------------------------------------------------------------------
// Encode with a chunk cache.  _encode_chunk is a pure function of the chunk's bytes, and under a
// GPT-style split a text is a few hundred thousand distinct chunks repeated a hundred million
// times: each DISTINCT short chunk is encoded once, by one of its occurrences (its "owner"), and
// every other occurrence copies the owner's tokens.  Exact whatever the input: two chunks are the
// same only if their BYTES are -- a chunk of up to 7 bytes is its own 64-bit table key, a longer one
// is keyed by a hash and compared byte for byte with the slot's owner -- and a chunk the table has
// no room for is simply encoded on its own.
//   k_enc_pass1  every chunk finds (or claims, compare-and-swap) its slot of an open-addressing
//                table.  Up to 7 bytes: the thread whose claim succeeds is the owner, and the wave
//                encodes its owners one after the other (encode_wave).  8..32 bytes: packed into a
//                per-workgroup list, hashed and probed by the workgroup's first waves; the slot's
//                owner is the lowest chunk index that hashed there (atomicMin), settled when the
//                launch ends
//   k_enc_pass2  over those lists: the owner encodes; every other occurrence compares its bytes with
//                the owner's (a different chunk behind the same hash: encoded on its own)
//   k_enc_place_chained  per chunk, the token count of its slot; the scan over the counts (a chained
//                scan over tiles) and the copy of the tokens, the first four of which sit in the slot
//                itself -- one pass.  (k_enc_lens + k_scan_top + k_enc_place: the same in three
//                launches, option enc_chain = 0.)
// Hot words: a slot is read before it is written -- a plain load first (a key never changes once it is
// there, so a cached copy that shows it is as good as memory), a relaxed agent-scope load past the
// per-CU L1 only when the slot looks empty -- so a word that occurs five million times costs a handful
// of atomics, not five million on one address.
constexpr uint32_t ENC_NOSLOT = 0xFFFFFFFFu;
constexpr uint32_t ENC_PROBES = 64;  // slots tried before a chunk goes uncached
constexpr uint32_t ENC_KEYBYTES = 7;  // chunks up to this length are their own key
struct __attribute__((aligned(32))) EncEntry {
    unsigned long long key;  // 0 = empty | (0x80 | len) << 56 | the chunk's bytes | 0x40 << 56 | 56 bits of hash
    uint32_t rep;            // the owner's chunk index
    uint32_t ntok;           // its token count ...
    uint32_t tok[4];         // ... and first four tokens (the rest: staging area, at the owner's byte offset)
};

// the chunk's bytes as four little-endian 64-bit words (zero beyond len), from aligned loads
__device__ __forceinline__ void chunk_words(const uint8_t *__restrict__ bytes, uint64_t s0, uint32_t len,
                                            unsigned long long (&w)[4]) {
    const unsigned long long *p = reinterpret_cast<const unsigned long long *>(bytes + (s0 & ~7ull));
    const uint32_t sh = (uint32_t)(s0 & 7u) * 8u;
    const uint32_t nw = (len + (uint32_t)(s0 & 7u) + 7u) / 8u;  // aligned words the chunk touches (1..5)
    unsigned long long a[5];
#pragma unroll
    for (int i = 0; i < 5; i++) a[i] = (uint32_t)i < nw ? p[i] : 0ull;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        unsigned long long v = a[i] >> sh;
        if (sh) v |= a[i + 1] << (64u - sh);
        const int rem = (int)len - 8 * i;  // bytes of this word that belong to the chunk
        w[i] = rem >= 8 ? v : (rem <= 0 ? 0ull : (v & ((1ull << (8 * rem)) - 1ull)));
    }
}
__device__ __forceinline__ unsigned long long chunk_hash(const unsigned long long (&w)[4], uint32_t len) {
    unsigned long long h = 0x9E3779B97F4A7C15ull ^ len;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        h ^= w[i];
        h *= 0xFF51AFD7ED558CCDull;
        h ^= h >> 32;
    }
    h *= 0xC4CEB9FE1A85EC53ull;
    h ^= h >> 29;
    return h;
}

typedef unsigned long long __attribute__((address_space(1))) enc_gu64;
typedef uint32_t __attribute__((address_space(1))) enc_gu32;

// ---------------------------------------------------------------------------
// One chunk by ONE WAVE: lane i holds token i (i < L <= 64), ranks of the adjacent pairs are looked up
// by all lanes at once, and a merge step is one wave-wide minimum, one lane shift and the two new
// look-ups side by side -- (merges + 1) dependent table accesses instead of the (L + 2 * merges) a
// single lane needs (encode_lane).  Owners are few (one per DISTINCT chunk), so a wave spends its
// lanes on one of them at a time instead of waiting for one lane.  Returns the new length; tok is
// updated (lanes >= L: undefined).
__device__ __forceinline__ uint32_t encode_wave(uint32_t &tok, uint32_t L, const unsigned long long *__restrict__ keys,
                                                const uint32_t *__restrict__ vals, uint32_t mask,
                                                const int32_t *__restrict__ merge_ids) {
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    const uint32_t lane = (uint32_t)lane_id();
    uint32_t nx = lane_next(tok, 0u);
    uint32_t rk = (lane + 1 < L) ? rank_lookup(keys, vals, mask, tok, nx) : NONE;
    while (L >= 2) {
        const uint32_t m = wave_umin_dpp(rk);
        if (m == NONE) break;  // nothing else can be merged
        const uint32_t bi = (uint32_t)__ffsll((long long)__ballot(rk == m)) - 1u;  // leftmost occurrence of the lowest rank
        const uint32_t z = merge_ids ? (uint32_t)merge_ids[m] : 256u + m;
        const uint32_t tn = lane_next(tok, 0u), rn = lane_next(rk, NONE);
        if (lane == bi) tok = z;
        if (lane > bi) {
            tok = tn;
            rk = rn;
        }
        L--;
        nx = lane_next(tok, 0u);
        if (lane + 1 >= L) rk = NONE;
        else if (lane + 1 == bi || lane == bi) rk = rank_lookup(keys, vals, mask, tok, nx);
    }
    return L;
}
// ... and its result: into the owner's slot (token count, first four tokens; the rest: staging area, at the
// owner's byte offset); slot == ENC_NOSLOT: everything into the staging area
__device__ __forceinline__ void enc_store_wave(uint32_t tok, uint32_t L, EncEntry *__restrict__ tab, uint32_t slot,
                                               uint32_t *__restrict__ tmp, uint64_t s0, uint32_t *__restrict__ outlen,
                                               uint64_t c) {
    const uint32_t lane = (uint32_t)lane_id();
    if (slot != ENC_NOSLOT) {
        if (lane < 4) tab[slot].tok[lane] = lane < L ? tok : 0u;
        if (lane == 0) {
            tab[slot].rep = (uint32_t)c;
            tab[slot].ntok = L;
        }
    }
    if ((slot == ENC_NOSLOT || L > 4) && lane < L) tmp[s0 + lane] = tok;
    if (lane == 0) outlen[c] = L;
}
__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int l) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

// the first min(len, 8) bytes of a chunk as one little-endian word (zero beyond len), from aligned loads
__device__ __forceinline__ unsigned long long chunk_word0(const uint8_t *__restrict__ bytes, uint64_t s0, uint32_t len) {
    const unsigned long long *p = reinterpret_cast<const unsigned long long *>(bytes + (s0 & ~7ull));
    const uint32_t sh = (uint32_t)(s0 & 7u) * 8u;
    unsigned long long v = p[0] >> sh;
    if (sh + 8u * len > 64u) v |= p[1] << (64u - sh);  // (sh != 0 here)
    return len >= 8 ? v : (v & ((1ull << (8 * len)) - 1ull));
}
// home slot of a key: 32-bit multiplies only (a 64-bit multiply is four of them on this part, and pass 1
// is bound by instruction issue, not by memory)
__device__ __forceinline__ uint32_t key_home(unsigned long long key, uint32_t tmask) {
    uint32_t h = (uint32_t)key * 0x9E3779B1u + (uint32_t)(key >> 32) * 0x85EBCA77u;
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    h ^= h >> 13;
    return h & tmask;
}
// find the key's slot or claim an empty one; ENC_NOSLOT after ENC_PROBES occupied slots
__device__ __forceinline__ uint32_t enc_probe(EncEntry *__restrict__ tab, uint32_t tmask, unsigned long long key, bool &claimed) {
    uint32_t h = key_home(key, tmask);
    claimed = false;
    for (uint32_t probe = 0; probe < ENC_PROBES; probe++) {
        // A key is written once and never changes: a cached copy that shows it is as good as memory.
        // Only an empty-looking slot is read again past the caches (another CU's insert never
        // refreshes this CU's L1), and claimed if it is still empty.
        unsigned long long cur = tab[h].key;  // (plain: may come from this CU's L1)
        if (cur == 0) cur = __hip_atomic_load((enc_gu64 *)&tab[h].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == 0) {
            cur = atomicCAS(&tab[h].key, 0ull, key);
            claimed = cur == 0;
        }
        if (claimed || cur == key) return h;
        h = (h + 1) & tmask;
    }
    return ENC_NOSLOT;
}

// Pass 1: the chunks that are their own key (up to ENC_KEYBYTES bytes: five in six under a GPT-style split)
// -- one aligned word or two, a 32-bit hash, a probe; the thread whose claim succeeds is the owner, and
// the owners of a wave are encoded by the whole wave, one after the other.  Every other short chunk (8..32
// bytes; or all of them when the test option cuts the hash) only goes on the workgroup's list for the
// workgroup's list -- lane numbers, packed -- and the workgroup's first wave(s) then hash them and find their
// slots: full waves over a sixth of the chunks instead of one live lane in six carrying the 64-bit arithmetic
// through every wave (pass 1 is bound by instruction issue).  Pass 2 works off the same list.
__global__ void __launch_bounds__(ENC_THREADS)
k_enc_pass1(const uint8_t *__restrict__ bytes, const uint64_t *__restrict__ off, uint64_t n_chunks, uint64_t n,
            EncEntry *__restrict__ tab, uint32_t tmask, uint32_t *__restrict__ slot_of, unsigned long long hash_keep,
            const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ vals, uint32_t mask,
            const int32_t *__restrict__ merge_ids, uint32_t *__restrict__ tmp, uint32_t *__restrict__ outlen,
            unsigned long long *__restrict__ long_list, unsigned long long *__restrict__ n_long,
            uint8_t *__restrict__ mid_list, uint32_t *__restrict__ mid_cnt) {
    __shared__ uint32_t s_wcnt[ENC_THREADS / 64];
    __shared__ uint8_t s_mid[ENC_THREADS];
    const uint64_t c = (uint64_t)blockIdx.x * ENC_THREADS + threadIdx.x;
    const int lane = lane_id();
    uint64_t s0 = 0;
    uint32_t L = 0, slot = ENC_NOSLOT;
    unsigned long long w0 = 0;
    bool own = false, mid = false;
    if (c < n_chunks) {
        s0 = off[c];
        const uint64_t e0 = (c + 1 < n_chunks) ? off[c + 1] : n;
        L = (uint32_t)min(e0 - s0, (uint64_t)0xFFFFFFFFu);
        if (L == 0 || L > ENC_LMAX) {  // (empty: nothing to encode; long: the stream-wide path)
            slot_of[c] = ENC_NOSLOT;
            outlen[c] = 0;
            if (L) long_list[atomicAdd(n_long, 1ull)] = c;
        } else if (L <= ENC_KEYBYTES && hash_keep == ~0ull) {
            w0 = chunk_word0(bytes, s0, L);
            bool claimed;
            slot = enc_probe(tab, tmask, w0 | ((unsigned long long)(0x80u | L) << 56), claimed);
            slot_of[c] = slot;
            own = slot == ENC_NOSLOT || claimed;  // (otherwise another occurrence owns the slot)
        } else {
            mid = true;
        }
    }
    // the owners among this wave's chunks, one after the other
    for (unsigned long long ob = __ballot(own); ob; ob &= ob - 1) {
        const int ol = __ffsll((long long)ob) - 1;
        const unsigned long long ww = readlane_u64(w0, ol);
        const uint32_t Lo = (uint32_t)__builtin_amdgcn_readlane((int)L, ol);
        const uint32_t oslot = (uint32_t)__builtin_amdgcn_readlane((int)slot, ol);
        const uint64_t oc = (uint64_t)blockIdx.x * ENC_THREADS + (threadIdx.x & ~63u) + (uint32_t)ol;
        const uint64_t os0 = readlane_u64(s0, ol);
        uint32_t tok = (uint32_t)((ww >> (8 * (lane & 7))) & 0xFFu);
        const uint32_t Ln = encode_wave(tok, Lo, keys, vals, mask, merge_ids);
        enc_store_wave(tok, Ln, tab, oslot, tmp, os0, outlen, oc);
    }
    // the hashed chunks of this workgroup, packed: its first waves find (or claim) their slots -- the slot's
    // owner is the lowest chunk index that hashed there (atomicMin), settled when the launch ends -- and the
    // list goes to pass 2
    const unsigned long long mb = __ballot(mid);
    if (lane == 0) s_wcnt[wave_id()] = (uint32_t)__popcll(mb);
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (int v = 0; v < ENC_THREADS / 64; v++) {
        if (v < wave_id()) base += s_wcnt[v];
        total += s_wcnt[v];
    }
    if (mid) s_mid[base + (uint32_t)__popcll(mb & ((1ull << lane) - 1ull))] = (uint8_t)threadIdx.x;
    if (threadIdx.x == 0) mid_cnt[blockIdx.x] = total;
    __syncthreads();
    if (threadIdx.x >= total) return;
    const uint32_t ml = s_mid[threadIdx.x];
    mid_list[(uint64_t)blockIdx.x * ENC_THREADS + threadIdx.x] = (uint8_t)ml;
    const uint64_t cm = (uint64_t)blockIdx.x * ENC_THREADS + ml;
    const uint64_t sm = off[cm];
    const uint64_t em = (cm + 1 < n_chunks) ? off[cm + 1] : n;
    const uint32_t Lm = (uint32_t)(em - sm);
    unsigned long long w[4];
    chunk_words(bytes, sm, Lm, w);
    // (hash_keep: all ones -- or, in tests, a few bits only: thousands of different chunks collide, so that
    // the byte comparison of pass 2 has to tell them apart)
    const unsigned long long key = (chunk_hash(w, Lm) & hash_keep & 0x00FFFFFFFFFFFFFFull) | (0x40ull << 56);
    bool claimed;
    const uint32_t mslot = enc_probe(tab, tmask, key, claimed);
    slot_of[cm] = mslot;
    if (mslot != ENC_NOSLOT) {
        const uint32_t cur = __hip_atomic_load((enc_gu32 *)&tab[mslot].rep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)cm < cur) atomicMin(&tab[mslot].rep, (uint32_t)cm);
    }
}

// Pass 2, over pass 1's lists: the slot's owner encodes; every other occurrence compares its bytes with the
// owner's (a different chunk behind the same hash: encoded on its own, uncached).
__global__ void __launch_bounds__(ENC_THREADS)
k_enc_pass2(const uint8_t *__restrict__ bytes, const uint64_t *__restrict__ off, uint64_t n_chunks, uint64_t n,
            EncEntry *__restrict__ tab, uint32_t *__restrict__ slot_of,
            const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ vals, uint32_t mask,
            const int32_t *__restrict__ merge_ids, uint32_t *__restrict__ tmp, uint32_t *__restrict__ outlen,
            const uint8_t *__restrict__ mid_list, const uint32_t *__restrict__ mid_cnt) {
    const uint32_t nmid = mid_cnt[blockIdx.x];
    if ((threadIdx.x & ~63u) >= nmid) return;  // (whole waves without work leave)
    const int lane = lane_id();
    const bool act = threadIdx.x < nmid;
    uint64_t c = 0, s0 = 0;
    uint32_t L = 0, slot = ENC_NOSLOT;
    unsigned long long w[4] = {0, 0, 0, 0};
    bool own = false;
    if (act) {
        c = (uint64_t)blockIdx.x * ENC_THREADS + mid_list[(uint64_t)blockIdx.x * ENC_THREADS + threadIdx.x];
        s0 = off[c];
        const uint64_t e0 = (c + 1 < n_chunks) ? off[c + 1] : n;
        L = (uint32_t)(e0 - s0);
        chunk_words(bytes, s0, L, w);
        slot = slot_of[c];
        own = true;
        if (slot != ENC_NOSLOT) {
            const uint32_t r = tab[slot].rep;
            if (r != (uint32_t)c) {
                // same slot, same hash: the same bytes, unless the hash collided -- look
                const uint64_t rs = off[r];
                const uint64_t re = ((uint64_t)r + 1 < n_chunks) ? off[r + 1] : n;
                bool same = (re - rs) == (uint64_t)L;
                if (same) {
                    unsigned long long v[4];
                    chunk_words(bytes, rs, L, v);
                    same = (v[0] == w[0]) & (v[1] == w[1]) & (v[2] == w[2]) & (v[3] == w[3]);
                }
                if (same) {
                    own = false;  // the owner's tokens are mine
                } else {
                    slot = ENC_NOSLOT;  // a different chunk behind the same hash: on its own
                    slot_of[c] = ENC_NOSLOT;
                }
            }
        }
    }
    for (unsigned long long ob = __ballot(own); ob; ob &= ob - 1) {
        const int ol = __ffsll((long long)ob) - 1;
        const unsigned long long w0 = readlane_u64(w[0], ol), w1 = readlane_u64(w[1], ol), w2 = readlane_u64(w[2], ol),
                                 w3 = readlane_u64(w[3], ol);
        const uint32_t Lo = (uint32_t)__builtin_amdgcn_readlane((int)L, ol);
        const uint32_t oslot = (uint32_t)__builtin_amdgcn_readlane((int)slot, ol);
        const uint64_t oc = readlane_u64(c, ol), os0 = readlane_u64(s0, ol);
        const unsigned long long wsel = (lane & 16) ? ((lane & 8) ? w3 : w2) : ((lane & 8) ? w1 : w0);  // (lanes 0..31)
        uint32_t tok = (uint32_t)((wsel >> (8 * (lane & 7))) & 0xFFu);
        const uint32_t Ln = encode_wave(tok, Lo, keys, vals, mask, merge_ids);
        enc_store_wave(tok, Ln, tab, oslot, tmp, os0, outlen, oc);
    }
}

__global__ void __launch_bounds__(256) k_enc_tab_init(EncEntry *__restrict__ tab, uint64_t nslots) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nslots) tab[i].rep = 0xFFFFFFFFu;  // (the rest of the table was zeroed)
}

// Output offsets and placement with the cache, two launches around k_scan_top:
//   k_enc_lens   every cached chunk takes its slot's token count (uncached, long and empty chunks have
//                theirs already); the sum of each tile of SCAN_TILE chunks
//   k_enc_place  exclusive scan inside the tile (thread t owns SCAN_TILE / 256 consecutive chunks), the
//                chunk's output offset, and its tokens: the first four sit in the slot itself
__global__ void __launch_bounds__(256)
k_enc_lens(const EncEntry *__restrict__ tab, const uint32_t *__restrict__ slot_of, uint64_t n_chunks,
           uint32_t *__restrict__ outlen, unsigned long long *__restrict__ bsum) {
    __shared__ unsigned long long s_red[4];
    constexpr int PER = SCAN_TILE / 256;
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + threadIdx.x;  // element j of this thread: base + 256 j
    uint32_t sl[PER], len[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) sl[j] = (base + 256u * j < n_chunks) ? slot_of[base + 256u * j] : ENC_NOSLOT;
#pragma unroll
    for (int j = 0; j < PER; j++)
        len[j] = (base + 256u * j < n_chunks) ? (sl[j] != ENC_NOSLOT ? tab[sl[j]].ntok : outlen[base + 256u * j]) : 0u;
    unsigned long long acc = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) {
        if (sl[j] != ENC_NOSLOT) outlen[base + 256u * j] = len[j];
        acc += len[j];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if (lane_id() == 0) s_red[wave_id()] = acc;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
}
// (thread t owns elements 16 t .. 16 t + 15 of the tile for the scan, element 256 j + t for every global
// access: the lengths go through LDS once each way, padded by one word per sixteen against bank conflicts)
__global__ void __launch_bounds__(256)
k_enc_place(const uint32_t *__restrict__ tmp, const uint64_t *__restrict__ off, const EncEntry *__restrict__ tab,
            const uint32_t *__restrict__ slot_of, const uint32_t *__restrict__ outlen,
            const unsigned long long *__restrict__ bsum, unsigned long long *__restrict__ out_off, uint64_t n_chunks,
            int32_t *__restrict__ out) {
    constexpr int PER = SCAN_TILE / 256;
    __shared__ uint32_t s_x[SCAN_TILE + SCAN_TILE / 16];
    __shared__ uint32_t s_w[4];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + threadIdx.x;
    uint32_t x[PER], sl[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const uint64_t c = base + 256u * j;
        x[j] = c < n_chunks ? outlen[c] : 0u;
        sl[j] = c < n_chunks ? slot_of[c] : ENC_NOSLOT;
        const uint32_t e = 256u * j + threadIdx.x;
        s_x[e + (e >> 4)] = x[j];
    }
    __syncthreads();
    uint32_t mine[PER], acc = 0;  // (a tile's tokens: below 2^32)
#pragma unroll
    for (int i = 0; i < PER; i++) {
        mine[i] = s_x[threadIdx.x * 17u + i];
        acc += mine[i];
    }
    uint32_t inc = wave_iscan_add(acc);
    if (lane_id() == 63) s_w[wave_id()] = inc;
    __syncthreads();
    uint32_t run = inc - acc;
    for (int v = 0; v < wave_id(); v++) run += s_w[v];
#pragma unroll
    for (int i = 0; i < PER; i++) {
        s_x[threadIdx.x * 17u + i] = run;
        run += mine[i];
    }
    __syncthreads();
    const unsigned long long tile0 = bsum[blockIdx.x];
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const uint64_t c = base + 256u * j;
        if (c >= n_chunks) break;
        const uint32_t e = 256u * j + threadIdx.x;
        const unsigned long long d0 = tile0 + s_x[e + (e >> 4)];
        out_off[c] = d0;
        const uint32_t L = x[j];
        if (L == 0) continue;
        if (sl[j] == ENC_NOSLOT) {
            const uint64_t s0 = off[c];
            for (uint32_t k = 0; k < L; k++) out[d0 + k] = (int32_t)tmp[s0 + k];
            continue;
        }
        const uint4 t4 = *reinterpret_cast<const uint4 *>(tab[sl[j]].tok);
        const uint32_t t[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
        for (uint32_t k = 0; k < 4; k++)
            if (k < L) out[d0 + k] = (int32_t)t[k];
        if (L > 4) {
            const uint64_t s0 = off[tab[sl[j]].rep];
            for (uint32_t k = 4; k < L; k++) out[d0 + k] = (int32_t)tmp[s0 + k];
        }
    }
}

// The two launches above and k_scan_top in ONE pass: a tile takes its number from a ticket (so that every
// tile before it has started), publishes its token count and looks back over its predecessors' descriptors
// for its offset (a chained scan: descriptor = status << 62 | value, 1 = the tile's own count, 2 = the
// count of everything up to and including the tile; agent-scope relaxed accesses, the data is the flag).
// Each chunk's slot is read once -- token count and first four tokens sit in the same 32 bytes -- and the
// per-chunk lengths never go through memory.  *total receives the batch's token count.
typedef unsigned long long __attribute__((address_space(1))) enc_desc_t;
constexpr int ENC_PLACE_TILE = 2048;  // chunks per tile of the chained pass (eight per thread: registers)
__global__ void __launch_bounds__(256)
k_enc_place_chained(const uint32_t *__restrict__ tmp, const uint64_t *__restrict__ off, const EncEntry *__restrict__ tab,
                    const uint32_t *__restrict__ slot_of, const uint32_t *__restrict__ outlen,
                    unsigned long long *__restrict__ desc, uint32_t *__restrict__ ticket,
                    unsigned long long *__restrict__ out_off, uint64_t n_chunks, int32_t *__restrict__ out,
                    unsigned long long *__restrict__ total) {
    constexpr int PER = ENC_PLACE_TILE / 256;
    __shared__ uint32_t s_x[ENC_PLACE_TILE + ENC_PLACE_TILE / PER];
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_tile;
    __shared__ unsigned long long s_excl;
    // chunks of more than ENC_PLACE_BIG tokens that sit in the staging area (long chunks; a BasicTokenizer text is ONE
    // chunk of millions of tokens): copied by the whole workgroup after the per-thread loop, not by the one thread
    constexpr uint32_t ENC_PLACE_BIG = 256, ENC_PLACE_NBIG = 32;
    __shared__ uint32_t s_nbig, s_bl[ENC_PLACE_NBIG];
    __shared__ unsigned long long s_bd[ENC_PLACE_NBIG], s_bs[ENC_PLACE_NBIG];
    if (threadIdx.x == 0) {
        s_tile = atomicAdd(ticket, 1u);
        s_nbig = 0;
    }
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint64_t base = (uint64_t)tile * ENC_PLACE_TILE + threadIdx.x;  // element j of this thread: base + 256 j
    uint32_t x[PER], sl[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) sl[j] = (base + 256u * j < n_chunks) ? slot_of[base + 256u * j] : ENC_NOSLOT;
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const uint64_t c = base + 256u * j;
        if (sl[j] != ENC_NOSLOT) {
            x[j] = tab[sl[j]].ntok;  // (the tokens are read further down, from the same 32 bytes: cached by then)
        } else {
            x[j] = c < n_chunks ? outlen[c] : 0u;
        }
        const uint32_t e2 = 256u * j + threadIdx.x;
        s_x[e2 + e2 / PER] = x[j];
    }
    __syncthreads();
    uint32_t mine[PER], acc = 0;  // (a tile's tokens: below 2^32)
#pragma unroll
    for (int i = 0; i < PER; i++) {
        mine[i] = s_x[threadIdx.x * (PER + 1u) + i];
        acc += mine[i];
    }
    const uint32_t inc = wave_iscan_add(acc);
    if (lane_id() == 63) s_w[wave_id()] = inc;
    __syncthreads();
    uint32_t run = inc - acc;
    for (int v = 0; v < wave_id(); v++) run += s_w[v];
#pragma unroll
    for (int i = 0; i < PER; i++) {
        s_x[threadIdx.x * (PER + 1u) + i] = run;
        run += mine[i];
    }
    const unsigned long long tile_total = (unsigned long long)s_w[0] + s_w[1] + s_w[2] + s_w[3];
    // ---- the tile's offset: look back ---------------------------------------------------------------
    if (wave_id() == 0) {
        const int lane = lane_id();
        constexpr unsigned long long VAL = (1ull << 62) - 1ull;
        if (lane == 0)
            __hip_atomic_store((enc_desc_t *)&desc[tile], ((tile == 0 ? 2ull : 1ull) << 62) | tile_total, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long excl = 0;
        long long idx = (long long)tile - 1;  // the nearest tile not yet accounted for
        while (idx >= 0) {
            const long long mine_i = idx - lane;
            unsigned long long d = 2ull << 62;  // (before the first tile: a prefix of nothing)
            if (mine_i >= 0) {
                do {
                    d = __hip_atomic_load((enc_desc_t *)&desc[mine_i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (d >> 62) break;
                    __builtin_amdgcn_s_sleep(1);
                } while (true);
            }
            const unsigned long long pre = __ballot((d >> 62) == 2ull);
            const int stop = pre ? __ffsll((long long)pre) - 1 : 64;  // the nearest tile that knows its prefix
            unsigned long long v = lane <= stop ? (d & VAL) : 0ull;
#pragma unroll
            for (int k = 32; k >= 1; k >>= 1) v += __shfl_xor(v, k);
            excl += v;
            if (pre) break;
            idx -= 64;
        }
        if (lane == 0) {
            if (tile != 0)
                __hip_atomic_store((enc_desc_t *)&desc[tile], (2ull << 62) | (excl + tile_total), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            s_excl = excl;
            if ((uint64_t)(tile + 1) * ENC_PLACE_TILE >= n_chunks) *total = excl + tile_total;
        }
    }
    __syncthreads();
    const unsigned long long tile0 = s_excl;
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const uint64_t c = base + 256u * j;
        if (c >= n_chunks) break;
        const uint32_t e2 = 256u * j + threadIdx.x;
        const unsigned long long d0 = tile0 + s_x[e2 + e2 / PER];
        out_off[c] = d0;
        const uint32_t L = x[j];
        if (L == 0) continue;
        if (sl[j] == ENC_NOSLOT) {
            const uint64_t s0 = off[c];
            if (L > ENC_PLACE_BIG) {
                const uint32_t b = atomicAdd(&s_nbig, 1u);
                if (b < ENC_PLACE_NBIG) {
                    s_bl[b] = L;
                    s_bd[b] = d0;
                    s_bs[b] = s0;
                    continue;
                }
            }
            for (uint32_t k = 0; k < L; k++) out[d0 + k] = (int32_t)tmp[s0 + k];
            continue;
        }
        const uint4 t4 = *reinterpret_cast<const uint4 *>(tab[sl[j]].tok);
        const uint32_t t[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
        for (uint32_t k = 0; k < 4; k++)
            if (k < L) out[d0 + k] = (int32_t)t[k];
        if (L > 4) {
            const uint64_t s0 = off[tab[sl[j]].rep];
            for (uint32_t k = 4; k < L; k++) out[d0 + k] = (int32_t)tmp[s0 + k];
        }
    }
    __syncthreads();
    const uint32_t nbig = min(s_nbig, ENC_PLACE_NBIG);
    for (uint32_t b = 0; b < nbig; b++) {
        const uint32_t L = s_bl[b];
        const unsigned long long d0 = s_bd[b], s0 = s_bs[b];
        for (uint32_t k = threadIdx.x; k < L; k += 256) out[d0 + k] = (int32_t)tmp[s0 + k];
    }
}

// long chunks: lowest rank present anywhere in the (flagged) stream
__global__ void __launch_bounds__(256)
k_min_rank(const uint32_t *__restrict__ ids, const DevState *__restrict__ st, int par,
           const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ vals,
           uint32_t mask, uint32_t *__restrict__ out_min) {
    const uint64_t n = st->n[par];
    const uint64_t total = (uint64_t)gridDim.x * blockDim.x;
    uint32_t best = 0xFFFFFFFFu;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p + 1 < n; p += total) {
        const uint32_t w1 = ids[p + 1];
        if (w1 & FLAG) continue;
        best = min(best, rank_lookup(keys, vals, mask, ids[p] & IDMASK, w1));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, d));
    if (lane_id() == 0 && best != 0xFFFFFFFFu) atomicMin(out_min, best);
}

// gather the bytes of the queued long chunks into one flagged id stream
__global__ void __launch_bounds__(256)
k_long_gather(const uint8_t *__restrict__ bytes, const unsigned long long *__restrict__ src_off,
              const unsigned long long *__restrict__ dst_off, uint64_t n_long,
              uint32_t *__restrict__ ids) {
    const uint64_t k = blockIdx.x;
    if (k >= n_long) return;
    const unsigned long long s0 = src_off[k], d0 = dst_off[k], len = dst_off[k + 1] - d0;
    for (unsigned long long i = threadIdx.x; i < len; i += 256)
        ids[d0 + i] = (uint32_t)bytes[s0 + i] | (i == 0 ? FLAG : 0u);
}

// ... and put their encoded tokens back into the per-chunk staging area
__global__ void __launch_bounds__(256)
k_long_scatter(const uint32_t *__restrict__ ids, const unsigned long long *__restrict__ starts,
               const unsigned long long *__restrict__ chunk_id, const unsigned long long *__restrict__ src_off,
               uint64_t n_long, uint32_t *__restrict__ tmp, uint32_t *__restrict__ outlen) {
    const uint64_t k = blockIdx.x;
    if (k >= n_long) return;
    const unsigned long long p0 = starts[k], len = starts[k + 1] - p0, d0 = src_off[k];
    for (unsigned long long i = threadIdx.x; i < len; i += 256) tmp[d0 + i] = ids[p0 + i] & IDMASK;
    if (threadIdx.x == 0) outlen[chunk_id[k]] = (uint32_t)len;
}

// ---------------------------------------------------------------------------
// Long chunks on the device (lengths ENC_LMAX + 1 .. ENC_LONG_TOP bytes: URLs, code, whitespace runs): one workgroup
// per chunk -- ONE WAVE up to ENC_LONG_MID bytes, four up to ENC_LONG_MAX, sixteen beyond --, the chunk's tokens and the ranks of its adjacent
// pairs in LDS.  A round is what one iteration of the reference's loop does (regex.py:92-109 == basic.py:57-74): the
// lowest rank present (a minimum over the rank array), every occurrence of that pair merged left to right (an a == a
// run is walked by the thread that owns its start: the greedy pairing of base.py:25-41), the survivors compacted into
// the other buffer (ballots per 64 positions, one scan over the ballots' counts), and only the ranks next to a new
// token looked up again -- no stream-wide pass, no host round trip.  Thread t owns positions t, t + NT, ... .  The
// launch works through the list pass 1 left (its length is read on the device); CAP = the most bytes this
// instantiation takes, chunks of `lo` bytes or fewer belong to a smaller one, longer ones go on to `huge_list` (the
// stream-wide rounds of api_encode.hip: a BasicTokenizer text is one such chunk).
// (three instantiations: one wave up to ENC_LONG_MID bytes, four up to ENC_LONG_MAX, sixteen up to ENC_LONG_TOP -- 9 KiB: what
// two token and two rank buffers of a chunk leave of a CU's 160 KB of LDS)
constexpr uint32_t ENC_LONG_MID = 512, ENC_LONG_MAX = 4096, ENC_LONG_TOP = 9216;
template <uint32_t CAP, uint32_t NT>
__global__ void __launch_bounds__(NT)
k_enc_long(const uint8_t *__restrict__ bytes, const uint64_t *__restrict__ off, uint64_t n_chunks, uint64_t n,
           const unsigned long long *__restrict__ long_list, const unsigned long long *__restrict__ n_long,
           const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ vals, uint32_t mask,
           const int32_t *__restrict__ merge_ids, uint32_t *__restrict__ tmp, uint32_t *__restrict__ outlen,
           unsigned long long *__restrict__ huge_list, unsigned long long *__restrict__ n_huge, uint32_t lo) {
    constexpr uint32_t NW = NT / 64, NB = CAP / 64;  // waves; groups of 64 positions (one ballot each)
    constexpr uint32_t NG = (NB + 63) / 64;  // groups whose counts one lane of wave 0 scans
    static_assert(CAP % NT == 0 && CAP % 64 == 0, "whole groups of 64 positions");
    static_assert((2 * 2 * 4 + 1) * CAP + 4 * NB + 4 * NW + 64 <= 160 * 1024, "the chunk's buffers in LDS");
    __shared__ uint32_t s_tok[2][CAP], s_rk[2][CAP];
    __shared__ uint8_t s_fl[CAP];
    __shared__ uint32_t s_cnt[NB], s_red[NW], s_len;
    constexpr uint32_t NONE = 0xFFFFFFFFu, REDO = 0xFFFFFFFEu;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    auto sync = [] {
        if (NT == 64) {  // (one wave: its LDS operations complete in issue order)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            __syncthreads();
        }
    };
    const unsigned long long nl = *n_long;
    for (unsigned long long k = blockIdx.x; k < nl; k += gridDim.x) {
        const unsigned long long c = long_list[k];
        const uint64_t s0 = off[c];
        const uint64_t e0 = (c + 1 < n_chunks) ? off[c + 1] : n;
        const uint64_t Lb = e0 - s0;
        if (Lb <= lo || Lb > CAP) {  // (uniform) not this instantiation's
            if (CAP == ENC_LONG_TOP && Lb > CAP && tid == 0) huge_list[atomicAdd(n_huge, 1ull)] = c;
            continue;
        }
        uint32_t L = (uint32_t)Lb, cur = 0;
        for (uint32_t i = tid; i < L; i += NT) s_tok[0][i] = bytes[s0 + i];
        sync();
        for (uint32_t i = tid; i < L; i += NT)
            s_rk[0][i] = (i + 1 < L) ? rank_lookup(keys, vals, mask, s_tok[0][i], s_tok[0][i + 1]) : NONE;
        sync();
        for (;;) {
            const uint32_t *tok = s_tok[cur], *rk = s_rk[cur];
            uint32_t *ntok = s_tok[cur ^ 1], *nrk = s_rk[cur ^ 1];
            uint32_t m = NONE;
            for (uint32_t i = tid; i + 1 < L; i += NT) m = min(m, rk[i]);
            m = wave_umin_dpp(m);
            if (NW > 1) {
                if (lane == 0) s_red[wave] = m;
                sync();
#pragma unroll
                for (uint32_t w = 0; w < NW; w++) m = min(m, s_red[w]);
            }
            if (m >= REDO) break;  // (uniform) nothing else can be merged
            const uint32_t Z = merge_ids ? (uint32_t)merge_ids[m] : 256u + m;
            // bit 0: the pair starts here; bit 1: ... and is merged (a run of overlapping occurrences -- a == a only --
            // takes every other one, from its left end)
            for (uint32_t i = tid; i < L; i += NT) s_fl[i] = (i + 1 < L && rk[i] == m) ? 1 : 0;
            sync();
            for (uint32_t i = tid; i + 1 < L; i += NT) {
                if ((s_fl[i] & 1) && (i == 0 || !(s_fl[i - 1] & 1))) {
                    bool take = true;
                    for (uint32_t j = i; j + 1 < L && (s_fl[j] & 1); j++) {
                        if (take) s_fl[j] |= 2;
                        take = !take;
                    }
                }
            }
            sync();
            // survivors: everything but the second token of a merged pair.  One ballot per group of 64 positions, the
            // counts scanned by wave 0, then every survivor knows its new place
            const uint32_t ngroups = (L + 63) / 64;
            for (uint32_t g = wave; g < ngroups; g += NW) {
                const uint32_t i = g * 64 + lane;
                const bool kp = i < L && !(i > 0 && (s_fl[i - 1] & 2));
                const unsigned long long bal = __ballot(kp);
                if (lane == 0) s_cnt[g] = (uint32_t)__popcll(bal);
            }
            sync();
            if (wave == 0) {  // (lane l scans groups NG l .. NG l + NG - 1)
                uint32_t loc[NG], v = 0;
#pragma unroll
                for (uint32_t g = 0; g < NG; g++) {
                    const uint32_t i = lane * NG + g;
                    loc[g] = i < ngroups ? s_cnt[i] : 0u;
                    v += loc[g];
                }
                const uint32_t inc = wave_iscan_add(v);
                uint32_t run = inc - v;
#pragma unroll
                for (uint32_t g = 0; g < NG; g++) {
                    const uint32_t i = lane * NG + g;
                    if (i < ngroups) s_cnt[i] = run;
                    run += loc[g];
                }
                if (lane == 63) s_len = inc;
            }
            sync();
            for (uint32_t g = wave; g < ngroups; g += NW) {
                const uint32_t i = g * 64 + lane;
                const bool in = i < L;
                const bool kp = in && !(i > 0 && (s_fl[i - 1] & 2));
                const unsigned long long bal = __ballot(kp);
                if (kp) {
                    const uint32_t p = s_cnt[g] + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                    const bool site = (s_fl[i] & 2) != 0;
                    // the rank to the right of a new token, and of the token before one, is looked up below
                    const bool next_site = (i + 1 < L) && (s_fl[i + 1] & 2);
                    ntok[p] = site ? Z : tok[i];
                    nrk[p] = (site || next_site) ? REDO : rk[i];
                }
            }
            const uint32_t Ln = s_len;
            sync();
            L = Ln;
            cur ^= 1;
            for (uint32_t i = tid; i < L; i += NT)
                if (nrk[i] == REDO) nrk[i] = (i + 1 < L) ? rank_lookup(keys, vals, mask, ntok[i], ntok[i + 1]) : NONE;
            sync();
        }
        for (uint32_t i = tid; i < L; i += NT) tmp[s0 + i] = s_tok[cur][i];
        if (tid == 0) outlen[c] = L;
        sync();  // (the next chunk overwrites the buffers)
    }
}
// the byte ranges of the chunks the stream-wide rounds take, in one pass (instead of a copy per chunk)
__global__ void __launch_bounds__(256)
k_long_ranges(const uint64_t *__restrict__ off, uint64_t n_chunks, uint64_t n, const unsigned long long *__restrict__ list,
              uint64_t n_list, unsigned long long *__restrict__ range) {
    const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n_list) return;
    const unsigned long long c = list[k];
    range[2 * k] = off[c];
    range[2 * k + 1] = (c + 1 < n_chunks) ? off[c + 1] : n;
}

// exclusive scan of the per-chunk output lengths (u32 -> u64), three small kernels
__global__ void __launch_bounds__(256)
k_scan_blocksum(const uint32_t *__restrict__ v, uint64_t n, unsigned long long *__restrict__ bsum) {
    __shared__ unsigned long long s_red[4];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
    unsigned long long acc = 0;
    for (uint32_t i = threadIdx.x; i < SCAN_TILE; i += 256)
        if (base + i < n) acc += v[base + i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if (lane_id() == 0) s_red[wave_id()] = acc;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
}
__global__ void __launch_bounds__(1024)
k_scan_top(unsigned long long *__restrict__ bsum, uint64_t nb, unsigned long long *__restrict__ total) {
    __shared__ unsigned long long s_part[1024];
    const uint64_t R = (nb + 1023) / 1024;
    const uint64_t b0 = min((uint64_t)threadIdx.x * R, nb), b1 = min(b0 + R, nb);
    unsigned long long acc = 0;
    for (uint64_t b = b0; b < b1; b++) acc += bsum[b];
    s_part[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < 1024; i++) {
            const unsigned long long t = s_part[i];
            s_part[i] = run;
            run += t;
        }
        *total = run;
    }
    __syncthreads();
    unsigned long long run = s_part[threadIdx.x];
    for (uint64_t b = b0; b < b1; b++) {
        const unsigned long long t = bsum[b];
        bsum[b] = run;
        run += t;
    }
}
__global__ void __launch_bounds__(256)
k_scan_apply(const uint32_t *__restrict__ v, uint64_t n, const unsigned long long *__restrict__ bsum,
             unsigned long long *__restrict__ out) {
    // thread t owns SCAN_TILE/256 consecutive values of its block
    __shared__ unsigned long long s_w[4];
    constexpr int PER = SCAN_TILE / 256;
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * PER;
    uint32_t x[PER];
    unsigned long long acc = 0;
#pragma unroll
    for (int i = 0; i < PER; i++) {
        x[i] = (base + i < n) ? v[base + i] : 0u;
        acc += x[i];
    }
    unsigned long long inc = acc;
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) s_w[wave_id()] = inc;
    __syncthreads();
    unsigned long long run = bsum[blockIdx.x] + inc - acc;
    for (int w = 0; w < wave_id(); w++) run += s_w[w];
#pragma unroll
    for (int i = 0; i < PER; i++) {
        if (base + i < n) out[base + i] = run;
        run += x[i];
    }
}

// final placement: chunk c's tokens go to out[out_off[c] ...]
__global__ void __launch_bounds__(256)
k_encode_place(const uint32_t *__restrict__ tmp, const uint64_t *__restrict__ off,
               const uint32_t *__restrict__ outlen, const unsigned long long *__restrict__ out_off,
               uint64_t n_chunks, int32_t *__restrict__ out) {
    const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n_chunks) return;
    const uint32_t L = outlen[c];
    const uint64_t s0 = off[c];
    const unsigned long long d0 = out_off[c];
    for (uint32_t i = 0; i < L; i++) out[d0 + i] = (int32_t)tmp[s0 + i];
}

}  // namespace bpe
