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
//                table.  Up to 7 bytes: the thread whose claim succeeds is the owner and encodes on
//                the spot (one chunk per lane, as k_encode_short).  8..32 bytes: the slot's owner is
//                the lowest chunk index that hashed there (atomicMin), settled when the launch ends
//   k_enc_pass2  8..32 bytes: the owner encodes; every other occurrence compares its bytes with
//                the owner's (a different chunk behind the same hash: encoded on its own)
//   k_enc_count  per chunk, the token count of its slot; then the usual scan, and
//   k_enc_place  copies the tokens: the first four sit in the slot itself
// Hot words: a slot is read before it is written (relaxed agent-scope loads: L2-served, past the
// per-CU L1 that another CU's insert never refreshes), so a word that occurs five million times
// costs a handful of atomics, not five million on one address.
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
__device__ __forceinline__ uint32_t key_home(unsigned long long key, uint32_t tmask) {
    return (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 36) & tmask;
}

typedef unsigned long long __attribute__((address_space(1))) enc_gu64;
typedef uint32_t __attribute__((address_space(1))) enc_gu32;

// the owner's result goes into its slot (and, beyond four tokens, into the staging area)
template <typename TT>
__device__ __forceinline__ void enc_store(const TT *tok, uint32_t L, EncEntry *e, uint32_t *__restrict__ tmp, uint64_t s0,
                                          uint32_t *__restrict__ outlen, uint64_t c) {
    if (e) {
        uint32_t t4[4];
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) t4[i] = i < L ? (uint32_t)tok[i * ENC_THREADS] : 0u;
        e->ntok = L;
        *reinterpret_cast<uint4 *>(e->tok) = make_uint4(t4[0], t4[1], t4[2], t4[3]);
    }
    if (!e || L > 4)
        for (uint32_t i = 0; i < L; i++) tmp[s0 + i] = tok[i * ENC_THREADS];
    outlen[c] = L;
}

template <typename TT>
__global__ void __launch_bounds__(ENC_THREADS)
k_enc_pass1(const uint8_t *__restrict__ bytes, const uint64_t *__restrict__ off, uint64_t n_chunks, uint64_t n,
            EncEntry *__restrict__ tab, uint32_t tmask, uint32_t *__restrict__ slot_of, unsigned long long hash_keep,
            const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ vals, uint32_t mask,
            const int32_t *__restrict__ merge_ids, uint32_t *__restrict__ tmp, uint32_t *__restrict__ outlen,
            unsigned long long *__restrict__ long_list, unsigned long long *__restrict__ n_long) {
    __shared__ TT s_tok[ENC_LMAX * ENC_THREADS];
    __shared__ TT s_rk[ENC_LMAX * ENC_THREADS];
    const uint64_t c = (uint64_t)blockIdx.x * ENC_THREADS + threadIdx.x;
    if (c >= n_chunks) return;
    const uint64_t s0 = off[c];
    const uint64_t e0 = (c + 1 < n_chunks) ? off[c + 1] : n;
    uint32_t L = (uint32_t)min(e0 - s0, (uint64_t)0xFFFFFFFFu);
    if (L == 0 || L > ENC_LMAX) {  // (empty: nothing to encode; long: the stream-wide path)
        slot_of[c] = ENC_NOSLOT;
        outlen[c] = 0;
        if (L) long_list[atomicAdd(n_long, 1ull)] = c;
        return;
    }
    unsigned long long w[4];
    chunk_words(bytes, s0, L, w);
    const bool exact = L <= ENC_KEYBYTES && hash_keep == ~0ull;
    // (hash_keep: all ones -- or, in tests, a few bits only: every chunk goes the hashed way and
    // thousands of different ones collide, so that the byte comparison of pass 2 has to tell them apart)
    const unsigned long long key = exact ? (w[0] | ((unsigned long long)(0x80u | L) << 56))
                                         : ((chunk_hash(w, L) & hash_keep & 0x00FFFFFFFFFFFFFFull) | (0x40ull << 56));
    uint32_t h = key_home(key, tmask);
    uint32_t slot = ENC_NOSLOT;
    bool claimed = false;
    for (uint32_t probe = 0; probe < ENC_PROBES; probe++) {
        unsigned long long cur = __hip_atomic_load((enc_gu64 *)&tab[h].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == 0) {
            cur = atomicCAS(&tab[h].key, 0ull, key);
            claimed = cur == 0;
        }
        if (claimed || cur == key) {
            slot = h;
            break;
        }
        h = (h + 1) & tmask;
    }
    slot_of[c] = slot;
    if (!exact) {
        if (slot != ENC_NOSLOT) {  // the owner is settled when this launch ends: pass 2 goes on
            const uint32_t cur = __hip_atomic_load((enc_gu32 *)&tab[slot].rep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((uint32_t)c < cur) atomicMin(&tab[slot].rep, (uint32_t)c);
        }
        return;
    }
    if (slot != ENC_NOSLOT && !claimed) return;  // another occurrence owns the slot
    TT *tok = s_tok + threadIdx.x;  // element i at tok[i * ENC_THREADS]
    TT *rk = s_rk + threadIdx.x;
    for (uint32_t i = 0; i < L; i++) tok[i * ENC_THREADS] = (TT)((w[0] >> (8 * i)) & 0xFFu);
    L = encode_lane<TT>(tok, rk, L, keys, vals, mask, merge_ids);
    if (slot != ENC_NOSLOT) tab[slot].rep = (uint32_t)c;
    enc_store<TT>(tok, L, slot != ENC_NOSLOT ? &tab[slot] : nullptr, tmp, s0, outlen, c);
}

template <typename TT>
__global__ void __launch_bounds__(ENC_THREADS)
k_enc_pass2(const uint8_t *__restrict__ bytes, const uint64_t *__restrict__ off, uint64_t n_chunks, uint64_t n,
            EncEntry *__restrict__ tab, uint32_t *__restrict__ slot_of, unsigned long long hash_keep,
            const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ vals, uint32_t mask,
            const int32_t *__restrict__ merge_ids, uint32_t *__restrict__ tmp, uint32_t *__restrict__ outlen) {
    __shared__ TT s_tok[ENC_LMAX * ENC_THREADS];
    __shared__ TT s_rk[ENC_LMAX * ENC_THREADS];
    const uint64_t c = (uint64_t)blockIdx.x * ENC_THREADS + threadIdx.x;
    if (c >= n_chunks) return;
    const uint64_t s0 = off[c];
    const uint64_t e0 = (c + 1 < n_chunks) ? off[c + 1] : n;
    uint32_t L = (uint32_t)min(e0 - s0, (uint64_t)0xFFFFFFFFu);
    if (L == 0 || L > ENC_LMAX) return;
    if (L <= ENC_KEYBYTES && hash_keep == ~0ull) return;  // settled in pass 1
    unsigned long long w[4];
    chunk_words(bytes, s0, L, w);
    uint32_t slot = slot_of[c];
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
            if (same) return;  // the owner's tokens are mine
            slot = ENC_NOSLOT;   // a different chunk behind the same hash: on its own
            slot_of[c] = ENC_NOSLOT;
        }
    }
    TT *tok = s_tok + threadIdx.x;
    TT *rk = s_rk + threadIdx.x;
    for (uint32_t i = 0; i < L; i++) tok[i * ENC_THREADS] = (TT)((w[i >> 3] >> (8 * (i & 7))) & 0xFFu);
    L = encode_lane<TT>(tok, rk, L, keys, vals, mask, merge_ids);
    enc_store<TT>(tok, L, slot != ENC_NOSLOT ? &tab[slot] : nullptr, tmp, s0, outlen, c);
}

__global__ void __launch_bounds__(256) k_enc_tab_init(EncEntry *__restrict__ tab, uint64_t nslots) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nslots) tab[i].rep = 0xFFFFFFFFu;  // (the rest of the table was zeroed)
}

// every cached chunk takes its slot's count (uncached, long and empty chunks have theirs already)
__global__ void __launch_bounds__(256)
k_enc_count(const EncEntry *__restrict__ tab, const uint32_t *__restrict__ slot_of, uint64_t n_chunks,
            uint32_t *__restrict__ outlen) {
    const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n_chunks) return;
    const uint32_t slot = slot_of[c];
    if (slot != ENC_NOSLOT) outlen[c] = tab[slot].ntok;
}

// final placement with the cache
__global__ void __launch_bounds__(256)
k_enc_place(const uint32_t *__restrict__ tmp, const uint64_t *__restrict__ off, const EncEntry *__restrict__ tab,
            const uint32_t *__restrict__ slot_of, const uint32_t *__restrict__ outlen,
            const unsigned long long *__restrict__ out_off, uint64_t n_chunks, int32_t *__restrict__ out) {
    const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n_chunks) return;
    const uint32_t L = outlen[c];
    if (L == 0) return;
    const unsigned long long d0 = out_off[c];
    const uint32_t slot = slot_of[c];
    if (slot == ENC_NOSLOT) {
        const uint64_t s0 = off[c];
        for (uint32_t i = 0; i < L; i++) out[d0 + i] = (int32_t)tmp[s0 + i];
        return;
    }
    const uint4 t4 = *reinterpret_cast<const uint4 *>(tab[slot].tok);
    const uint32_t t[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
    for (uint32_t i = 0; i < 4; i++)
        if (i < L) out[d0 + i] = (int32_t)t[i];
    if (L > 4) {
        const uint64_t s0 = off[tab[slot].rep];
        for (uint32_t i = 4; i < L; i++) out[d0 + i] = (int32_t)tmp[s0 + i];
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
