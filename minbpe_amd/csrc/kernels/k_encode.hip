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
// times: each DISTINCT short chunk is encoded once, by its first occurrence (its "owner"), and
// every other occurrence copies the owner's tokens.  Exact whatever the input: equality of chunks is
// decided by comparing their bytes, never by the hash alone, and a chunk the table has no room for
// is simply its own owner.
//   k_enc_hash   chunk -> slot of an open-addressing table keyed by a 64-bit hash of the bytes; the
//                slot's representative = the lowest chunk index that hashed there (atomicMin)
//   k_enc_owner  chunk == its slot's representative, or its bytes differ from the representative's
//                (a hash collision), or it has no slot: owner, encoded here (one chunk per lane);
//                else rep[c] = the representative
//   k_enc_count  every other chunk takes its owner's token count; then the usual scan, and
//   k_enc_place  copies the owner's tokens
// Hot words: a slot is read before it is written (relaxed agent-scope loads: L2-served, past the
// per-CU L1 that another CU's insert never refreshes), so a word that occurs five million times
// costs a handful of atomics, not five million on one address.
constexpr uint32_t ENC_NOSLOT = 0xFFFFFFFFu;
constexpr uint32_t ENC_PROBES = 64;  // slots tried before a chunk goes uncached

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
    return h ? h : 1ull;  // 0 = empty slot
}

typedef unsigned long long __attribute__((address_space(1))) enc_gu64;
typedef uint32_t __attribute__((address_space(1))) enc_gu32;

__global__ void __launch_bounds__(256)
k_enc_hash(const uint8_t *__restrict__ bytes, const uint64_t *__restrict__ off, uint64_t n_chunks, uint64_t n,
           unsigned long long *__restrict__ tab_hash, uint32_t *__restrict__ tab_rep, uint32_t tmask,
           uint32_t *__restrict__ slot_of, unsigned long long hash_keep) {
    const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n_chunks) return;
    const uint64_t s0 = off[c];
    const uint64_t e0 = (c + 1 < n_chunks) ? off[c + 1] : n;
    const uint64_t len = e0 - s0;
    if (len == 0 || len > ENC_LMAX) {  // (empty: nothing to encode; long: the stream-wide path)
        slot_of[c] = ENC_NOSLOT;
        return;
    }
    unsigned long long w[4];
    chunk_words(bytes, s0, (uint32_t)len, w);
    // (hash_keep: all ones -- or, in tests, a few bits only, so that different chunks collide and the
    // byte comparison in k_enc_owner has to tell them apart)
    const unsigned long long hsh = (chunk_hash(w, (uint32_t)len) & hash_keep) | 1ull;
    uint32_t h = (uint32_t)(hsh >> 20) & tmask;
    uint32_t slot = ENC_NOSLOT;
    for (uint32_t probe = 0; probe < ENC_PROBES; probe++) {
        unsigned long long cur = __hip_atomic_load((enc_gu64 *)&tab_hash[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == 0) cur = atomicCAS(&tab_hash[h], 0ull, hsh);
        if (cur == 0 || cur == hsh) {
            slot = h;
            break;
        }
        h = (h + 1) & tmask;
    }
    slot_of[c] = slot;
    if (slot != ENC_NOSLOT) {
        const uint32_t cur = __hip_atomic_load((enc_gu32 *)&tab_rep[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)c < cur) atomicMin(&tab_rep[slot], (uint32_t)c);
    }
}

template <typename TT>
__global__ void __launch_bounds__(ENC_THREADS)
k_enc_owner(const uint8_t *__restrict__ bytes, const uint64_t *__restrict__ off, uint64_t n_chunks, uint64_t n,
            const uint32_t *__restrict__ tab_rep, uint32_t *__restrict__ slot_rep,
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
    const uint32_t slot = slot_rep[c];  // (k_enc_hash left the slot here; the owner's index replaces it)
    slot_rep[c] = (uint32_t)c;
    if (L == 0) {
        outlen[c] = 0;
        return;
    }
    if (L > ENC_LMAX) {
        outlen[c] = 0;
        long_list[atomicAdd(n_long, 1ull)] = c;
        return;
    }
    unsigned long long w[4];
    chunk_words(bytes, s0, L, w);
    if (slot != ENC_NOSLOT) {
        const uint32_t r = tab_rep[slot];
        if (r != (uint32_t)c) {
            // same slot, same 64-bit hash: the same bytes, unless the hash collided -- look
            const uint64_t rs = off[r];
            const uint64_t re = ((uint64_t)r + 1 < n_chunks) ? off[r + 1] : n;
            bool same = (re - rs) == (uint64_t)L;
            if (same) {
                unsigned long long v[4];
                chunk_words(bytes, rs, L, v);
                same = (v[0] == w[0]) & (v[1] == w[1]) & (v[2] == w[2]) & (v[3] == w[3]);
            }
            if (same) {
                slot_rep[c] = r;  // the count follows in k_enc_count
                return;
            }
        }
    }
    TT *tok = s_tok + threadIdx.x;  // element i at tok[i * ENC_THREADS]
    TT *rk = s_rk + threadIdx.x;
    for (uint32_t i = 0; i < L; i++) tok[i * ENC_THREADS] = (TT)((w[i >> 3] >> (8 * (i & 7))) & 0xFFu);
    L = encode_lane<TT>(tok, rk, L, keys, vals, mask, merge_ids);
    for (uint32_t i = 0; i < L; i++) tmp[s0 + i] = tok[i * ENC_THREADS];
    outlen[c] = L;
}

// every chunk that is not its own owner takes the owner's count (owners are final: written by the
// launch before, or by the long-chunk path)
__global__ void __launch_bounds__(256)
k_enc_count(const uint32_t *__restrict__ rep, uint64_t n_chunks, uint32_t *__restrict__ outlen) {
    const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n_chunks) return;
    const uint32_t r = rep[c];
    if (r != (uint32_t)c) outlen[c] = outlen[r];
}

// final placement with the cache: chunk c's tokens are its owner's
__global__ void __launch_bounds__(256)
k_enc_place(const uint32_t *__restrict__ tmp, const uint64_t *__restrict__ off, const uint32_t *__restrict__ rep,
            const uint32_t *__restrict__ outlen, const unsigned long long *__restrict__ out_off, uint64_t n_chunks,
            int32_t *__restrict__ out) {
    const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n_chunks) return;
    const uint32_t L = outlen[c];
    const uint64_t s0 = off[rep[c]];
    const unsigned long long d0 = out_off[c];
    for (uint32_t i = 0; i < L; i++) out[d0 + i] = (int32_t)tmp[s0 + i];
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
