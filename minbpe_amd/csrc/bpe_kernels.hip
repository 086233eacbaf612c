// bpe_kernels.hip -- gfx950 (MI355X / CDNA4) kernels for the BPE hot path.
//
// Written for 64-wide wavefronts, 256 CUs in 8 XCDs, 160 KiB LDS per CU and an
// HBM3E-bound workload: every kernel here is integer / byte work whose roofline
// is HBM bandwidth, so the rules that matter are coalesced 16 B-per-lane
// accesses, LDS pre-aggregation of atomics, and keeping the id stream resident.
// No MFMA anywhere.
//
// Data layout
//   id stream : uint32 words, ping-pong buffers.  bits 0..30 = token id,
//               bit 31 = "this token starts a chunk" (regex.py:44: pairs never
//               span chunks).  A pair (p, p+1) exists iff word[p+1] has bit 31
//               clear, so "next word == b" is a complete validity test.
//   pair table: dense row-major uint32 matrix count[a][b], stride = vcap.
//               288 GB of HBM makes the dense form affordable (vocab 32000 ->
//               4.1 GB) and turns per-iteration table maintenance into four
//               dense vectors (delta mode) -- see k_apply_delta.
//   row maxima: rowmax[a] = max_b count[a][b]; argmax scans V values, not V^2.
//
// Reference semantics restated here:
//   F1 get_stats counts every adjacent pair        -> k_pair_count_*
//   F2 merge is greedy left-to-right               -> mbit() / run-parity scan
//   F3 argmax ties go to the earliest first occurrence -> k_argmax + k_tiebreak
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bpe_device.h"

namespace bpe {

// ---------------------------------------------------------------------------
// small wave / block helpers (wave = 64 lanes, hard-coded: gfx950 only)

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// what one pair inside the chunk of word w adds to a count: 2^(weight exponent)
__device__ __forceinline__ uint32_t word_weight(uint32_t w) { return 1u << ((w >> WSHIFT) & 31u); }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d));
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d);
    return v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = min(v, __shfl_xor(v, d));
    return v;
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        unsigned long long o = __shfl_xor(v, d);
        v = o < v ? o : v;
    }
    return v;
}

// DPP cross-lane moves (VALU speed; __shfl_* lower to ds_bpermute through the LDS crossbar).
// ctrl: 0x110+n row_shr:n | 0x130 wave_shl:1 | 0x138 wave_shr:1 | 0x142 row_bcast:15 | 0x143 row_bcast:31
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_mov(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xF, false);
}
// inclusive scans over the 64 lanes: 4 steps inside each row of 16, then two row broadcasts
__device__ __forceinline__ int wave_iscan_max(int v) {  // identity -1 (values are >= -1)
    v = max(v, dpp_mov<0x111>(-1, v));
    v = max(v, dpp_mov<0x112>(-1, v));
    v = max(v, dpp_mov<0x114>(-1, v));
    v = max(v, dpp_mov<0x118>(-1, v));
    v = max(v, dpp_mov<0x142, 0xA>(-1, v));
    v = max(v, dpp_mov<0x143, 0xC>(-1, v));
    return v;
}
__device__ __forceinline__ uint32_t wave_iscan_add(uint32_t x) {
    int v = (int)x;
    v += dpp_mov<0x111>(0, v);
    v += dpp_mov<0x112>(0, v);
    v += dpp_mov<0x114>(0, v);
    v += dpp_mov<0x118>(0, v);
    v += dpp_mov<0x142, 0xA>(0, v);
    v += dpp_mov<0x143, 0xC>(0, v);
    return (uint32_t)v;
}
__device__ __forceinline__ uint32_t lane_next(uint32_t x, uint32_t fill) {  // value of lane+1 (lane 63: fill)
    return (uint32_t)dpp_mov<0x130>((int)fill, (int)x);
}
__device__ __forceinline__ uint32_t lane_first(uint32_t x) { return (uint32_t)__builtin_amdgcn_readlane((int)x, 0); }
__device__ __forceinline__ uint32_t lane_last(uint32_t x) { return (uint32_t)__builtin_amdgcn_readlane((int)x, 63); }

// ---------------------------------------------------------------------------
// K0: list(text_bytes)  (basic.py:25-26, regex.py:44)
// 16 B read -> 64 B written per lane; HBM-bound, 5 B of traffic per id.

__global__ void __launch_bounds__(256)
k_widen(const uint8_t *__restrict__ src, uint32_t *__restrict__ dst, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * 16;
    for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; i < n; i += stride) {
        if (i + 16 <= n) {
            const uint4 v = *reinterpret_cast<const uint4 *>(src + i);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint4 o;
                o.x = w[k] & 0xffu;
                o.y = (w[k] >> 8) & 0xffu;
                o.z = (w[k] >> 16) & 0xffu;
                o.w = w[k] >> 24;
                *reinterpret_cast<uint4 *>(dst + i + 4 * k) = o;
            }
        } else {
            for (uint64_t j = i; j < n; j++) dst[j] = src[j];
        }
    }
}

// int32 ids from the host (module-level get_stats/merge drop-ins): strip sign.
__global__ void k_mark_starts(uint32_t *ids, const uint64_t *__restrict__ off, uint64_t n_chunks,
                              uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks; c += stride) {
        const uint64_t o = off[c];
        if (o < n) atomicOr(&ids[o], FLAG);  // duplicate offsets (empty chunks) are idempotent
    }
}

// weighted chunks (N1): every word of chunk c carries the chunk's weight exponent
__global__ void k_mark_weights(uint32_t *ids, const uint64_t *__restrict__ off, const uint8_t *__restrict__ wexp,
                               uint64_t n_chunks, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks; c += stride) {
        const uint32_t e = (uint32_t)(wexp[c] & 31u) << WSHIFT;
        if (!e) continue;
        const uint64_t p0 = off[c], p1 = (c + 1 < n_chunks) ? off[c + 1] : n;
        for (uint64_t p = p0; p < p1 && p < n; p++) ids[p] |= e;
    }
}

// ---------------------------------------------------------------------------
// K1: get_stats  (base.py:13-22; shared dict over chunks regex.py:51-54)
//
// k_pair_count_simple: one global atomic per position. Used when the first
// position of every pair is wanted too (bpe_get_stats: dict insertion order).
template <bool FIRST>
__global__ void __launch_bounds__(256)
k_pair_count_simple(const uint32_t *__restrict__ ids, const DevState *__restrict__ st, int par,
                    uint32_t *__restrict__ mat, uint32_t stride, uint32_t *__restrict__ first) {
    const uint64_t n = st->n[par];
    const uint64_t total = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g * 4 < n; g += total) {
        const uint64_t p = g * 4;
        const uint4 v = *reinterpret_cast<const uint4 *>(ids + p);  // buffers are tile-padded
        uint32_t x[5] = {v.x, v.y, v.z, v.w, ids[p + 4]};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (p + k + 1 < n && !(x[k + 1] & FLAG)) {
                const size_t idx = (size_t)(x[k] & IDMASK) * stride + (x[k + 1] & IDMASK);
                atomicAdd(&mat[idx], word_weight(x[k]));
                if (FIRST) atomicMin(&first[idx], (uint32_t)(p + k));
            }
        }
    }
}

// k_pair_count_lds: the general histogram.  Each workgroup (1024 threads, one
// per CU) owns a contiguous span of the stream and a 16 Ki-slot LDS cache
// {key = table index, count} in all 128 KiB of dynamic LDS.  A position costs
// one ds_read + one ds_add on a hit; a key that finds its slot and the next
// three taken goes straight to an L2 atomic.  The cache is flushed once per
// workgroup (one global atomic per resident key), which turns the Zipf-hot
// pairs -- the ones that would serialise at one L2 channel -- into ~#CU atomics.
__device__ __forceinline__ void cache_add(uint32_t *keys, uint32_t *vals, uint32_t *__restrict__ g,
                                          uint32_t idx, uint32_t v) {
    uint32_t h = (idx * 0x9E3779B1u) >> (32 - PC_BITS);
#pragma unroll
    for (int probe = 0; probe < 4; probe++) {
        uint32_t k = __atomic_load_n(&keys[h], __ATOMIC_RELAXED);
        if (k == EMPTY_KEY) {
            const uint32_t old = atomicCAS(&keys[h], EMPTY_KEY, idx);
            k = (old == EMPTY_KEY) ? idx : old;
        }
        if (k == idx) {
            atomicAdd(&vals[h], v);
            return;
        }
        h = (h + 1) & ((1u << PC_BITS) - 1);
    }
    atomicAdd(&g[idx], v);
}

__global__ void __launch_bounds__(PC_THREADS)
k_pair_count_lds(const uint32_t *__restrict__ ids, const DevState *__restrict__ st, int par,
                 uint32_t *__restrict__ mat, uint32_t stride) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_pc[];
    uint32_t *s_keys = s_pc, *s_vals = s_pc + (1 << PC_BITS);
    for (int i = threadIdx.x; i < (1 << PC_BITS); i += PC_THREADS) {
        s_keys[i] = EMPTY_KEY;
        s_vals[i] = 0;
    }
    __syncthreads();
    const uint64_t n = st->n[par];
    // contiguous span per workgroup, rounded to whole 4-id groups per thread
    const uint64_t groups = (n + 3) / 4;
    const uint64_t per_wg = (groups + gridDim.x - 1) / gridDim.x;
    const uint64_t g0 = per_wg * blockIdx.x;
    const uint64_t g1 = min(g0 + per_wg, groups);
    for (uint64_t g = g0 + threadIdx.x; g < g1; g += PC_THREADS) {
        const uint64_t p = g * 4;
        const uint4 v = *reinterpret_cast<const uint4 *>(ids + p);
        const uint32_t x[5] = {v.x, v.y, v.z, v.w, ids[p + 4]};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (p + k + 1 < n && !(x[k + 1] & FLAG))
                cache_add(s_keys, s_vals, mat, (x[k] & IDMASK) * stride + (x[k + 1] & IDMASK), word_weight(x[k]));
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (1 << PC_BITS); i += PC_THREADS) {
        const uint32_t c = s_vals[i];
        if (c) atomicAdd(&mat[s_keys[i]], c);
    }
}

// k_pair_count_bytes: get_stats of a freshly widened stream (every id < 256) --
// the one full histogram a delta-mode train() runs.  The whole 256 x 256 table
// fits in LDS as 16-bit counters (two per word, 128 KiB): one ds_add per
// position, no keys, no probing.  A workgroup flushes every PCB_ROUND positions
// (< 65536), so a half-word can never carry into its neighbour.
__global__ void __launch_bounds__(PC_THREADS)
k_pair_count_bytes(const uint32_t *__restrict__ ids, const DevState *__restrict__ st, int par,
                   uint32_t *__restrict__ mat, uint32_t stride) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_pc[];  // 32768 words
    for (int i = threadIdx.x; i < 32768; i += PC_THREADS) s_pc[i] = 0;
    __syncthreads();
    const uint64_t n = st->n[par];
    const uint64_t groups = (n + 3) / 4;
    const uint64_t per_wg = (groups + gridDim.x - 1) / gridDim.x;
    const uint64_t g0 = per_wg * blockIdx.x;
    const uint64_t g1 = min(g0 + per_wg, groups);
    constexpr uint64_t ROUND_GROUPS = PCB_ROUND / 4;
    constexpr int U = 4;  // 4-id groups per thread in flight: the kernel is latency-bound otherwise
    for (uint64_t r0 = g0; r0 < g1; r0 += ROUND_GROUPS) {
        const uint64_t r1 = min(r0 + ROUND_GROUPS, g1);
        for (uint64_t gb = r0; gb < r1; gb += (uint64_t)U * PC_THREADS) {
            uint4 v[U];
            uint32_t nx[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint64_t g = gb + (uint64_t)u * PC_THREADS + threadIdx.x;
                if (g < r1) {
                    v[u] = *reinterpret_cast<const uint4 *>(ids + g * 4);
                    nx[u] = ids[g * 4 + 4];
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint64_t g = gb + (uint64_t)u * PC_THREADS + threadIdx.x;
                if (g >= r1) continue;
                const uint64_t p = g * 4;
                const uint32_t x[5] = {v[u].x, v[u].y, v[u].z, v[u].w, nx[u]};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    // branch-free: a position that is not a pair adds 0 (one ds_add per
                    // position either way; no exec-mask juggling around every atomic)
                    const bool ok = (p + k + 1 < n) & !(x[k + 1] & FLAG);
                    const uint32_t idx = ((x[k] & 0xFFu) << 8) | (x[k + 1] & 0xFFu);
                    atomicAdd(&s_pc[idx >> 1], ok ? ((idx & 1u) ? 0x10000u : 1u) : 0u);
                }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 32768; i += PC_THREADS) {
            const uint32_t w = s_pc[i];
            if (w) {
                s_pc[i] = 0;
                const uint32_t i0 = 2u * (uint32_t)i;  // idx = a<<8 | b
                if (w & 0xFFFFu) atomicAdd(&mat[(size_t)(i0 >> 8) * stride + (i0 & 0xFFu)], w & 0xFFFFu);
                if (w >> 16) atomicAdd(&mat[(size_t)((i0 + 1) >> 8) * stride + ((i0 + 1) & 0xFFu)], w >> 16);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// K2: pair = max(stats, key=stats.get)  (basic.py:35, regex.py:56)

// one workgroup per row: rowmax[x] = max_y count[x][y]
__global__ void __launch_bounds__(256)
k_rowmax_all(const uint32_t *__restrict__ mat, uint32_t stride, uint32_t vcur,
             uint32_t *__restrict__ rowmax) {
    __shared__ uint32_t s_red[4];
    const uint32_t x = blockIdx.x;
    const uint32_t *row = mat + (size_t)x * stride;
    uint32_t m = 0;
    const uint32_t v4 = vcur & ~3u;
    for (uint32_t y = threadIdx.x * 4; y < v4; y += 256 * 4) {
        const uint4 q = *reinterpret_cast<const uint4 *>(row + y);
        m = max(max(m, q.x), max(max(q.y, q.z), q.w));
    }
    for (uint32_t y = v4 + threadIdx.x; y < vcur; y += 256) m = max(m, row[y]);
    m = wave_max_u32(m);
    if (lane_id() == 0) s_red[wave_id()] = m;
    __syncthreads();
    if (threadIdx.x == 0) rowmax[x] = max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3]));
}

// ---------------------------------------------------------------------------
// stream views (contiguous or slotted), used by the tie-break scans

__device__ __forceinline__ bool slot_get(const SlotRef &r, uint64_t n, uint64_t p, uint32_t &w) {
    if (!r.meta) {
        if (p >= n) return false;
        w = r.b0[p];
        return true;
    }
    const uint64_t t = p / TILE;
    if (t >= r.T) return false;
    const uint32_t m = r.meta[t];
    if ((uint32_t)(p % TILE) >= (m & 0x7FFFFFFFu)) return false;
    w = ((m >> 31) ? r.b1 : r.b0)[p];
    return true;
}
// the word that follows position p in stream order
__device__ __forceinline__ bool slot_next(const SlotRef &r, uint64_t n, uint64_t p, uint32_t &w) {
    if (!r.meta) return slot_get(r, n, p + 1, w);
    uint64_t t = p / TILE;
    if ((uint32_t)(p % TILE) + 1 < (r.meta[t] & 0x7FFFFFFFu)) return slot_get(r, n, p + 1, w);
    for (t = t + 1; t < r.T; t++)
        if (r.meta[t] & 0x7FFFFFFFu) return slot_get(r, n, t * TILE, w);
    return false;
}
__device__ __forceinline__ uint64_t slot_space(const SlotRef &r, uint64_t n) {
    return r.meta ? r.T * (uint64_t)TILE : n;
}

// the pair test of the tie-break: is (a, w1) one of the pairs tied at the max?
__device__ __forceinline__ bool tie_hit(const int32_t *s_tied, uint32_t nt, uint32_t M,
                                        const uint32_t *__restrict__ mat, uint32_t stride,
                                        uint32_t a, uint32_t w1) {
    if (nt <= TIE_CAP) {
        bool hit = false;
        for (uint32_t t = 0; t < nt; t++)
            hit |= (s_tied[2 * t] == (int32_t)a) & (s_tied[2 * t + 1] == (int32_t)w1);
        return hit;
    }
    return mat[(size_t)a * stride + w1] == M;
}

// K2, single workgroup: global max over rowmax, gather every pair that attains
// it (the candidates of the reference's first-occurrence tie-break, F3), and --
// if there is a tie -- search the first TIE_WINDOW0 positions of the stream for
// the earliest tied pair.  Ties among frequent pairs always resolve there; the
// rest of the stream is k_tiebreak's job.
__device__ __forceinline__ void select_body(const uint32_t *__restrict__ rowmax,
                                            const uint32_t *__restrict__ mat, uint32_t stride,
                                            uint32_t vcur, DevState *st, const SlotRef &ref, int par,
                                            int dist) {
    __shared__ uint32_t s_red[16];
    __shared__ uint32_t s_M, s_nrows, s_nt, s_first;
    __shared__ uint32_t s_rows[ARGMAX_ROWS];
    __shared__ int32_t s_tied[2 * TIE_CAP];
    if (st->status) return;
    uint32_t m = 0;
    for (uint32_t x = threadIdx.x; x < vcur; x += 1024) m = max(m, rowmax[x]);
    m = wave_max_u32(m);
    if (lane_id() == 0) s_red[wave_id()] = m;
    if (threadIdx.x == 0) {
        s_nrows = 0;
        s_nt = 0;
        s_first = 0xFFFFFFFFu;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t M = 0;
        for (int i = 0; i < 16; i++) M = max(M, s_red[i]);
        s_M = M;
    }
    __syncthreads();
    const uint32_t M = s_M;
    if (M == 0) {  // stats is empty: max() raises ValueError in the reference (F6)
        if (threadIdx.x == 0) {
            st->status = ST_EMPTY;
            st->count = 0;
            st->found = 0;
        }
        return;
    }
    for (uint32_t x = threadIdx.x; x < vcur; x += 1024) {
        if (rowmax[x] == M) {
            const uint32_t s = atomicAdd(&s_nrows, 1u);
            if (s < ARGMAX_ROWS) s_rows[s] = x;
        }
    }
    __syncthreads();
    const uint32_t nrows = s_nrows;
    if (nrows <= ARGMAX_ROWS) {
        for (uint32_t r = 0; r < nrows; r++) {
            const uint32_t x = s_rows[r];
            const uint32_t *row = mat + (size_t)x * stride;
            for (uint32_t y = threadIdx.x; y < vcur; y += 1024) {
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
    const uint32_t nt = (nrows > ARGMAX_ROWS) ? (TIE_CAP + 1) : min(s_nt, (uint32_t)TIE_CAP + 1);
    if (threadIdx.x < 2 * min(nt, (uint32_t)TIE_CAP)) st->tied[threadIdx.x] = s_tied[threadIdx.x];
    if (nt > 1) {  // tie: first window, positions ascending per thread
        const uint64_t n = st->n[par];
        const uint32_t hi = (uint32_t)min((uint64_t)TIE_WINDOW0, slot_space(ref, n));
        if (ref.meta) {
            // slot by slot: one meta lookup per slot, coalesced reads inside it
            for (uint32_t u = 0; u < hi / TILE + 1 && (uint64_t)u < ref.T; u++) {
                if (__atomic_load_n(&s_first, __ATOMIC_RELAXED) != 0xFFFFFFFFu) break;  // earlier slot hit
                const uint32_t mu = ref.meta[u];
                const uint32_t len = mu & 0x7FFFFFFFu;
                const uint32_t *src = ((mu >> 31) ? ref.b1 : ref.b0) + (size_t)u * TILE;
                for (uint32_t q = threadIdx.x; q < len; q += 1024) {
                    uint32_t w1;
                    if (q + 1 < len) w1 = src[q + 1];
                    else if (!slot_next(ref, n, (uint64_t)u * TILE + q, w1)) continue;
                    if (w1 & FLAG) continue;
                    if (tie_hit(s_tied, nt, M, mat, stride, src[q] & IDMASK, w1 & IDMASK)) {
                        atomicMin(&s_first, u * TILE + q);
                        break;
                    }
                }
                __syncthreads();
            }
        } else
        for (uint32_t p = threadIdx.x; p < hi; p += 1024) {
            if (__atomic_load_n(&s_first, __ATOMIC_RELAXED) < p) break;  // an earlier hit exists
            uint32_t w0, w1;
            if (!slot_get(ref, n, p, w0) || !slot_next(ref, n, p, w1) || (w1 & FLAG)) continue;
            if (tie_hit(s_tied, nt, M, mat, stride, w0 & IDMASK, w1 & IDMASK)) {
                atomicMin(&s_first, p);
                break;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        st->count = M;
        st->ntied = nt;
        st->firstpos = NOPOS;
        if (nt == 1) {
            st->found = 1;
            st->a = s_tied[0];
            st->b = s_tied[1];
        } else if (s_first != 0xFFFFFFFFu && dist) {
            st->found = 0;  // sharded stream: only a candidate, the ranks compare positions
            st->firstpos = s_first;
        } else if (s_first != 0xFFFFFFFFu) {
            uint32_t w0 = 0, w1 = 0;
            slot_get(ref, st->n[par], s_first, w0);
            slot_next(ref, st->n[par], s_first, w1);
            st->found = 1;
            st->a = (int32_t)(w0 & IDMASK);
            st->b = (int32_t)(w1 & IDMASK);
        } else {
            st->found = 0;
        }
    }
}

// K2 kernel.  Block 0 decides (select_body); the other blocks wait for its
// decision (one flag, agent-scope release/acquire -- cdna_hip_programming.md
// G16) and, only if a tie is open, ALL blocks sweep the stream front to back
// for the earliest position holding a tied pair (each sweep step covers
// gridDim*1024 consecutive positions, so a block stops as soon as an earlier
// position has been reported).  One launch instead of two; block 0 never waits,
// so there is no circular dependency whatever the residency.
__global__ void __launch_bounds__(1024)
k_select(const uint32_t *__restrict__ rowmax, const uint32_t *__restrict__ mat, uint32_t stride,
         uint32_t vcur, DevState *st, SlotRef ref, int par, int dist, uint32_t epoch) {
    __shared__ int32_t s_tied[2 * TIE_CAP];
    __shared__ uint32_t s_go;
    if (blockIdx.x == 0) {
        select_body(rowmax, mat, stride, vcur, st, ref, par, dist);
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&st->sel_flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_go = (st->status == 0 && st->found == 0 && st->firstpos == NOPOS);
        }
    } else if (threadIdx.x == 0) {
        bool ok = false;
        for (uint32_t spins = 0; spins < LOOKBACK_SPINS; spins++) {
            if (__hip_atomic_load(&st->sel_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) {
                ok = true;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        s_go = (ok && st->status == 0 && st->found == 0 &&
                __atomic_load_n(&st->firstpos, __ATOMIC_RELAXED) == NOPOS);
    }
    __syncthreads();
    if (!s_go) return;
    const uint32_t nt = st->ntied;
    const uint32_t M = st->count;
    if (nt <= TIE_CAP && threadIdx.x < 2 * nt) s_tied[threadIdx.x] = st->tied[threadIdx.x];
    __syncthreads();
    const uint64_t n = st->n[par];
    const uint64_t space = slot_space(ref, n);
    // every block, block 0 included, sweeps: position order = (sweep step, block, thread)
    const uint64_t total = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t p = TIE_WINDOW0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < space; p += total) {
        if (__atomic_load_n(&st->firstpos, __ATOMIC_RELAXED) < p) break;
        uint32_t w0, w1;
        if (!slot_get(ref, n, p, w0) || !slot_next(ref, n, p, w1) || (w1 & FLAG)) continue;
        if (tie_hit(s_tied, nt, M, mat, stride, w0 & IDMASK, w1 & IDMASK)) {
            atomicMin(&st->firstpos, (unsigned long long)p);
            break;  // later positions of this thread cannot be earlier
        }
    }
}

// The pair to merge as every kernel after K2 sees it: decided by k_select, or
// the pair found at the earliest tied position by k_tiebreak.
__device__ __forceinline__ bool resolved_pair(const DevState *st, const uint32_t *__restrict__ ids,
                                              uint32_t &a, uint32_t &b) {
    if (st->found) {
        a = (uint32_t)st->a;
        b = (uint32_t)st->b;
        return true;
    }
    const unsigned long long p = st->firstpos;
    if (p == NOPOS) return false;
    a = ids[p] & IDMASK;
    b = ids[p + 1] & IDMASK;
    return true;
}

__device__ __forceinline__ bool resolved_pair(const DevState *st, const SlotRef &ref, uint64_t n,
                                              uint32_t &a, uint32_t &b) {
    if (st->found) {
        a = (uint32_t)st->a;
        b = (uint32_t)st->b;
        return true;
    }
    const unsigned long long p = st->firstpos;
    uint32_t w0, w1;
    if (p == NOPOS || !slot_get(ref, n, p, w0) || !slot_next(ref, n, p, w1)) return false;
    a = w0 & IDMASK;
    b = w1 & IDMASK;
    return true;
}

// single-step API (bpe_argmax): make the decision final in st
__global__ void k_finalize(SlotRef ref, int par, DevState *st) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (st->status == 0 && !st->found) {
        uint32_t a, b;
        if (!resolved_pair(st, ref, st->n[par], a, b)) {
            st->status = ST_INTERNAL;  // a tie was reported but no tied pair is in the stream
        } else {
            st->a = (int32_t)a;
            st->b = (int32_t)b;
            st->found = 1;
        }
    }
}

// host-chosen pair for the single-step bpe_merge()
__global__ void k_set_pair(DevState *st, int32_t a, int32_t b) {
    st->a = a;
    st->b = b;
    st->found = 1;
    st->status = 0;
    st->count = 0;
}

// ---------------------------------------------------------------------------
// K3: merge  (base.py:25-41, applied to every chunk regex.py:60)
//
// Greedy left-to-right replacement.  r[p] = 1 iff (word[p], word[p+1]) is the
// pair; a site starts at p iff m[p] = r[p] & !m[p-1].  With L_p = length of the
// run of ones of r ending at p, m[p] = r[p] & (L_p odd) -- for a != b runs of r
// have length 1 and m = r; for a == b this is exactly the reference's pairing
// inside a run "aaaa..." (F2).  L_p comes from a max-scan of "index of the last
// zero of r", so one code path serves both cases.  A run that reaches the tile
// start takes the carry s = m[tile_start-1] of the previous tile.
//
// Tile = 4 waves; each wave owns MJ stripes of 256 consecutive ids, lane l holds
// ids [4l, 4l+4) of each stripe: every global load is a full 1 KiB wave access.

struct Tile {
    uint32_t x[MJ][4];  // words
    uint32_t rb[MJ];    // r bits of my 4 elements per stripe
    int E[MJ];          // tile-relative index of the last zero of r before my group (-1: none)
    uint32_t tail[3];   // the three words after this wave's span (INVALID_WORD past n)
};

// fetch from a contiguous stream of n ids
__device__ __forceinline__ void tile_fetch(Tile &t, const uint32_t *__restrict__ ids, uint64_t n,
                                           uint64_t tile_base) {
    const int lane = lane_id(), wave = wave_id();
    const uint64_t wbase = tile_base + (uint64_t)wave * WAVE_SPAN;
#pragma unroll
    for (int j = 0; j < MJ; j++) {
        const uint64_t p0 = wbase + j * 256 + lane * 4;
        const uint4 v = *reinterpret_cast<const uint4 *>(ids + p0);
        t.x[j][0] = (p0 + 0 < n) ? v.x : INVALID_WORD;
        t.x[j][1] = (p0 + 1 < n) ? v.y : INVALID_WORD;
        t.x[j][2] = (p0 + 2 < n) ? v.z : INVALID_WORD;
        t.x[j][3] = (p0 + 3 < n) ? v.w : INVALID_WORD;
    }
    const uint64_t tailp = wbase + WAVE_SPAN;
#pragma unroll
    for (int i = 0; i < 3; i++) t.tail[i] = (tailp + i < n) ? ids[tailp + i] : INVALID_WORD;
}

// fetch slot `src` holding `len` owned ids, followed (in stream order) by the
// three words halo[0..2] that belong to later slots (INVALID_WORD at the end of
// the stream).  Positions >= len + 3 are INVALID_WORD.  Two steps, so that the
// slot's own loads are in flight while thread 0 looks the neighbours up.
struct SlotRaw {
    uint4 v[MJ];
    uint32_t tail[3];
};
__device__ __forceinline__ void slot_raw_load(SlotRaw &r, const uint32_t *__restrict__ src, int len) {
    const int lane = lane_id(), wrel = wave_id() * WAVE_SPAN;
#pragma unroll
    for (int j = 0; j < MJ; j++) {
        const int q0 = wrel + j * 256 + lane * 4;
        r.v[j] = make_uint4(0, 0, 0, 0);
        if (q0 < len) r.v[j] = *reinterpret_cast<const uint4 *>(src + q0);
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int q = wrel + WAVE_SPAN + i;
        r.tail[i] = (q < len) ? src[q] : 0u;
    }
}
__device__ __forceinline__ void tile_from_slot(Tile &t, const SlotRaw &r, int len, const uint32_t *halo) {
    const int lane = lane_id(), wrel = wave_id() * WAVE_SPAN;
#pragma unroll
    for (int j = 0; j < MJ; j++) {
        const int q0 = wrel + j * 256 + lane * 4;
        const uint32_t w[4] = {r.v[j].x, r.v[j].y, r.v[j].z, r.v[j].w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int q = q0 + k;
            t.x[j][k] = (q < len) ? w[k] : ((q < len + 3) ? halo[q - len] : INVALID_WORD);
        }
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int q = wrel + WAVE_SPAN + i;
        t.tail[i] = (q < len) ? r.tail[i] : ((q < len + 3) ? halo[q - len] : INVALID_WORD);
    }
}

// r bits of my elements: r[p] = 1 iff (word[p], word[p+1]) is the pair
__device__ __forceinline__ void tile_rbits(Tile &t, uint32_t a, uint32_t b) {
    uint32_t nx[MJ];
    const uint32_t tail = t.tail[0];
#pragma unroll
    for (int j = 0; j < MJ; j++) {
        const uint32_t up = (j < MJ - 1) ? lane_first(t.x[(j + 1) % MJ][0]) : tail;
        nx[j] = lane_next(t.x[j][0], up);
    }
#pragma unroll
    for (int j = 0; j < MJ; j++) {
        uint32_t rb = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t nxt = (k < 3) ? t.x[j][k + 1] : nx[j];
            rb |= (uint32_t)(((t.x[j][k] & IDMASK) == a) & ((nxt & NWMASK) == b)) << k;
        }
        t.rb[j] = rb;
    }
}
// exclusive max-scan of "index of the last zero of r" in (wave, stripe, lane) order:
// everything the m bits need (contains one __syncthreads)
__device__ __forceinline__ void tile_lzscan(Tile &t, int *s_wave) {
    const int lane = lane_id(), wave = wave_id();
    int lzg[MJ];
    const int gb0 = wave * WAVE_SPAN + lane * 4;
#pragma unroll
    for (int j = 0; j < MJ; j++) {
        const uint32_t z = (~t.rb[j]) & 0xFu;
        lzg[j] = z ? (gb0 + j * 256 + (31 - __clz((int)z))) : -1;
    }
    int carry = -1;
#pragma unroll
    for (int j = 0; j < MJ; j++) {
        const int v = wave_iscan_max(lzg[j]);
        const int ex = dpp_mov<0x138>(-1, v);  // wave_shr:1 -> exclusive
        t.E[j] = max(carry, ex);
        carry = max(carry, (int)lane_last((uint32_t)v));
    }
    if (lane == 0) s_wave[wave] = carry;
    __syncthreads();
    int win = -1;
    for (int w = 0; w < wave; w++) win = max(win, s_wave[w]);
#pragma unroll
    for (int j = 0; j < MJ; j++) t.E[j] = max(t.E[j], win);
}
__device__ __forceinline__ void tile_prepare(Tile &t, uint32_t a, uint32_t b, int *s_wave) {
    tile_rbits(t, a, b);
    tile_lzscan(t, s_wave);
}

__device__ __forceinline__ void tile_load(Tile &t, const uint32_t *__restrict__ ids, uint64_t n,
                                          uint64_t tile_base, uint32_t a, uint32_t b, int *s_wave) {
    tile_fetch(t, ids, n, tile_base);
    tile_prepare(t, a, b, s_wave);
}

// m bit of tile-relative position q given lz = index of the last zero at or
// before q's predecessor... see callers.  s = carry into the tile.
__device__ __forceinline__ uint32_t parity_bit(int q, int lz, uint32_t s) {
    return (uint32_t)((q - lz) & 1) ^ ((lz < 0) ? s : 0u);
}

// m bits (4) of my group in stripe j, and mprev = m of the element before it.
__device__ __forceinline__ uint32_t group_mbits(const Tile &t, int j, uint32_t s, uint32_t &mprev) {
    const int q0 = wave_id() * WAVE_SPAN + j * 256 + lane_id() * 4;
    int lz = t.E[j];
    // predecessor q0-1: r = 1 unless it is the last zero itself
    mprev = (q0 == 0) ? s : ((lz == q0 - 1) ? 0u : parity_bit(q0 - 1, lz, s));
    uint32_t mb = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if ((t.rb[j] >> k) & 1u) {
            mb |= parity_bit(q0 + k, lz, s) << k;
        } else {
            lz = q0 + k;
        }
    }
    return mb;
}

// Per-tile summary, as a function of the unknown carry s (packed in 64 bits):
//   M0      sites in the tile for s = 0
//   Podd    length of the all-ones prefix of r is odd   (M1 = M0 - Podd)
//   allones r is 1 on the whole tile                    (o1 = !o0, else o1 = o0)
//   o0      m[last] for s = 0 (carry into the next tile)
// Returned to every thread of the workgroup.
struct SummaryLds {
    uint32_t cnt[MT / 64];
    int fz[MT / 64];
    uint32_t o0;
    unsigned long long packed;
};
__device__ __forceinline__ uint64_t tile_summary(const Tile &t, int len, SummaryLds &L) {
    if (threadIdx.x == 0) L.o0 = 0;
    __syncthreads();
    uint32_t cnt = 0;
    int fz = 0x7fffffff;
    const int gb0 = wave_id() * WAVE_SPAN + lane_id() * 4;
#pragma unroll
    for (int j = 0; j < MJ; j++) {
        uint32_t mprev;
        const uint32_t mb = group_mbits(t, j, 0u, mprev);
        cnt += __popc(mb);
        const uint32_t z = (~t.rb[j]) & 0xFu;
        if (z) fz = min(fz, gb0 + j * 256 + (__ffs((int)z) - 1));
        const int q0 = gb0 + j * 256;
        if (len - 1 >= q0 && len - 1 < q0 + 4) L.o0 = (mb >> (len - 1 - q0)) & 1u;
    }
    cnt = wave_sum_u32(cnt);
    fz = wave_min_i32(fz);
    if (lane_id() == 0) {
        L.cnt[wave_id()] = cnt;
        L.fz[wave_id()] = fz;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t M0 = 0;
        int F = 0x7fffffff;
        for (int w = 0; w < MT / 64; w++) {
            M0 += L.cnt[w];
            F = min(F, L.fz[w]);
        }
        const int P = min(F, len);
        L.packed = (uint64_t)M0 | ((uint64_t)(P & 1) << 32) | ((uint64_t)(F >= len) << 33) |
                   ((uint64_t)L.o0 << 34);
    }
    __syncthreads();
    return L.packed;
}

// pass 1 of the three-pass merge
__global__ void __launch_bounds__(MT)
k_merge_count(const uint32_t *__restrict__ ids, const DevState *__restrict__ st, int par,
              uint64_t *__restrict__ tsum) {
    __shared__ int s_wave[MT / 64];
    __shared__ SummaryLds s_sum;
    if (st->status) return;
    const uint64_t n = st->n[par];
    const uint64_t tile_base = (uint64_t)blockIdx.x * TILE;
    if (tile_base >= n) return;
    const int len = (int)min((uint64_t)TILE, n - tile_base);
    uint32_t a, b;
    if (!resolved_pair(st, ids, a, b)) return;  // k_tile_scan raises ST_INTERNAL
    Tile t;
    tile_load(t, ids, n, tile_base, a, b, s_wave);  // contains a __syncthreads
    const uint64_t w = tile_summary(t, len, s_sum);
    if (threadIdx.x == 0) tsum[blockIdx.x] = w;
}

// pass 2: one workgroup turns the tile summaries into (carry s, output offset)
// per tile.  A tile acts on the carry as a 2-state transducer; transducers
// compose associatively, so the 1024 per-thread range summaries are combined
// with a wave-shuffle scan instead of a serial walk.
struct TS {
    unsigned long long k0, k1;  // ids kept by the range for carry-in 0 / 1
    uint32_t o;                 // bit 0: carry-out for carry-in 0, bit 1: for carry-in 1
};
__device__ __forceinline__ TS ts_then(const TS &A, const TS &B) {  // A followed by B
    const uint32_t a0 = A.o & 1u, a1 = (A.o >> 1) & 1u, b0 = B.o & 1u, b1 = (B.o >> 1) & 1u;
    TS r;
    r.k0 = A.k0 + (a0 ? B.k1 : B.k0);
    r.k1 = A.k1 + (a1 ? B.k1 : B.k0);
    r.o = (a0 ? b1 : b0) | ((a1 ? b1 : b0) << 1);
    return r;
}
__device__ __forceinline__ TS ts_shfl_up(const TS &v, int d) {
    TS r;
    r.k0 = __shfl_up(v.k0, d);
    r.k1 = __shfl_up(v.k1, d);
    r.o = (uint32_t)__shfl_up((int)v.o, d);
    return r;
}
__device__ __forceinline__ void tile_step(uint64_t w, uint32_t len, uint32_t s,
                                          unsigned long long &kept, uint32_t &sout) {
    const uint32_t M0 = (uint32_t)w, podd = (w >> 32) & 1, allones = (w >> 33) & 1, o0 = (w >> 34) & 1;
    const uint32_t Ms = M0 - (s & podd);
    const uint32_t os = allones ? (o0 ^ s) : o0;
    kept += len - s - (Ms - os);
    sout = os;
}

__global__ void __launch_bounds__(1024)
k_tile_scan(const uint64_t *__restrict__ tsum, uint64_t ntiles, uint64_t *__restrict__ tile_off,
            uint8_t *__restrict__ tile_sin, DevState *st, int par, IterRec *rec, int iter,
            const uint32_t *__restrict__ ids, uint32_t *dirty_n) {
    __shared__ TS s_w[16];
    __shared__ uint32_t s_status;
    if (threadIdx.x == 0) {
        // make the pair decision final (k_select / k_tiebreak) and report it
        if (dirty_n) *dirty_n = 0;
        if (st->status == 0 && !st->found) {
            uint32_t a, b;
            if (resolved_pair(st, ids, a, b)) {
                st->a = (int32_t)a;
                st->b = (int32_t)b;
                st->found = 1;
            } else {
                st->status = ST_INTERNAL;
            }
        }
        st->fin_a = st->a;
        st->fin_b = st->b;
        s_status = st->status;
        if (rec) {
            rec[iter].a = st->a;
            rec[iter].b = st->b;
            rec[iter].count = st->count;
            rec[iter].status = st->status;
        }
    }
    __syncthreads();
    if (s_status) {
        if (threadIdx.x == 0 && rec) {
            rec[iter].new_len = st->n[par];
            __threadfence_system();
            rec[iter].seq = (unsigned long long)iter + 1;
        }
        return;
    }
    const uint64_t n = st->n[par];
    const uint64_t R = (ntiles + 1023) / 1024;
    const uint64_t t0 = min((uint64_t)threadIdx.x * R, ntiles), t1 = min(t0 + R, ntiles);
    TS mine;
    mine.k0 = mine.k1 = 0;
    uint32_t sc0 = 0, sc1 = 1;
    for (uint64_t t = t0; t < t1; t++) {
        const uint64_t tb = t * TILE;
        if (tb >= n) break;
        const uint64_t w = tsum[t];
        const uint32_t len = (uint32_t)min((uint64_t)TILE, n - tb);
        tile_step(w, len, sc0, mine.k0, sc0);
        tile_step(w, len, sc1, mine.k1, sc1);
    }
    mine.o = sc0 | (sc1 << 1);
    // inclusive scan across the workgroup
    const int lane = lane_id(), wave = wave_id();
    TS inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const TS p = ts_shfl_up(inc, d);
        if (lane >= d) inc = ts_then(p, inc);
    }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    TS pre;  // everything before this thread
    pre.k0 = pre.k1 = 0;
    pre.o = 2u;  // identity
    for (int w = 0; w < wave; w++) pre = ts_then(pre, s_w[w]);
    TS exl = ts_shfl_up(inc, 1);
    if (lane == 0) {
        exl.k0 = exl.k1 = 0;
        exl.o = 2u;
    }
    pre = ts_then(pre, exl);
    uint32_t s = pre.o & 1u;               // carry-in of my first tile (stream starts with 0)
    unsigned long long off = pre.k0;
    if (threadIdx.x == 1023) {
        const TS all = ts_then(pre, mine);
        st->n[par ^ 1] = all.k0;
        if (rec) {
            rec[iter].new_len = all.k0;
            __threadfence_system();
            rec[iter].seq = (unsigned long long)iter + 1;
        }
    }
    for (uint64_t t = t0; t < t1; t++) {
        const uint64_t tb = t * TILE;
        if (tb >= n) break;
        const uint64_t w = tsum[t];
        const uint32_t len = (uint32_t)min((uint64_t)TILE, n - tb);
        tile_off[t] = off;
        tile_sin[t] = (uint8_t)s;
        tile_step(w, len, s, off, s);
    }
}

// Rewrite of one tile.  kept[p] = !m[p-1]; a site start emits the new id (and
// keeps the chunk-start flag of its first element).  dst = where the tile's
// first kept id goes.
//
// DELTA: the same pass also records how the pair table changes (SURVEY.md N3,
// done inside the full streaming pass).  Every old pair with a merged element
// disappears, every new pair with a new token appears; with (a,b) -> Z they are
// exactly (L,a), (b,R), (L,Z), (Z,R), so four vectors indexed by one token
// describe the whole update:
//   decL[L] : pairs (L,a) destroyed      decR[R] : pairs (b,R) destroyed
//   incL[L] : pairs (L,Z) created        incR[R] : pairs (Z,R) created (R may be Z)
// Each destroyed pair is charged to its left element, each created pair to its
// left output element, so nothing is counted twice.
// own_len: the tile owns positions [0, own_len); words beyond are context only.
// SKIP_UNCHANGED: do not store when no owned element changes (slotted streams:
// the slot simply stays where it is).  *kept_out / *changed_out: block totals.
template <bool DELTA, bool SKIP_UNCHANGED>
__device__ __forceinline__ void tile_rewrite(const Tile &t, uint32_t s, uint32_t a, uint32_t b,
                                             uint32_t newid, uint32_t *__restrict__ dst_tile,
                                             uint32_t *s_wsum, uint32_t *__restrict__ delta,
                                             uint32_t vcap, int own_len, uint32_t *kept_out,
                                             bool *changed_out, uint32_t *__restrict__ hdr4 = nullptr) {
    const int lane = lane_id(), wave = wave_id();
    uint32_t mb[MJ], mp[MJ], kb[MJ], ex[MJ];
    uint32_t carry = 0, chg = 0;
    const int qw = wave * WAVE_SPAN + lane * 4;
#pragma unroll
    for (int j = 0; j < MJ; j++) {
        mb[j] = group_mbits(t, j, s, mp[j]);
        // kept bit k = !m[k-1], only for owned positions
        uint32_t valid = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) valid |= (uint32_t)(qw + j * 256 + k < own_len) << k;
        chg |= mb[j] & valid;
        kb[j] = (~((mb[j] << 1) | mp[j])) & valid & 0xFu;
        const uint32_t v = wave_iscan_add((uint32_t)__popc(kb[j]));
        ex[j] = carry + v - __popc(kb[j]);
        carry += lane_last(v);
    }
    const bool wchg = __any(chg != 0);
    if (lane == 0) s_wsum[wave] = carry | (wchg ? 0x80000000u : 0u);
    __syncthreads();
    uint32_t wbase = 0, total = 0;
    bool changed = (s != 0);
    for (int w = 0; w < MT / 64; w++) {
        const uint32_t v = s_wsum[w];
        if (w < wave) wbase += v & 0x7FFFFFFFu;
        total += v & 0x7FFFFFFFu;
        changed |= (v >> 31) != 0;
    }
    if (kept_out) *kept_out = total;
    if (changed_out) *changed_out = changed;
    if (!SKIP_UNCHANGED || changed) {
        uint32_t *dst = dst_tile + wbase;
#pragma unroll
        for (int j = 0; j < MJ; j++) {
            uint32_t o = ex[j];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if ((kb[j] >> k) & 1u) {
                    const uint32_t w = t.x[j][k];
                    dst[o++] = ((mb[j] >> k) & 1u) ? (newid | (w & (FLAG | WMASK))) : w;
                }
            }
        }
        if (hdr4) {
            // the slot's first three and last output words (the neighbours' context next
            // pass).  Kept out of the store loop above: only the first and the last writer
            // of the tile ever get here.
#pragma unroll
            for (int j = 0; j < MJ; j++) {
                const uint32_t lo = wbase + ex[j], hi = lo + __popc(kb[j]);
                if (kb[j] && (lo < 3 || hi == total)) {
                    uint32_t gi = lo;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if ((kb[j] >> k) & 1u) {
                            const uint32_t w = t.x[j][k];
                            const uint32_t ow = ((mb[j] >> k) & 1u) ? (newid | (w & (FLAG | WMASK))) : w;
                            if (gi < 3) hdr4[gi] = ow;
                            if (gi + 1 == total) hdr4[3] = ow;
                            gi++;
                        }
                    }
                }
            }
        }
    }
    if (DELTA) {
        // m bits and words of the two elements after my group: from the next
        // lane, the next stripe, or (end of the wave) recomputed from the tail.
        const uint32_t t0 = t.tail[0], t1 = t.tail[1], t2 = t.tail[2];
        // most waves are far from any site: skip the whole section for them
        uint32_t near = 0;
#pragma unroll
        for (int j = 0; j < MJ; j++) near |= mb[j] | mp[j];
        near |= (uint32_t)((((t0 & IDMASK) == a) & ((t1 & NWMASK) == b)) |
                           (((t1 & IDMASK) == a) & ((t2 & NWMASK) == b)));
        if (!__any(near != 0)) return;
        // same-address atomics serialise (~11 ns each): spread them over replicas
        const uint32_t nrep = 1u << (vcap >> 24);  // host packs log2(replicas) above the stride
        vcap &= 0xFFFFFFu;
        delta += (size_t)(blockIdx.x & (nrep - 1)) * 4 * vcap;
#pragma unroll
        for (int j = 0; j < MJ; j++) {
            const uint32_t nb_m = lane_next(mb[j], 0);
            const uint32_t nb_x0 = lane_next(t.x[j][0], 0);
            const uint32_t nb_x1 = lane_next(t.x[j][1], 0);
            uint32_t up_m, up_x0, up_x1;
            if (j < MJ - 1) {
                up_m = lane_first(mb[(j + 1) % MJ]);
                up_x0 = lane_first(t.x[(j + 1) % MJ][0]);
                up_x1 = lane_first(t.x[(j + 1) % MJ][1]);
            } else {
                const uint32_t m3 = (mb[j] >> 3) & 1u;  // only lane 63's value is used
                const uint32_t r4 = (uint32_t)(((t0 & IDMASK) == a) & ((t1 & NWMASK) == b));
                const uint32_t m4 = r4 & (m3 ^ 1u);
                const uint32_t r5 = (uint32_t)(((t1 & IDMASK) == a) & ((t2 & NWMASK) == b));
                const uint32_t m5 = r5 & (m4 ^ 1u);
                up_m = m4 | (m5 << 1);
                up_x0 = t0;
                up_x1 = t1;
            }
            const bool last = (lane == 63);
            const uint32_t X[6] = {t.x[j][0], t.x[j][1], t.x[j][2], t.x[j][3],
                                   last ? up_x0 : nb_x0, last ? up_x1 : nb_x1};
            // bit (k+1) = m[k], k = -1..5
            const uint32_t Mx = mp[j] | (mb[j] << 1) | (((last ? up_m : nb_m) & 3u) << 5);
            if (mb[j] | mp[j] | (Mx >> 5)) {  // nothing to record far from any site
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t Mk = (Mx >> (k + 1)) & 1u, Mkm1 = (Mx >> k) & 1u,
                                   Mkp1 = (Mx >> (k + 2)) & 1u;
                    if (qw + j * 256 + k >= own_len) continue;  // context word, not mine
                    const uint32_t wt = word_weight(X[k]);  // every word involved shares X[k]'s chunk
                    if (!(X[k + 1] & FLAG) && !Mk) {  // an old pair that is not the site itself
                        if (Mkm1) atomicAdd(&delta[1 * (size_t)vcap + (X[k + 1] & IDMASK)], wt);
                        else if (Mkp1) atomicAdd(&delta[0 * (size_t)vcap + (X[k] & IDMASK)], wt);
                    }
                    if (!Mkm1) {  // output element
                        const uint32_t Xq = Mk ? X[k + 2] : X[k + 1];
                        const uint32_t Mq = Mk ? ((Mx >> (k + 3)) & 1u) : Mkp1;
                        if (!(Xq & FLAG)) {
                            if (Mk) atomicAdd(&delta[3 * (size_t)vcap + (Mq ? newid : (Xq & IDMASK))], wt);
                            else if (Mq) atomicAdd(&delta[2 * (size_t)vcap + (X[k] & IDMASK)], wt);
                        }
                    }
                }
            }
        }
    }
}

// pass 3 of the three-pass merge
template <bool DELTA>
__global__ void __launch_bounds__(MT)
k_merge_scatter(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                const DevState *__restrict__ st, int par, const uint64_t *__restrict__ tile_off,
                const uint8_t *__restrict__ tile_sin, uint32_t newid, uint32_t *__restrict__ delta,
                uint32_t vcap) {
    __shared__ int s_wave[MT / 64];
    __shared__ uint32_t s_wsum[MT / 64];
    if (st->status) return;
    const uint64_t n = st->n[par];
    const uint64_t tile_base = (uint64_t)blockIdx.x * TILE;
    if (tile_base >= n) return;
    const uint32_t a = (uint32_t)st->fin_a, b = (uint32_t)st->fin_b;
    Tile t;
    tile_load(t, in, n, tile_base, a, b, s_wave);
    tile_rewrite<DELTA, false>(t, tile_sin[blockIdx.x], a, b, newid, out + tile_off[blockIdx.x], s_wsum,
                               delta, vcap, (int)min((uint64_t)TILE, n - tile_base), nullptr, nullptr);
}

// ---------------------------------------------------------------------------
// Single-pass merge: summary, carry/offset resolution and rewrite in ONE sweep
// over the ids (reads 4N, writes 4N' -- the three-pass form reads 8N).
//
// Chained scan with TWO-LEVEL decoupled look-back.  Tile t publishes its
// transducer summary ("aggregate") as soon as it has read its ids; the last
// tile of every group of 64 also publishes the group's aggregate.  A tile then
// resolves its carry and output offset in two hops: (1) the <= 63 tiles before
// it in its own group, (2) the groups before its group, 64 per hop, until one
// is found whose inclusive prefix is known.  With a single level the prefix
// frontier advances 64 tiles per L2 round trip (~1 us) -- measured: that alone
// caps the pass at ~2 TB/s; with two levels it advances 4096 tiles per hop.
//
// Descriptors are single 8-byte words written/read with agent-scope relaxed
// atomics (sc1: they bypass the non-coherent per-CU L1 / per-XCD L2), so the
// data IS the flag and no fence is needed (cdna_hip_programming.md G16, R2).
// They carry an epoch, so they never need clearing between launches.
//   bits 63..62 status (1 aggregate, 2 inclusive prefix)   bits 61..42 epoch
//   aggregate: bits 0..19 k0, 20..39 k1, 40 o0, 41 o1  (kept ids / carry-out per carry-in)
//   prefix   : bits 0..35 inclusive kept count, bit 36 carry-out
// Progress: tiles are workgroup ids, dispatched in order, so every tile a
// workgroup waits on is resident or done, and aggregates are published before
// any waiting; the spin is bounded anyway and raises ST_LOOKBACK, never hangs.
__device__ __forceinline__ unsigned long long desc_load(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void desc_store(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long desc_pack_agg(const TS &v, unsigned long long tag) {
    return (1ull << 62) | tag | v.k0 | (v.k1 << 20) | ((unsigned long long)(v.o & 3u) << 40);
}
__device__ __forceinline__ unsigned long long desc_pack_prefix(unsigned long long incl, uint32_t sout,
                                                               unsigned long long tag) {
    return (2ull << 62) | tag | (incl & 0xFFFFFFFFFull) | ((unsigned long long)sout << 36);
}
// descriptor -> transducer (a prefix is a constant function)
__device__ __forceinline__ TS desc_unpack(unsigned long long d, uint32_t stt) {
    TS v;
    if (stt == 2) {
        v.k0 = v.k1 = d & 0xFFFFFFFFFull;
        v.o = ((d >> 36) & 1u) ? 3u : 0u;
    } else {
        v.k0 = d & 0xFFFFFu;
        v.k1 = (d >> 20) & 0xFFFFFu;
        v.o = (uint32_t)((d >> 40) & 3u);
    }
    return v;
}

// One look-back hop over descriptors arr[base], arr[base-1], ... (lane i reads
// arr[base-i]; indices below `floor` do not exist: below 0 they act as the
// prefix (0, carry 0), otherwise they are simply outside the window).  Waits
// until the nearest prefix and every nearer descriptor are published, composes
// them far -> near.  Returns the composition in `win`; found_prefix tells
// whether it is absolute.  false on timeout.
__device__ __forceinline__ bool lookback_hop(const unsigned long long *arr, long long base,
                                             long long floor_idx, int count, uint32_t epoch,
                                             TS &win, bool &found_prefix, uint32_t tune) {
    const int lane = lane_id();
    const long long idx = base - lane;
    const bool inwin = lane < count && idx >= floor_idx;
    unsigned long long d = 0;
    uint32_t stt = 0;
    unsigned long long pmask = 0;
    for (uint32_t spins = 0;; spins++) {
        if (inwin) {
            if (idx >= 0) {
                d = desc_load(&arr[idx]);
                stt = ((d >> 42) & EPOCH_MASK) == (epoch & EPOCH_MASK) ? (uint32_t)(d >> 62) : 0u;
            } else {
                d = 0;
                stt = 2;  // before the stream: prefix 0, carry 0
            }
        } else {
            stt = 1;  // outside the window: neutral
        }
        pmask = __ballot(inwin && stt == 2);
        const int np = pmask ? (__ffsll((long long)pmask) - 1) : 63;
        if (!__ballot(inwin && stt == 0 && lane <= np)) break;
        if (spins > LOOKBACK_SPINS) return false;
        // back off: every poll is an L2-bypassing load that competes with the stream
        for (uint32_t z = 0; z < (tune & 0xFFu); z++) __builtin_amdgcn_s_sleep(8);
    }
    const int np = pmask ? (__ffsll((long long)pmask) - 1) : 64;
    TS v;
    if (!inwin || lane > np) {
        v.k0 = v.k1 = 0;
        v.o = 2u;  // identity
    } else {
        v = desc_unpack(d, stt);
    }
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {  // ordered: far tiles first, lane 0 last
        TS far;
        far.k0 = __shfl_down(v.k0, sft);
        far.k1 = __shfl_down(v.k1, sft);
        far.o = (uint32_t)__shfl_down((int)v.o, sft);
        if (lane + sft < 64) v = ts_then(far, v);
    }
    win.k0 = __shfl(v.k0, 0);
    win.k1 = __shfl(v.k1, 0);
    win.o = (uint32_t)__shfl((int)v.o, 0);
    found_prefix = pmask != 0;
    return true;
}

template <bool DELTA>
__global__ void __launch_bounds__(MT)
k_merge_lookback(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, DevState *st, int par,
                 unsigned long long *__restrict__ desc, unsigned long long *__restrict__ gdesc,
                 uint32_t epoch, uint32_t newid, uint32_t *__restrict__ delta, uint32_t vcap,
                 IterRec *rec, int iter, uint32_t *dirty_n, uint32_t tune) {
    __shared__ int s_wave[MT / 64];
    __shared__ uint32_t s_wsum[MT / 64];
    __shared__ SummaryLds s_sum;
    __shared__ unsigned long long s_excl;
    __shared__ uint32_t s_sin, s_fail;
    const uint64_t n = st->n[par];
    const uint64_t tile = blockIdx.x;
    const uint64_t tile_base = tile * TILE;
    uint32_t a = 0, b = 0;
    const bool ok = (st->status == 0) && resolved_pair(st, in, a, b);
    if (!ok) {
        // nothing to merge: tile 0 reports (empty stats, or a tie nobody resolved)
        if (tile == 0 && threadIdx.x == 0) {
            if (st->status == 0) st->status = ST_INTERNAL;
            if (dirty_n) *dirty_n = 0;
            if (rec) {
                rec[iter].a = st->a;
                rec[iter].b = st->b;
                rec[iter].count = st->count;
                rec[iter].status = st->status;
                rec[iter].new_len = n;
                __threadfence_system();
                rec[iter].seq = (unsigned long long)iter + 1;
            }
        }
        return;
    }
    if (tile_base >= n) return;
    const int len = (int)min((uint64_t)TILE, n - tile_base);
    Tile t;
    tile_load(t, in, n, tile_base, a, b, s_wave);
    const uint64_t w = tile_summary(t, len, s_sum);
    TS own;  // the tile as a transducer
    own.k0 = own.k1 = 0;
    uint32_t o0 = 0, o1 = 1;
    tile_step(w, (uint32_t)len, 0u, own.k0, o0);
    tile_step(w, (uint32_t)len, 1u, own.k1, o1);
    own.o = o0 | (o1 << 1);
    const unsigned long long tag = ((unsigned long long)(epoch & EPOCH_MASK)) << 42;
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const long long grp = (long long)(tile >> 6);
        const int li = (int)(tile & 63);
        bool fail = false;
        if (lane == 0) desc_store(&desc[tile], desc_pack_agg(own, tag));
        const bool fake = (tune >> 8) & 1u;  // measurement only: skip the waiting (wrong output)
        // Both windows are polled in the same round trip: lane i reads the
        // descriptor of tile t-1-i (my group only) AND of group grp-1-i.
        TS pre;
        pre.k0 = pre.k1 = 0;
        pre.o = 2u;
        if (!fake) {
            const long long i1 = (long long)tile - 1 - lane;   // level 1
            const bool in1 = lane < li;
            const long long i2 = grp - 1 - lane;               // level 2
            unsigned long long d1 = 0, d2 = 0;
            uint32_t s1 = 1, s2 = 1;
            unsigned long long p1 = 0, p2 = 0;
            bool done1 = false, pubbed = false;
            TS w1;
            w1.k0 = w1.k1 = 0;
            w1.o = 2u;
            for (uint32_t spins = 0;; spins++) {
                if (in1 && !done1) d1 = desc_load(&desc[i1]);
                if (i2 >= 0) d2 = desc_load(&gdesc[i2]);
                if (in1 && !done1)
                    s1 = ((d1 >> 42) & EPOCH_MASK) == (epoch & EPOCH_MASK) ? (uint32_t)(d1 >> 62) : 0u;
                s2 = (i2 >= 0) ? (((d2 >> 42) & EPOCH_MASK) == (epoch & EPOCH_MASK) ? (uint32_t)(d2 >> 62) : 0u)
                               : 2u;  // before the stream: prefix 0, carry 0
                if (i2 < 0) d2 = 0;
                if (!done1) {
                    p1 = __ballot(in1 && s1 == 2);
                    const int np1 = p1 ? (__ffsll((long long)p1) - 1) : 63;
                    done1 = !__ballot(in1 && s1 == 0 && lane <= np1);
                    if (done1) {  // compose my group's tiles before me, far -> near
                        const int np = p1 ? (__ffsll((long long)p1) - 1) : 64;
                        TS v;
                        if (!in1 || lane > np) {
                            v.k0 = v.k1 = 0;
                            v.o = 2u;
                        } else {
                            v = desc_unpack(d1, s1);
                        }
#pragma unroll
                        for (int sft = 1; sft < 64; sft <<= 1) {
                            TS far;
                            far.k0 = __shfl_down(v.k0, sft);
                            far.k1 = __shfl_down(v.k1, sft);
                            far.o = (uint32_t)__shfl_down((int)v.o, sft);
                            if (lane + sft < 64) v = ts_then(far, v);
                        }
                        w1.k0 = __shfl(v.k0, 0);
                        w1.k1 = __shfl(v.k1, 0);
                        w1.o = (uint32_t)__shfl((int)v.o, 0);
                    }
                }
                if (done1 && !p1 && li == 63 && !pubbed) {  // my group's aggregate, as early as possible
                    if (lane == 0) desc_store(&gdesc[grp], desc_pack_agg(ts_then(w1, own), tag));
                    pubbed = true;
                }
                if (done1 && p1) {  // a prefix inside my own group: absolute already
                    pre = w1;
                    break;
                }
                p2 = __ballot(s2 == 2);
                const int np2 = p2 ? (__ffsll((long long)p2) - 1) : 63;
                const bool done2 = !__ballot(s2 == 0 && lane <= np2);
                if (done1 && done2) {
                    const int np = p2 ? (__ffsll((long long)p2) - 1) : 64;
                    TS v;
                    if (lane > np) {
                        v.k0 = v.k1 = 0;
                        v.o = 2u;
                    } else {
                        v = desc_unpack(d2, s2);
                    }
#pragma unroll
                    for (int sft = 1; sft < 64; sft <<= 1) {
                        TS far;
                        far.k0 = __shfl_down(v.k0, sft);
                        far.k1 = __shfl_down(v.k1, sft);
                        far.o = (uint32_t)__shfl_down((int)v.o, sft);
                        if (lane + sft < 64) v = ts_then(far, v);
                    }
                    TS w2;
                    w2.k0 = __shfl(v.k0, 0);
                    w2.k1 = __shfl(v.k1, 0);
                    w2.o = (uint32_t)__shfl((int)v.o, 0);
                    pre = ts_then(w2, w1);
                    if (!p2) {  // 64 groups of aggregates and still no prefix: keep walking back
                        long long gb = grp - 1 - 64;
                        for (;;) {
                            TS win;
                            bool found = false;
                            if (!lookback_hop(gdesc, gb, -(1ll << 62), 64, epoch, win, found, tune)) {
                                fail = true;
                                break;
                            }
                            pre = ts_then(win, pre);
                            if (found) break;
                            gb -= 64;
                        }
                    }
                    break;
                }
                if (spins > LOOKBACK_SPINS) {
                    fail = true;
                    break;
                }
                for (uint32_t z = 0; z < (tune & 0xFFu); z++) __builtin_amdgcn_s_sleep(8);
            }
        }
        if (lane == 0) {
            if (fake) pre.k0 = tile * TILE;
            const unsigned long long excl = pre.k0;  // chain starts at a prefix: input-independent
            const uint32_t sin = pre.o & 1u;
            const unsigned long long incl = excl + (sin ? own.k1 : own.k0);
            const uint32_t sout = sin ? o1 : o0;
            const unsigned long long pd = desc_pack_prefix(incl, sout, tag);
            desc_store(&desc[tile], pd);
            if (li == 63) desc_store(&gdesc[grp], pd);
            s_excl = excl;
            s_sin = sin;
            s_fail = fail;
            if (fail) atomicExch(&st->status, ST_LOOKBACK);
            if (tile_base + TILE >= n) {  // last tile: totals, report, final pair
                st->n[par ^ 1] = incl;
                st->fin_a = (int32_t)a;
                st->fin_b = (int32_t)b;
                if (dirty_n) *dirty_n = 0;
                if (rec) {
                    rec[iter].a = (int32_t)a;
                    rec[iter].b = (int32_t)b;
                    rec[iter].count = st->count;
                    rec[iter].status = fail ? ST_LOOKBACK : 0u;
                    rec[iter].new_len = incl;
                    __threadfence_system();
                    rec[iter].seq = (unsigned long long)iter + 1;
                }
            }
        }
    }
    __syncthreads();
    if (s_fail) return;
    tile_rewrite<DELTA, false>(t, s_sin, a, b, newid, out + s_excl, s_wsum, delta, vcap, len, nullptr,
                               nullptr);
}

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

// Apply the four delta vectors to the dense table and keep rowmax[] current.
// Thread t owns token t: column a, row b, the new column Z and the new row Z.
// Rows whose maximum may have dropped are queued for k_rowmax_list; for every
// other row the only entry that grew is the brand-new column Z.
template <bool FOLDED>
__device__ __forceinline__ void apply_body(uint32_t *__restrict__ mat, uint32_t stride,
                                           uint32_t *__restrict__ delta, uint32_t vcap,
                                           uint32_t *__restrict__ rowmax, DevState *st, uint32_t Z,
                                           uint32_t *__restrict__ dirty_list,
                                           uint32_t *__restrict__ dirty_n, int par, IterRec *rec, int iter,
                                           int slot_finish) {
    if (slot_finish && blockIdx.x == 0 && threadIdx.x == 0) {
        // slotted pass: new stream length and this iteration's record
        const unsigned long long n = st->n[par];
        unsigned long long nn = n;
        if (st->status == 0) {
            nn = n - st->removed;
            st->n[par ^ 1] = nn;
        }
        st->removed = 0;
        if (rec) {
            rec[iter].a = st->status == 0 ? st->fin_a : st->a;
            rec[iter].b = st->status == 0 ? st->fin_b : st->b;
            rec[iter].count = st->count;
            rec[iter].status = st->status;
            rec[iter].new_len = nn;
            __threadfence_system();
            rec[iter].seq = (unsigned long long)iter + 1;
        }
    }
    if (st->status) return;
    // 8 lanes per token: each folds a quarter of the replicas (all its loads in flight at
    // once), then a 3-step shuffle sum.  The kernel is latency-bound, so width, not work, counts.
    const uint32_t g = threadIdx.x & 7u;
    const uint32_t t = blockIdx.x * (blockDim.x / 8) + (threadIdx.x >> 3);
    const bool live = t <= Z;
    const uint32_t a = (uint32_t)st->fin_a, b = (uint32_t)st->fin_b;
    uint32_t acc4[4] = {0, 0, 0, 0};
    const uint32_t nrep = 1u << (vcap >> 24);
    vcap &= 0xFFFFFFu;
    if (FOLDED) {
        if (live && g == 0) {
#pragma unroll
            for (int v = 0; v < 4; v++) acc4[v] = delta[(size_t)v * vcap + t];
        }
    } else if (live) {
        uint32_t x[4][4];
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const uint32_t r = g + 8u * k;
                x[k][v] = (r < nrep) ? delta[((size_t)r * 4 + v) * vcap + t] : 0u;
            }
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int v = 0; v < 4; v++) {
                if (x[k][v]) delta[((size_t)(g + 8u * k) * 4 + v) * vcap + t] = 0;
                acc4[v] += x[k][v];
            }
    }
#pragma unroll
    for (int v = 0; v < 4; v++) {
        acc4[v] += (uint32_t)__shfl_xor((int)acc4[v], 1);
        acc4[v] += (uint32_t)__shfl_xor((int)acc4[v], 2);
        acc4[v] += (uint32_t)__shfl_xor((int)acc4[v], 4);
    }
    if (!live || g != 0) return;
    const uint32_t dl = acc4[0], dr = acc4[1], il = acc4[2], ir = acc4[3];
    bool dirty = (t == a) | (t == b) | (t == Z);  // always recomputed
    if (dl) {
        const uint32_t old = atomicSub(&mat[(size_t)t * stride + a], dl);
        if (t != Z && old == rowmax[t]) dirty = true;
    }
    if (dr) atomicSub(&mat[(size_t)b * stride + t], dr);
    if (il) atomicAdd(&mat[(size_t)t * stride + Z], il);
    if (ir) atomicAdd(&mat[(size_t)Z * stride + t], ir);
    if (dirty) {
        dirty_list[atomicAdd(dirty_n, 1u)] = t;
    } else if (il > rowmax[t]) {
        rowmax[t] = il;  // column Z was empty before this iteration
    }
}

// Recompute rowmax for the queued rows; also retires the merged pair: after the
// merge no (a,b) remains (F2), whatever the a == b bookkeeping left there.
__device__ __forceinline__ void rowmax_body(uint32_t *__restrict__ mat, uint32_t stride, uint32_t vnew,
                                            uint32_t *__restrict__ rowmax, const DevState *st,
                                            const uint32_t *__restrict__ dirty_list,
                                            const uint32_t *__restrict__ dirty_n, uint32_t first,
                                            uint32_t step) {
    __shared__ uint32_t s_red[4];
    if (st->status) return;
    const uint32_t a = (uint32_t)st->fin_a, b = (uint32_t)st->fin_b;
    const uint32_t nd = *dirty_n;
    for (uint32_t i = first; i < nd; i += step) {
        const uint32_t x = dirty_list[i];
        uint32_t *row = mat + (size_t)x * stride;
        uint32_t m = 0;
        for (uint32_t y = threadIdx.x; y < vnew; y += 256) {
            uint32_t v = row[y];
            if (x == a && y == b) {
                v = 0;
                row[y] = 0;
            }
            m = max(m, v);
        }
        m = wave_max_u32(m);
        __syncthreads();
        if (lane_id() == 0) s_red[wave_id()] = m;
        __syncthreads();
        if (threadIdx.x == 0) rowmax[x] = max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3]));
    }
}
__global__ void __launch_bounds__(256)
k_rowmax_list(uint32_t *__restrict__ mat, uint32_t stride, uint32_t vnew,
              uint32_t *__restrict__ rowmax, const DevState *__restrict__ st,
              const uint32_t *__restrict__ dirty_list, const uint32_t *__restrict__ dirty_n) {
    rowmax_body(mat, stride, vnew, rowmax, st, dirty_list, dirty_n, blockIdx.x, gridDim.x);
}

// Table update in one launch: blocks [0, na) apply the delta vectors, blocks
// [na, gridDim) wait until all of them are done (a monotonic counter, agent-scope
// release/acquire) and recompute the queued row maxima.  The apply blocks never
// wait and come first in dispatch order, so the wait always ends.
template <bool FOLDED>
__global__ void __launch_bounds__(256)
k_apply_delta(uint32_t *__restrict__ mat, uint32_t stride, uint32_t *__restrict__ delta,
              uint32_t vcap, uint32_t *__restrict__ rowmax, DevState *st, uint32_t Z,
              uint32_t *__restrict__ dirty_list, uint32_t *__restrict__ dirty_n, int par, IterRec *rec,
              int iter, int slot_finish, uint32_t na, unsigned long long target) {
    if (blockIdx.x < na) {
        apply_body<FOLDED>(mat, stride, delta, vcap, rowmax, st, Z, dirty_list, dirty_n, par, rec, iter,
                           slot_finish);
        if (target == 0) return;  // row maxima run as their own launch (the default)
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(&st->apply_done, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    __shared__ uint32_t s_ok;
    if (threadIdx.x == 0) {
        bool ok = false;
        for (uint32_t spins = 0; spins < LOOKBACK_SPINS; spins++) {
            if (__hip_atomic_load(&st->apply_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) {
                ok = true;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (!ok) atomicExch(&st->status, ST_LOOKBACK);
        s_ok = ok;
    }
    __syncthreads();
    if (!s_ok) return;
    rowmax_body(mat, stride, Z + 1, rowmax, st, dirty_list, dirty_n, blockIdx.x - na, gridDim.x - na);
}

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
__global__ void k_dp_key(SlotRef ref, int par, const DevState *__restrict__ st,
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
}
__global__ void k_dp_resolve(DevState *st, const long long *__restrict__ key) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (st->status) return;
    if (key[0] == 0x7FFFFFFFFFFFFFFFll) {
        st->status = ST_INTERNAL;  // a tie at the maximum, yet no rank holds a tied pair
        return;
    }
    st->a = (int32_t)(key[0] & 0xFFFF);
    st->b = (int32_t)(key[1] & 0xFFFF);
    st->found = 1;
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
                const uint32_t x = delta[((size_t)r * 4 + v) * vcap + t];
                if (x) delta[((size_t)r * 4 + v) * vcap + t] = 0;
                acc += x;
            }
        }
        folded[(size_t)v * vcap + t] = acc;
    }
}

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

__global__ void __launch_bounds__(ENC_THREADS)
k_encode_short(const uint8_t *__restrict__ bytes, const uint64_t *__restrict__ off, uint64_t n_chunks,
               uint64_t n, const unsigned long long *__restrict__ keys,
               const uint32_t *__restrict__ vals, uint32_t mask, const int32_t *__restrict__ merge_ids,
               uint32_t *__restrict__ tmp, uint32_t *__restrict__ outlen,
               unsigned long long *__restrict__ long_list, unsigned long long *__restrict__ n_long) {
    __shared__ uint32_t s_tok[ENC_LMAX * ENC_THREADS];
    __shared__ uint32_t s_rk[ENC_LMAX * ENC_THREADS];
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
    uint32_t *tok = s_tok + threadIdx.x;  // element i at tok[i * ENC_THREADS]
    uint32_t *rk = s_rk + threadIdx.x;
    for (uint32_t i = 0; i < L; i++) tok[i * ENC_THREADS] = bytes[s0 + i];
    for (uint32_t i = 0; i + 1 < L; i++)
        rk[i * ENC_THREADS] = rank_lookup(keys, vals, mask, tok[i * ENC_THREADS], tok[(i + 1) * ENC_THREADS]);
    while (L >= 2) {
        uint32_t best = 0xFFFFFFFFu, bi = 0;
        for (uint32_t i = 0; i + 1 < L; i++) {
            const uint32_t r = rk[i * ENC_THREADS];
            if (r < best) {  // strict: leftmost occurrence of the lowest rank
                best = r;
                bi = i;
            }
        }
        if (best == 0xFFFFFFFFu) break;  // nothing else can be merged
        tok[bi * ENC_THREADS] = merge_ids ? (uint32_t)merge_ids[best] : 256u + best;
        for (uint32_t i = bi + 1; i + 1 < L; i++) {
            tok[i * ENC_THREADS] = tok[(i + 1) * ENC_THREADS];
            rk[i * ENC_THREADS] = rk[(i + 1) * ENC_THREADS];
        }
        L--;
        if (bi > 0)
            rk[(bi - 1) * ENC_THREADS] =
                rank_lookup(keys, vals, mask, tok[(bi - 1) * ENC_THREADS], tok[bi * ENC_THREADS]);
        if (bi + 1 < L)
            rk[bi * ENC_THREADS] =
                rank_lookup(keys, vals, mask, tok[bi * ENC_THREADS], tok[(bi + 1) * ENC_THREADS]);
    }
    for (uint32_t i = 0; i < L; i++) tmp[s0 + i] = tok[i * ENC_THREADS];
    outlen[c] = L;
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

// ---------------------------------------------------------------------------
// batch decode (N4): token id -> bytes through the vocab table resident in HBM

__global__ void __launch_bounds__(256)
k_decode_len(const int32_t *__restrict__ ids, uint64_t n, const unsigned long long *__restrict__ voff,
             uint32_t V, uint32_t *__restrict__ len, unsigned long long *bad) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t id = (uint32_t)ids[i];  // negative ids wrap above V
        uint32_t L = 0;
        if (id < V)
            L = (uint32_t)(voff[id + 1] - voff[id]);
        else
            atomicMin(bad, (unsigned long long)i);
        len[i] = L;
    }
}

// One token per lane.  Tokens are a few bytes each, so a wave's 64 tokens cover a few
// hundred consecutive output bytes; the table (<= a few MB) stays in L2.
__global__ void __launch_bounds__(256)
k_decode_copy(const int32_t *__restrict__ ids, uint64_t n, const unsigned long long *__restrict__ voff,
              uint32_t V, const uint8_t *__restrict__ blob, const unsigned long long *__restrict__ off,
              uint8_t *__restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t id = (uint32_t)ids[i];
        if (id >= V) continue;
        const unsigned long long s0 = voff[id], L = voff[id + 1] - s0, d0 = off[i];
        for (unsigned long long k = 0; k < L; k++) out[d0 + k] = blob[s0 + k];
    }
}

// dst[j] = byte offset of token position idx[j] (position n: the total)
__global__ void __launch_bounds__(256)
k_decode_doc_offsets(const unsigned long long *__restrict__ off, uint64_t n, unsigned long long total,
                     const unsigned long long *__restrict__ idx, uint64_t k,
                     unsigned long long *__restrict__ dst) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= k) return;
    const unsigned long long p = idx[j];
    dst[j] = p < n ? off[p] : total;
}

// ---------------------------------------------------------------------------
// table utilities for the single-step API

__global__ void __launch_bounds__(256)
k_count_nonzero(const uint32_t *__restrict__ mat, uint32_t stride, uint32_t vcur,
                unsigned long long *out) {
    const uint32_t x = blockIdx.x;
    uint32_t c = 0;
    for (uint32_t y = threadIdx.x; y < vcur; y += 256) c += mat[(size_t)x * stride + y] != 0;
    c = wave_sum_u32(c);
    if (lane_id() == 0 && c) atomicAdd(out, (unsigned long long)c);
}

__global__ void __launch_bounds__(256)
k_dump_stats(const uint32_t *__restrict__ mat, const uint32_t *__restrict__ first, uint32_t stride,
             uint32_t vcur, int32_t *oa, int32_t *ob, unsigned long long *oc,
             unsigned long long *of, unsigned long long cap, unsigned long long *cursor) {
    const uint32_t x = blockIdx.x;
    for (uint32_t y = threadIdx.x; y < vcur; y += 256) {
        const uint32_t c = mat[(size_t)x * stride + y];
        if (c) {
            const unsigned long long s = atomicAdd(cursor, 1ull);
            if (s < cap) {
                oa[s] = (int32_t)x;
                ob[s] = (int32_t)y;
                oc[s] = c;
                of[s] = first ? first[(size_t)x * stride + y] : 0;
            }
        }
    }
}

__global__ void __launch_bounds__(256)
k_strip_flags(const uint32_t *__restrict__ in, int32_t *__restrict__ out, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = (int32_t)(in[i] & IDMASK);
}

__global__ void __launch_bounds__(256)
k_collect_starts(const uint32_t *__restrict__ in, uint64_t n, unsigned long long *out,
                 unsigned long long cap, unsigned long long *cursor) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (in[i] & FLAG) {
            const unsigned long long s = atomicAdd(cursor, 1ull);
            if (s < cap) out[s] = i;
        }
    }
}

__global__ void k_load_ids(const int32_t *__restrict__ in, uint32_t *__restrict__ out, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = (uint32_t)in[i] & IDMASK;
}

__global__ void k_init_state(DevState *st, unsigned long long n) {
    st->n[0] = n;
    st->n[1] = 0;
    st->firstpos = NOPOS;
    st->a = st->b = 0;
    st->count = 0;
    st->ntied = 0;
    st->found = 0;
    st->status = 0;
    st->fin_a = st->fin_b = 0;
    st->removed = 0;
    st->apply_done = 0;
    st->sel_flag = 0;
}

}  // namespace bpe
