// k_merge.hip -- K3: the merge tile code, the three-pass merge.
// Part of bpe_kernels.hip, which includes the parts in order.
// (no include guard: bpe_kernels.hip includes this part once per geometry, namespace BPE_G)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../bpe_device.h"

namespace bpe {
namespace BPE_G {

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

// The same scan for streams of many tiles (round 6): k_tile_scan's one workgroup walks ntiles / 1024 summaries per
// thread, one dependent L2 round trip each, twice -- 0.40 ms per iteration at 1 GB (240 k tiles), 15 % of a recount
// iteration.  Three small launches instead: k_tile_sup composes the transducers of 256 consecutive tiles (one coalesced
// load per thread, a workgroup scan), k_tile_scan_sup scans the <= few thousand super-tiles as k_tile_scan scans tiles
// (and does its bookkeeping: the pair made final, the record, the new length), k_tile_expand gives every tile its
// carry-in and offset from its super-tile's.  Same outputs as k_tile_scan, bit for bit.
constexpr int SUP_TILES = 256;
__device__ __forceinline__ TS ts_identity() {
    TS r;
    r.k0 = r.k1 = 0;
    r.o = 2u;
    return r;
}
__device__ __forceinline__ TS ts_of_tile(uint64_t w, uint32_t len) {
    TS r = ts_identity();
    uint32_t s0 = 0, s1 = 1;
    tile_step(w, len, 0u, r.k0, s0);
    tile_step(w, len, 1u, r.k1, s1);
    r.o = s0 | (s1 << 1);
    return r;
}
// inclusive and exclusive scan of one TS per thread over a 256-thread workgroup (s_w: 4 entries)
__device__ __forceinline__ void ts_scan_256(const TS &mine, TS *s_w, TS &inc, TS &exc) {
    const int lane = lane_id(), wave = wave_id();
    inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const TS p = ts_shfl_up(inc, d);
        if (lane >= d) inc = ts_then(p, inc);
    }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    TS pre = ts_identity();
    for (int w = 0; w < wave; w++) pre = ts_then(pre, s_w[w]);
    TS exl = ts_shfl_up(inc, 1);
    if (lane == 0) exl = ts_identity();
    exc = ts_then(pre, exl);
    inc = ts_then(pre, inc);
}
__global__ void __launch_bounds__(SUP_TILES)
k_tile_sup(const uint64_t *__restrict__ tsum, uint64_t ntiles, const DevState *__restrict__ st, int par,
           TS *__restrict__ ssum) {
    __shared__ TS s_w[SUP_TILES / 64];
    const uint64_t n = st->n[par];
    const uint64_t t = (uint64_t)blockIdx.x * SUP_TILES + threadIdx.x;
    TS mine = ts_identity();
    if (t < ntiles && t * TILE < n) mine = ts_of_tile(tsum[t], (uint32_t)min((uint64_t)TILE, n - t * TILE));
    TS inc, exc;
    ts_scan_256(mine, s_w, inc, exc);
    if (threadIdx.x == SUP_TILES - 1) ssum[blockIdx.x] = inc;
}
__global__ void __launch_bounds__(1024)
k_tile_scan_sup(const TS *__restrict__ ssum, uint64_t nsup, TS *__restrict__ spre, DevState *st, int par, IterRec *rec,
                int iter, const uint32_t *__restrict__ ids, uint32_t *dirty_n) {
    __shared__ TS s_w[16];
    __shared__ uint32_t s_status;
    if (threadIdx.x == 0) {  // (k_tile_scan's bookkeeping)
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
    const uint64_t R = (nsup + 1023) / 1024;
    const uint64_t t0 = min((uint64_t)threadIdx.x * R, nsup), t1 = min(t0 + R, nsup);
    TS mine = ts_identity();
    for (uint64_t t = t0; t < t1; t++) mine = ts_then(mine, ssum[t]);
    const int lane = lane_id(), wave = wave_id();
    TS inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const TS p = ts_shfl_up(inc, d);
        if (lane >= d) inc = ts_then(p, inc);
    }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    TS pre = ts_identity();
    for (int w = 0; w < wave; w++) pre = ts_then(pre, s_w[w]);
    TS exl = ts_shfl_up(inc, 1);
    if (lane == 0) exl = ts_identity();
    pre = ts_then(pre, exl);  // everything before this thread's super-tiles
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
        spre[t] = pre;
        pre = ts_then(pre, ssum[t]);
    }
}
__global__ void __launch_bounds__(SUP_TILES)
k_tile_expand(const uint64_t *__restrict__ tsum, uint64_t ntiles, const DevState *__restrict__ st, int par,
              const TS *__restrict__ spre, uint64_t *__restrict__ tile_off, uint8_t *__restrict__ tile_sin) {
    __shared__ TS s_w[SUP_TILES / 64];
    if (st->status) return;  // (uniform; the scan returned before it wrote the prefixes)
    const uint64_t n = st->n[par];
    const uint64_t t = (uint64_t)blockIdx.x * SUP_TILES + threadIdx.x;
    const bool live = t < ntiles && t * TILE < n;
    TS mine = ts_identity();
    if (live) mine = ts_of_tile(tsum[t], (uint32_t)min((uint64_t)TILE, n - t * TILE));
    TS inc, exc;
    ts_scan_256(mine, s_w, inc, exc);
    if (live) {
        const TS pre = ts_then(spre[blockIdx.x], exc);  // everything before this tile; the stream starts with carry 0
        tile_off[t] = pre.k0;
        tile_sin[t] = (uint8_t)(pre.o & 1u);
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
// HL: layout behind hdr4 -- 0: {first three, last}; 1: a SlotHdr image {first three, (meta),
// second-to-last, last} (k_slots2.hip; hdr4 may then point into LDS).
template <bool DELTA, bool SKIP_UNCHANGED, int HL = 0>
__device__ __forceinline__ void tile_rewrite(const Tile &t, uint32_t s, uint32_t a, uint32_t b,
                                             uint32_t newid, uint32_t *__restrict__ dst_tile,
                                             uint32_t *s_wsum, uint32_t *__restrict__ delta,
                                             uint32_t vcap, int own_len, uint32_t *kept_out,
                                             bool *changed_out, uint32_t *__restrict__ hdr4 = nullptr,
                                             uint32_t *__restrict__ idx = nullptr, uint32_t istride = 0,
                                             uint32_t tself = 0, uint32_t tnext = 0xFFFFFFFFu) {
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
                if (kb[j] && (lo < 3 || (HL == 0 ? hi == total : hi + 1 >= total))) {
                    uint32_t gi = lo;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if ((kb[j] >> k) & 1u) {
                            const uint32_t w = t.x[j][k];
                            const uint32_t ow = ((mb[j] >> k) & 1u) ? (newid | (w & (FLAG | WMASK))) : w;
                            if (gi < 3) hdr4[gi] = ow;
                            if (HL == 0) {
                                if (gi + 1 == total) hdr4[3] = ow;
                            } else {
                                if (gi + 1 == total) hdr4[5] = ow;
                                if (gi + 2 == total) hdr4[4] = ow;
                            }
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
        delta += delta_rep_off(blockIdx.x & (nrep - 1), vcap);
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
                            // idx: the inverted slot index (k_index.hip) learns the pair this creates -- in
                            // the filter of the slot that owns its left element (tself) and, when its right
                            // element is the next slot's, in that one's too (a boundary pair is known to both)
                            const int qpos = qw + j * 256 + k;
                            if (Mk) {
                                const uint32_t y = Mq ? newid : (Xq & IDMASK);
                                atomicAdd(&delta[3 * (size_t)vcap + y], wt);
                                if (idx) {
                                    index_add(idx, istride, tself, newid, y);
                                    if (qpos + 2 >= own_len && tnext != 0xFFFFFFFFu) index_add(idx, istride, tnext, newid, y);
                                }
                            } else if (Mq) {
                                atomicAdd(&delta[2 * (size_t)vcap + (X[k] & IDMASK)], wt);
                                if (idx) {
                                    index_add(idx, istride, tself, X[k] & IDMASK, newid);
                                    if (qpos + 1 >= own_len && tnext != 0xFFFFFFFFu) index_add(idx, istride, tnext, X[k] & IDMASK, newid);
                                }
                            }
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

}  // namespace BPE_G
}  // namespace bpe
