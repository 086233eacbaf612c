// k_common.hip -- wave / block helpers shared by every kernel.
// Part of bpe_kernels.hip, which includes the parts in order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../bpe_device.h"

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
__device__ __forceinline__ uint32_t lane_first(uint32_t x) { return (uint32_t)__builtin_amdgcn_readlane((int)x, 0); }
__device__ __forceinline__ uint32_t lane_last(uint32_t x) { return (uint32_t)__builtin_amdgcn_readlane((int)x, 63); }
// unsigned maximum over the wave, every lane gets it (DPP scan + readlane: VALU speed)
__device__ __forceinline__ uint32_t wave_umax_dpp(uint32_t x) {
    int v = (int)x;
    auto umax = [](int p, int q) { return (int)max((uint32_t)p, (uint32_t)q); };
    v = umax(v, dpp_mov<0x111>(0, v));
    v = umax(v, dpp_mov<0x112>(0, v));
    v = umax(v, dpp_mov<0x114>(0, v));
    v = umax(v, dpp_mov<0x118>(0, v));
    v = umax(v, dpp_mov<0x142, 0xA>(0, v));
    v = umax(v, dpp_mov<0x143, 0xC>(0, v));
    return lane_last((uint32_t)v);
}
__device__ __forceinline__ uint32_t wave_umin_dpp(uint32_t x) { return ~wave_umax_dpp(~x); }
__device__ __forceinline__ uint32_t lane_next(uint32_t x, uint32_t fill) {  // value of lane+1 (lane 63: fill)
    return (uint32_t)dpp_mov<0x130>((int)fill, (int)x);
}


// ---------------------------------------------------------------------------
// Agent-scope accesses for data that crosses workgroups INSIDE one launch (k_step.hip).  The eight XCDs have an L2 each;
// a plain store may sit dirty in the writer's L2 and a plain load may hit a line the reader's L2 fetched earlier, until
// a kernel boundary cleans both -- and a fence that cleans them inside a launch (buffer_wbl2 + buffer_inv) costs 14 us
// per grid barrier with one fencing thread per workgroup, 120 us with all of them (tools/atomic_peak.hip,
// profiles/r6_atomic_peak.json).  A relaxed agent-scope store is written through to the memory side, a relaxed agent-scope
// load is served from there: what is handed over this way needs no cache maintenance, only s_waitcnt(0) before the
// writer signals (grid barrier without fences: 3.8 us).
typedef unsigned long long __attribute__((address_space(1))) g_u64;
typedef uint32_t __attribute__((address_space(1))) g_u32;
__device__ __forceinline__ uint32_t ld_agent(const uint32_t *p) {
    return __hip_atomic_load((g_u32 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_agent64(const unsigned long long *p) {
    return __hip_atomic_load((g_u64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(uint32_t *p, uint32_t v) {
    __hip_atomic_store((g_u32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent64(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store((g_u64 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The records a chain step leaves in pinned HOST memory (IterRec per merge, StepRec per step): every field written through
// with system-scope stores, the sequence word after the others have been acknowledged.  No __threadfence_system(): that
// fence writes back every dirty line of the L2 first -- in k_step the whole merge pass's slots -- to publish 32 bytes.
typedef unsigned long long __attribute__((address_space(1))) g_u64s;
__device__ __forceinline__ void st_system64(void *p, unsigned long long v) {
    __hip_atomic_store((g_u64s *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void iter_rec_put(IterRec *r, int32_t a, int32_t b, uint32_t count, uint32_t status, unsigned long long new_len) {
    static_assert(sizeof(IterRec) == 32, "IterRec: {a, b}, {count, status}, new_len, seq");
    unsigned long long *q = reinterpret_cast<unsigned long long *>(r);
    st_system64(q + 0, (unsigned long long)(uint32_t)a | ((unsigned long long)(uint32_t)b << 32));
    st_system64(q + 1, (unsigned long long)count | ((unsigned long long)status << 32));
    st_system64(q + 2, new_len);
}
__device__ __forceinline__ void iter_rec_seal(IterRec *r, unsigned long long seq) {  // (after s_waitcnt(0))
    st_system64(reinterpret_cast<unsigned long long *>(r) + 3, seq);
}
__device__ __forceinline__ void step_rec_put(StepRec *r, uint32_t first_iter, uint32_t k, uint32_t status, uint32_t pad, unsigned long long new_len) {
    static_assert(sizeof(StepRec) == 32, "StepRec: {first_iter, k}, {status, pad}, new_len, seq");
    unsigned long long *q = reinterpret_cast<unsigned long long *>(r);
    st_system64(q + 0, (unsigned long long)first_iter | ((unsigned long long)k << 32));
    st_system64(q + 1, (unsigned long long)status | ((unsigned long long)pad << 32));
    st_system64(q + 2, new_len);
}
__device__ __forceinline__ void step_rec_seal(StepRec *r, unsigned long long seq) {
    st_system64(reinterpret_cast<unsigned long long *>(r) + 3, seq);
}

// a staged slot header (sparse merge passes): written by the wave that rewrote slot t, read by whoever commits it --
// another workgroup, in k_step (THROUGH) in the same launch: only then written through and read at agent scope.  (Written
// through in every sweep the four 8-byte stores per changed slot cost the mid-training sweeps 40-70 %: they queue at the
// memory side with the sites' atomics -- profiles/r6_notes.md.)
template <bool THROUGH>
__device__ __forceinline__ void stage_put(StageRec *r, uint32_t t, const uint32_t (&h)[8]) {
    r->t = t;
    if (THROUGH) {
        unsigned long long *q = reinterpret_cast<unsigned long long *>(r->h);
#pragma unroll
        for (int i = 0; i < 4; i++) st_agent64(q + i, (unsigned long long)h[2 * i] | ((unsigned long long)h[2 * i + 1] << 32));
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) r->h[i] = h[i];
    }
}
template <bool THROUGH>
__device__ __forceinline__ void stage_get(const StageRec *r, uint32_t (&h)[8]) {
    if (THROUGH) {
        const unsigned long long *q = reinterpret_cast<const unsigned long long *>(r->h);
        unsigned long long v[4];
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = ld_agent64(q + i);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            h[2 * i] = (uint32_t)v[i];
            h[2 * i + 1] = (uint32_t)(v[i] >> 32);
        }
    } else {
        const uint4 lo = *reinterpret_cast<const uint4 *>(r->h), hi = *reinterpret_cast<const uint4 *>(r->h + 4);
        h[0] = lo.x; h[1] = lo.y; h[2] = lo.z; h[3] = lo.w;
        h[4] = hi.x; h[5] = hi.y; h[6] = hi.z; h[7] = hi.w;
    }
}

}  // namespace bpe
