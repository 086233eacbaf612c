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

}  // namespace bpe
