// k_load.hip -- K0: bytes -> id stream, chunk starts, chunk weights.
// Part of bpe_kernels.hip, which includes the parts in order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../bpe_device.h"
#include "k_common.hip"

namespace bpe {

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

}  // namespace bpe
