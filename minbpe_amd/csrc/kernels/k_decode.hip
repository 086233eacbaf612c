// k_decode.hip -- batch decode.
// Part of bpe_kernels.hip, which includes the parts in order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../bpe_device.h"
#include "k_common.hip"

namespace bpe {

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

}  // namespace bpe
