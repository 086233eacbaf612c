// k_util.hip -- single-step API utilities and state initialisation.
// Part of bpe_kernels.hip, which includes the parts in order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../bpe_device.h"
#include "k_common.hip"

namespace bpe {

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
    st->sel_tie = 0;
    st->sel_done = 0;
    st->adj = 0;
    st->ncand = 0;
    st->gap = 0;
    st->defer = 0;
    st->scan_a = st->scan_b = st->scan_z = 0xFFFFFFFFu;
    st->chain_n = st->chain_pos = st->chain_cut = st->chain_taken = 0;
    st->sel_ran = 0;
    st->iter = 0;
    st->num_merges = 0;
    st->sel_mode = CH_FULL;
    st->tl_n = st->tl_M = st->tl_skip = 0;
    st->bk = st->bz0 = 0;
}

}  // namespace bpe
