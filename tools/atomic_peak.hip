// atomic_peak.hip -- TIMED ceilings for what bounds the sparse merge pass ("roofline_atomics"):
//   (a) device-scope 4-byte atomics per second, addresses spread over every memory channel (the delta words and index
//       bits of a merge site), non-returning and returning, add and or;
//   (b) the same with workgroup scope on addresses private to a workgroup (does the scope change where an atomic
//       executes on this part?);
//   (c) a hot word behind 16 / 256 replicas (the delta vectors' replica blocks);
//   (d) what a hand-over costs: an empty launch, a dependent chain of empty launches, a grid barrier (one atomic + a
//       poll per workgroup, 256 resident workgroups of 1024 threads) inside one launch.
// hipEvents around `reps` launches; prints one JSON line.  No profiler needed.
// Build + run: hipcc --offload-arch=gfx950 -O3 tools/atomic_peak.hip -o tools/atomic_peak && tools/atomic_peak
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHK(x)                                                          \
    do {                                                                \
        hipError_t e_ = (x);                                            \
        if (e_ != hipSuccess) {                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));     \
            exit(1);                                                    \
        }                                                               \
    } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x *= 0x9E3779B1u;
    x ^= x >> 15;
    x *= 0x85EBCA77u;
    x ^= x >> 13;
    return x;
}

// MODE 0: atomicAdd agent, no return | 1: atomicAdd agent, returning | 2: atomicOr agent, no return
// MODE 3: atomicAdd workgroup scope, addresses inside the workgroup's own 1/grid share of the buffer
// MODE 4: atomicAdd agent on ONE word behind `nrep` replicas 256 bytes + one skewed line apart
template <int MODE>
__global__ void __launch_bounds__(1024) k_atomics(uint32_t *__restrict__ buf, uint32_t mask, uint32_t per_thread, uint32_t nrep,
                                                   uint32_t *sink) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < per_thread; i++) {
        const uint32_t r = mix(gid * 1315423911u + i * 2654435761u + 17u);
        if (MODE == 0) {
            atomicAdd(&buf[r & mask], 1u);
        } else if (MODE == 1) {
            acc += atomicAdd(&buf[r & mask], 1u);
        } else if (MODE == 2) {
            atomicOr(&buf[r & mask], 1u << (r >> 27));
        } else if (MODE == 3) {
            const uint32_t share = (mask + 1u) / gridDim.x;  // words per workgroup (a power of two when the grid is)
            __hip_atomic_fetch_add(&buf[blockIdx.x * share + (r & (share - 1u))], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            atomicAdd(&buf[(size_t)(r & (nrep - 1u)) * (4u * 32768u + 64u)], 1u);
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}

__global__ void k_empty(uint32_t *p) {
    if (p == nullptr) return;
}
__global__ void __launch_bounds__(1024) k_empty_big(uint32_t *p) {
    if (p == nullptr) return;
}
// a chain of dependent hand-overs inside ONE launch: `rounds` grid barriers (monotone counter, bounded spin).
// FENCE 0: every thread __threadfence() before and after (L2 write-back + invalidate by all 16 waves of every workgroup)
//       1: one thread per workgroup fences (release before its arrival, acquire after the wait)
//       2: no cache maintenance at all: s_waitcnt(0) + relaxed agent-scope atomics (what crosses workgroups then has to be
//          written and read with agent-scope accesses itself)
template <int FENCE>
__global__ void __launch_bounds__(1024) k_barriers(uint32_t *ctr, uint32_t rounds, uint32_t base, uint32_t *fail) {
    for (uint32_t r = 0; r < rounds; r++) {
        if (FENCE == 0) __threadfence();
        if (FENCE == 2) __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t target = base + (r + 1u) * gridDim.x;
            if (FENCE == 1) __threadfence();
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t spins = 0;
            while ((int32_t)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
                if (++spins > (1u << 22)) {
                    *fail = 1;
                    break;
                }
            }
            if (FENCE == 1) __threadfence();
        }
        __syncthreads();
        if (FENCE == 0) __threadfence();
    }
}

static float time_ms(hipEvent_t a, hipEvent_t b) {
    float ms = 0;
    CHK(hipEventSynchronize(b));
    CHK(hipEventElapsedTime(&ms, a, b));
    return ms;
}

int main() {
    const size_t words = 1ull << 28;  // 1 GiB of counters: every channel, far beyond the caches
    uint32_t *buf, *sink, *ctr, *fail;
    CHK(hipMalloc(&buf, words * 4));
    CHK(hipMalloc(&sink, 64));
    CHK(hipMalloc(&ctr, 64));
    CHK(hipMalloc(&fail, 64));
    CHK(hipMemset(buf, 0, words * 4));
    CHK(hipMemset(ctr, 0, 64));
    CHK(hipMemset(fail, 0, 64));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int grid = 256;
    printf("{\"device\": \"%s\", \"cus\": %d", prop.gcnArchName, prop.multiProcessorCount);
    const uint32_t per = 256;
    const double total = (double)grid * 1024 * per;
    auto run = [&](const char *name, auto kern, uint32_t mask, uint32_t nrep) {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), 0, 0, buf, mask, per, nrep, sink);  // warm
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0, 0));
        const int reps = 5;
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), 0, 0, buf, mask, per, nrep, sink);
        CHK(hipEventRecord(e1, 0));
        const float ms = time_ms(e0, e1);
        printf(", \"%s\": {\"atomics\": %.0f, \"ms_per_launch\": %.4f, \"G_per_s\": %.3f}", name, total, ms / reps,
               total * reps / (ms * 1e-3) / 1e9);
    };
    const uint32_t full = (uint32_t)(words - 1);
    run("add_agent_spread_1GiB", k_atomics<0>, full, 0);
    run("add_agent_returning_spread_1GiB", k_atomics<1>, full, 0);
    run("or_agent_spread_1GiB", k_atomics<2>, full, 0);
    run("add_agent_spread_64MiB", k_atomics<0>, (1u << 24) - 1u, 0);
    run("add_agent_spread_4MiB", k_atomics<0>, (1u << 20) - 1u, 0);
    run("add_workgroup_scope_private_share_1GiB", k_atomics<3>, full, 0);
    run("add_workgroup_scope_private_share_64MiB", k_atomics<3>, (1u << 24) - 1u, 0);
    run("add_agent_hot_word_16_replicas", k_atomics<4>, full, 16);
    run("add_agent_hot_word_256_replicas", k_atomics<4>, full, 256);
    // ---- hand-overs --------------------------------------------------------------------------------------------------
    auto launches = [&](const char *name, bool big, int n) {
        for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0, sink);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0, 0));
        for (int i = 0; i < n; i++) {
            if (big) hipLaunchKernelGGL(k_empty_big, dim3(grid), dim3(1024), 0, 0, sink);
            else hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0, sink);
        }
        CHK(hipEventRecord(e1, 0));
        const float ms = time_ms(e0, e1);
        printf(", \"%s\": {\"launches\": %d, \"us_per_launch\": %.3f}", name, n, ms * 1e3 / n);
    };
    launches("empty_launch_1x64_back_to_back", false, 2000);
    launches("empty_launch_256x1024_back_to_back", true, 2000);
    uint32_t base = 0;
    auto barriers = [&](const char *name, auto kern) {
        const uint32_t rounds = 200;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), 0, 0, ctr, rounds, base, fail);
        base += rounds * grid;
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), 0, 0, ctr, rounds, base, fail);
        base += rounds * grid;
        CHK(hipEventRecord(e1, 0));
        const float ms = time_ms(e0, e1);
        uint32_t f = 0;
        CHK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
        printf(", \"%s\": {\"rounds\": %u, \"us_per_barrier\": %.3f, \"timed_out\": %u}", name, rounds, ms * 1e3 / rounds, f);
    };
    barriers("grid_barrier_256x1024_every_thread_fences", k_barriers<0>);
    barriers("grid_barrier_256x1024_one_thread_fences", k_barriers<1>);
    barriers("grid_barrier_256x1024_no_cache_maintenance", k_barriers<2>);
    printf("}\n");
    return 0;
}
