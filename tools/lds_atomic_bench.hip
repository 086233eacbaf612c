// lds_atomic_bench.hip -- what an LDS atomic costs on gfx950, by address pattern (one 1024-thread workgroup per CU,
// a 128 KiB table of 32-bit words, R atomics per thread): the byte-pair histogram (k_pair_count_bytes, k_load_count)
// is bound by this rate.  Build: hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_bench.hip -o tools/lds_atomic_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int R = 2048;
__device__ __forceinline__ uint32_t rnd(uint32_t &s) { s = s * 1664525u + 1013904223u; return s ^ (s >> 15); }
template <int MODE>
__global__ void __launch_bounds__(1024) k_bench(uint32_t *out) {
    extern __shared__ uint32_t s[];
    for (int i = threadIdx.x; i < 32768; i += 1024) s[i] = 0;
    __syncthreads();
    uint32_t seed = blockIdx.x * 1024u + threadIdx.x + 12345u;
    uint32_t acc = 0;
    for (int it = 0; it < R; it++) {
        const uint32_t r = rnd(seed);
        uint32_t idx;
        if (MODE == 0) idx = threadIdx.x;                                   // every lane its own word, consecutive
        else if (MODE == 1 || MODE == 5 || MODE == 6 || MODE == 7) idx = r & 32767u;  // uniform over the table
        else if (MODE == 2) idx = 0;                                        // one word
        else if (MODE == 3) idx = (r & 63u) * 33u;                          // 64 hot words on distinct banks
        else if (MODE == 4) {                                               // skewed: idx ~ 32768 u^4
            const float u = (float)(r & 0xFFFFFFu) * (1.0f / 16777216.0f);
            idx = (uint32_t)(32767.0f * u * u * u * u);
        } else if (MODE == 8) idx = (threadIdx.x & 63u) * 2u + (r & 0x7F80u);  // distinct banks? (64 lanes, stride 2 words)
        else idx = ((threadIdx.x & 63u) + (r & 0x7FC0u)) & 32767u;            // MODE 9: lane l -> word base + l (conflict-free row)
        if (MODE == 5) { const uint32_t v = s[idx]; s[idx] = v + 1; }
        else if (MODE == 6) acc += atomicAdd(&s[idx], 1u);
        else if (MODE == 7) atomicAdd(reinterpret_cast<unsigned long long *>(s) + (idx >> 1), 1ull);
        else atomicAdd(&s[idx], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = s[0] + s[33] + acc;
}
template <int MODE>
void run(const char *what, uint32_t *out, int cus) {
    CHK(hipFuncSetAttribute((const void *)k_bench<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t a, b;
    CHK(hipEventCreate(&a));
    CHK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_bench<MODE>, dim3(cus), dim3(1024), 131072, 0, out);
    CHK(hipEventRecord(a, 0));
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k_bench<MODE>, dim3(cus), dim3(1024), 131072, 0, out);
    CHK(hipEventRecord(b, 0));
    CHK(hipEventSynchronize(b));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, a, b));
    ms /= 5;
    const double waveops_per_cu = 16.0 * R;  // 16 waves x R
    printf("{\"mode\": %d, \"pattern\": \"%s\", \"ms\": %.4f, \"ns_per_wave_op_per_CU\": %.2f, \"G_lane_atomics_per_s_chip\": %.1f}\n", MODE, what, ms,
           ms * 1e6 / waveops_per_cu, (double)cus * 1024.0 * R / (ms * 1e-3) / 1e9);
}
int main() {
    CHK(hipSetDevice(0));
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    uint32_t *out;
    CHK(hipMalloc((void **)&out, 4096 * 4));
    printf("{\"cus\": %d, \"clock_MHz\": %d, \"R_per_thread\": %d}\n", cus, p.clockRate / 1000, R);
    run<0>("ds_add_u32, lane l -> word l (consecutive)", out, cus);
    run<9>("ds_add_u32, lane l -> word base + l, random base per lane row (64 consecutive words of a random 64-word row... per lane random row)", out, cus);
    run<8>("ds_add_u32, lane l -> word 2l + random 128-word row", out, cus);
    run<1>("ds_add_u32, uniform random over 32768 words", out, cus);
    run<4>("ds_add_u32, skewed (idx ~ 32768 u^4)", out, cus);
    run<3>("ds_add_u32, 64 hot words", out, cus);
    run<2>("ds_add_u32, one word", out, cus);
    run<6>("ds_add_rtn_u32, uniform random", out, cus);
    run<7>("ds_add_u64, uniform random over 16384 words", out, cus);
    run<5>("ds_read + ds_write (no atomic), uniform random", out, cus);
    return 0;
}
