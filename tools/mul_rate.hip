// VALU issue rate of v_mul_lo_u32 against v_mul_u32_u24 and v_add_u32 on gfx950 (is a 32-bit integer multiply a
// quarter-rate instruction on this part?): 256 x 1024 threads, 8 independent chains per thread, 4096 rounds.
// hipcc --offload-arch=gfx950 -O3 tools/mul_rate.hip -o tools/mul_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int OP>
__global__ void __launch_bounds__(1024) k(uint32_t *out, uint32_t m, int rounds) {
    uint32_t v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = threadIdx.x * 8 + i + blockIdx.x;
    for (int r = 0; r < rounds; r++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "s"(m));
            if (OP == 1) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[i]) : "s"(m));
            if (OP == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "s"(m));
            if (OP == 3) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(v[i]) : "s"(m));
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
}
int main() {
    uint32_t *d;
    hipMalloc(&d, 256 * 1024 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char *names[4] = {"v_mul_lo_u32", "v_mul_u32_u24", "v_add_u32", "v_mad_u32_u24"};
    printf("{");
    for (int op = 0; op < 4; op++) {
        float best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (op == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 0, 0, d, 12345u, 4096);
            if (op == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 0, 0, d, 12345u, 4096);
            if (op == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(1024), 0, 0, d, 12345u, 4096);
            if (op == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(1024), 0, 0, d, 12345u, 4096);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        // wave-instructions per SIMD: 4 waves x 8 x 4096; cycles per wave-instruction per SIMD at 2.4 GHz
        const double inst = 4.0 * 8 * 4096;
        printf("%s\"%s\": {\"ms\": %.4f, \"ns_per_wave_instruction_per_simd\": %.3f}", op ? ", " : "", names[op], best, best * 1e6 / inst);
    }
    printf("}\n");
    return 0;
}
