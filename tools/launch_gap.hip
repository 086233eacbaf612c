// launch_gap.hip -- what does a dependent launch cost, by kernel-argument size and static LDS?  (k_step showed 5.7 us
// between two launches of itself where the three-launch sequence shows 0.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int N> struct Args { uint32_t *p; uint32_t v[N]; };
template <int N, int LDS> __global__ void __launch_bounds__(1024) k(Args<N> a) {
    __shared__ uint32_t s[LDS / 4 > 0 ? LDS / 4 : 1];
    if (a.p == nullptr) { s[threadIdx.x % (LDS / 4 > 0 ? LDS / 4 : 1)] = a.v[0]; __syncthreads(); a.p[0] = s[0]; }
}
template <int N, int LDS> int run(const char *name, uint32_t *d) {
    Args<N> a; a.p = d; for (int i = 0; i < N; i++) a.v[i] = i;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL((k<N, LDS>), dim3(256), dim3(1024), 0, 0, a);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0, 0));
    const int n = 3000;
    for (int i = 0; i < n; i++) hipLaunchKernelGGL((k<N, LDS>), dim3(256), dim3(1024), 0, 0, a);
    CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s\"%s\": %.3f", name[0] == '!' ? "" : ", ", name[0] == '!' ? name + 1 : name, ms * 1e3 / n);
    return 0;
}
int main() {
    uint32_t *d; CHK(hipMalloc(&d, 64));
    printf("{\"unit\": \"us per back-to-back launch, 256 x 1024 threads\", ");
    run<1, 0>("!args_16B_lds_0", d);
    run<14, 0>("args_64B_lds_0", d);
    run<62, 0>("args_256B_lds_0", d);
    run<110, 0>("args_448B_lds_0", d);
    run<254, 0>("args_1024B_lds_0", d);
    run<1, 34000>("args_16B_lds_34KB", d);
    run<1, 86000>("args_16B_lds_86KB", d);
    run<110, 86000>("args_448B_lds_86KB", d);
    printf("}\n");
    return 0;
}
