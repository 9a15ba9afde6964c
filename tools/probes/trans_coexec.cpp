// Does v_exp_f32 (quarter-rate transcendental) co-execute with full-rate VALU work of the same wave / SIMD on gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/trans_coexec.cpp -o tools/probes/trans_coexec.bin
// Three kernels, 256 threads x (CUs x 8) workgroups, 8 independent chains per thread: exp only, fma only, both interleaved.
// If the transcendental unit is a pipe of its own, "both" costs max(exp, fma); if it shares the VALU issue, the sum.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK_(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int MODE, int FMA_PER_EXP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float e[8], f[8];
    for (int i = 0; i < 8; ++i) { e[i] = seed + threadIdx.x * 1e-6f + i; f[i] = seed * 0.5f + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE & 1) e[i] = __builtin_amdgcn_exp2f(e[i] * 1e-3f);
            if (MODE & 2) {
#pragma unroll
                for (int r = 0; r < FMA_PER_EXP; ++r) f[i] = fmaf(f[i], 1.0000001f, 1e-7f);
            }
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += e[i] + f[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
template <int MODE, int FPE>
float run(float* d, int grid, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE, FPE>), dim3(grid), dim3(256), 0, 0, d, iters, 1.5f);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((k<MODE, FPE>), dim3(grid), dim3(256), 0, 0, d, iters, 1.5f);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    float* d; CK_(hipMalloc(&d, 4096));
    const int grid = 256 * 8, iters = 4096;
    const double nexp = (double)grid * 256 * iters * 8;
    float t;
    t = run<1, 1>(d, grid, iters); printf("exp only                     %8.3f ms  %6.2f cycles per wave-instruction per SIMD (2.4 GHz)\n", t, t * 1e-3 * 2.4e9 / (nexp / 64 / 1024));
    t = run<2, 1>(d, grid, iters); printf("fma only (1 per slot)        %8.3f ms  %6.2f\n", t, t * 1e-3 * 2.4e9 / (nexp / 64 / 1024));
    t = run<3, 1>(d, grid, iters); printf("exp + 1 fma interleaved      %8.3f ms  %6.2f\n", t, t * 1e-3 * 2.4e9 / (nexp / 64 / 1024));
    t = run<2, 2>(d, grid, iters); printf("fma only (2 per slot)        %8.3f ms  %6.2f\n", t, t * 1e-3 * 2.4e9 / (nexp / 64 / 1024));
    t = run<3, 2>(d, grid, iters); printf("exp + 2 fma interleaved      %8.3f ms  %6.2f\n", t, t * 1e-3 * 2.4e9 / (nexp / 64 / 1024));
    t = run<2, 4>(d, grid, iters); printf("fma only (4 per slot)        %8.3f ms  %6.2f\n", t, t * 1e-3 * 2.4e9 / (nexp / 64 / 1024));
    t = run<3, 4>(d, grid, iters); printf("exp + 4 fma interleaved      %8.3f ms  %6.2f\n", t, t * 1e-3 * 2.4e9 / (nexp / 64 / 1024));
    return 0;
}
