// Does kernarg preloading (the CP writes the first kernel arguments into SGPRs before the wave starts: -mllvm -amdgpu-kernarg-preload-count=N) shorten a chain of small
// dependent launches on gfx950?  Build twice (with / without the flag) and compare:
//   hipcc --offload-arch=gfx950 -O3 [-mllvm -amdgpu-kernarg-preload-count=8] tools/probes/kernarg_preload.cpp -o ...
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void hop(const float* a, float* b, int n, int stride) {
    const int i = (blockIdx.x * 256 + threadIdx.x) * stride;
    if (i < n) b[i] = a[i] + 1.0f;
}
int main() {
    const int n = 256 * 256;
    float *x, *y; hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMemset(x, 0, n * 4); hipMemset(y, 0, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int k = 0; k < 500; ++k) { hipLaunchKernelGGL(hop, dim3(256), dim3(256), 0, 0, (const float*)x, y, n, 1); hipLaunchKernelGGL(hop, dim3(256), dim3(256), 0, 0, (const float*)y, x, n, 1); }
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("chain of 1000 dependent launches: %.3f us per launch\n", ms);
    }
    return 0;
}
