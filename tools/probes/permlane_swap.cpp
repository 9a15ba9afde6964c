// Probe: what v_permlane16_swap_b32 (gfx950) does to a wave -- the row all-reduce of kernels_lafuse.h relies on it.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/permlane_swap.cpp -o tools/probes/permlane_swap.bin
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
    const unsigned v = threadIdx.x;
    const auto r = __builtin_amdgcn_permlane16_swap(v, v + 100u, false, false);
    o[threadIdx.x] = r[0];
    o[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 512);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("first operand = lane, second = lane + 100\nr[0]:"); for (int i = 0; i < 64; ++i) printf(" %u", h[i]);
    printf("\nr[1]:"); for (int i = 0; i < 64; ++i) printf(" %u", h[64 + i]);
    printf("\n");
    return 0;
}
