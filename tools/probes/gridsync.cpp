// Probe: what does a kernel boundary cost against an in-kernel grid barrier on MI355X, for a chain of small dependent
// layers (256 workgroups, one per CU, each reading 16 KB another workgroup wrote in the previous layer)?
//   (a) N separate launches on one stream            (b) one cooperative launch, cooperative_groups grid.sync()
//   (c) one cooperative launch, hand-written barrier (agent-scope release add + acquire spin, sense by generation)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/gridsync.cpp -o tools/probes/gridsync.bin
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <vector>
namespace cg = cooperative_groups;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int WG = 256, PER = 4096;  // floats per workgroup and layer (16 KB)

__device__ __forceinline__ void layer(const float* __restrict__ src, float* __restrict__ dst, int l, int nwg) {
    extern __shared__ float lds[];
    const int from = (blockIdx.x * 37 + l * 11 + 5) % nwg;  // a block written by some other workgroup (other XCD) one layer ago
    const float4* s = reinterpret_cast<const float4*>(src + (size_t)from * PER);
    float4* d = reinterpret_cast<float4*>(dst + (size_t)blockIdx.x * PER);
#pragma unroll
    for (int k = 0; k < PER / 4 / WG; ++k) {
        float4 v = s[threadIdx.x + k * WG];
        v.x = v.x * 0.5f + 1.f; v.y = v.y * 0.5f + 1.f; v.z = v.z * 0.5f + 1.f; v.w = v.w * 0.5f + 1.f;
        lds[threadIdx.x] = v.x;
        d[threadIdx.x + k * WG] = v;
    }
}
__global__ __launch_bounds__(WG) void k_one(const float* src, float* dst, int l, int nwg) { layer(src, dst, l, nwg); }
__global__ __launch_bounds__(WG) void k_coop(float* a, float* b, int layers, int nwg) {
    cg::grid_group g = cg::this_grid();
    for (int l = 0; l < layers; ++l) {
        layer((l & 1) ? b : a, (l & 1) ? a : b, l, nwg);
        g.sync();
    }
}
__device__ __forceinline__ void my_barrier(unsigned* ctr, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}
__global__ __launch_bounds__(WG) void k_mine(float* a, float* b, int layers, int nwg, unsigned* ctr) {
    for (int l = 0; l < layers; ++l) {
        layer((l & 1) ? b : a, (l & 1) ? a : b, l, nwg);
        my_barrier(ctr, (unsigned)(l + 1) * gridDim.x);
    }
}

int main() {
    const int nwg = 256, layers = 64, smem = 100 * 1024;
    float *a, *b; unsigned* ctr;
    CHK(hipMalloc(&a, (size_t)nwg * PER * 4)); CHK(hipMalloc(&b, (size_t)nwg * PER * 4)); CHK(hipMalloc(&ctr, 4));
    CHK(hipMemset(a, 0, (size_t)nwg * PER * 4)); CHK(hipMemset(b, 0, (size_t)nwg * PER * 4));
    CHK(hipFuncSetAttribute((const void*)k_one, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    CHK(hipFuncSetAttribute((const void*)k_coop, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    CHK(hipFuncSetAttribute((const void*)k_mine, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    hipStream_t s; CHK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float ms;
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipEventRecord(e0, s));
        for (int l = 0; l < layers; ++l) hipLaunchKernelGGL(k_one, dim3(nwg), dim3(WG), smem, s, (l & 1) ? b : a, (l & 1) ? a : b, l, nwg);
        CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("(a) separate launches      : %.2f us per layer\n", ms * 1e3 / layers);
        int L = layers, N = nwg;
        void* args[] = {&a, &b, &L, &N};
        CHK(hipEventRecord(e0, s));
        CHK(hipLaunchCooperativeKernel((const void*)k_coop, dim3(nwg), dim3(WG), args, smem, s));
        CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("(b) cooperative grid.sync(): %.2f us per layer\n", ms * 1e3 / layers);
        CHK(hipMemsetAsync(ctr, 0, 4, s));
        void* args2[] = {&a, &b, &L, &N, &ctr};
        CHK(hipEventRecord(e0, s));
        CHK(hipLaunchCooperativeKernel((const void*)k_mine, dim3(nwg), dim3(WG), args2, smem, s));
        CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("(c) hand-written barrier   : %.2f us per layer\n", ms * 1e3 / layers);
    }
    // graph capture of a cooperative launch?
    hipGraph_t gr = nullptr; hipGraphExec_t ge = nullptr;
    hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    if (e == hipSuccess) {
        int L = layers, N = nwg;
        void* args[] = {&a, &b, &L, &N};
        hipError_t el = hipLaunchCooperativeKernel((const void*)k_coop, dim3(nwg), dim3(WG), args, smem, s);
        hipError_t ee = hipStreamEndCapture(s, &gr);
        hipError_t ei = (ee == hipSuccess && gr) ? hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0) : hipErrorUnknown;
        printf("capture of a cooperative launch: launch=%s end=%s instantiate=%s\n", hipGetErrorString(el), hipGetErrorString(ee), hipGetErrorString(ei));
        if (ei == hipSuccess) {
            CHK(hipEventRecord(e0, s)); CHK(hipGraphLaunch(ge, s)); CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1));
            CHK(hipEventElapsedTime(&ms, e0, e1));
            printf("(d) graph replay of (b)    : %.2f us per layer\n", ms * 1e3 / layers);
        }
    }
    (void)hipGetLastError();
    printf("done\n");
    return 0;
}
