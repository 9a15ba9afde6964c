// Probe: the price of a SMALL-GROUP barrier inside a persistent kernel on MI355X -- what a per-sample resident kernel for the 8x8 / 16x16 levels would
// pay per layer instead of a kernel boundary.  256 workgroups (one per CU); groups of 4 workgroups hand 16 KB each to one another between "layers"
// (each workgroup reads the block its group neighbour wrote one layer ago), with one counter per group on its own cache line.
//   (a) separate launches, one per layer (the present design)
//   (b) group = 4 consecutive workgroup ids          (round-robin dispatch: four different XCDs)      agent-scope release add + acquire spin
//   (c) group = ids b, b+8, b+16, b+24               (round-robin dispatch: the SAME XCD, checked with XCC_ID)   same barrier
//   (e) as (c) with the light same-XCD protocol (no L2 write-back; see k_group)
//   (d) as (c), whole-grid barrier instead (all 256 on one counter) for reference (= gridsync.cpp (c))
// Results are checked against (a).     hipcc --offload-arch=gfx950 -O3 tools/probes/groupsync.cpp -o tools/probes/groupsync.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int WG = 256, PER = 4096;  // floats per workgroup and layer (16 KB)

__device__ __forceinline__ int partner(int b, int mode) {  // the group neighbour whose block this workgroup reads
    if (mode == 0) return (b & ~3) | ((b + 1) & 3);                      // consecutive ids
    const int base = (b & 7) | ((b >> 5) << 5), k = (b >> 3) & 3;        // ids base + 8 k, k = 0..3
    return base + 8 * ((k + 1) & 3);
}
__device__ __forceinline__ int group_of(int b, int mode) { return mode == 0 ? (b >> 2) : ((b & 7) | ((b >> 5) << 3)); }

__device__ __forceinline__ void layer(const float* __restrict__ src, float* __restrict__ dst, int from) {
    const float4* s = reinterpret_cast<const float4*>(src + (size_t)from * PER);
    float4* d = reinterpret_cast<float4*>(dst + (size_t)blockIdx.x * PER);
#pragma unroll
    for (int k = 0; k < PER / 4 / WG; ++k) {
        float4 v = s[threadIdx.x + k * WG];
        v.x = v.x * 0.5f + 1.f; v.y = v.y * 0.5f + 2.f; v.z = v.z * 0.5f + 3.f; v.w = v.w * 0.5f + (float)(blockIdx.x & 7);
        d[threadIdx.x + k * WG] = v;
    }
}
__global__ __launch_bounds__(WG) void k_one(const float* src, float* dst, int mode) { layer(src, dst, partner(blockIdx.x, mode)); }

// light = 1: the same-XCD protocol -- no L2 write-back (the group shares its XCD's L2): stores drained (vmcnt 0, i.e. acknowledged by L2), RELAXED add;
// RELAXED spin, then the reader's vector L1 invalidated (buffer_inv sc1).  Only meaningful for mode 1 groups.
__global__ __launch_bounds__(WG) void k_group(float* a, float* b, int layers, int mode, unsigned* ctr, int gsize, unsigned* xcc, int light) {
    extern __shared__ float lds[];
    if (threadIdx.x == 0 && xcc) xcc[blockIdx.x] = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | ((4 - 1) << 11));
    unsigned* c = ctr + (gsize == 4 ? group_of(blockIdx.x, mode) : 0) * 32;  // 128-byte lines
    const int from = partner(blockIdx.x, mode);
    for (int l = 0; l < layers; ++l) {
        layer((l & 1) ? b : a, (l & 1) ? a : b, from);
        __syncthreads();
        const unsigned target = (unsigned)(l + 1) * (unsigned)gsize;
        if (light) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its stores are in L2
            __syncthreads();
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
            }
            __syncthreads();
            asm volatile("buffer_inv sc1" ::: "memory");  // every wave: drop this CU's L1 lines
        } else {
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(c, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
            }
            __syncthreads();
        }
    }
    lds[threadIdx.x] = 0.f;
}

int main() {
    const int nwg = 256, layers = 64, smem = 100 * 1024;  // 100 KB of LDS: one workgroup per CU, all co-resident
    float *a, *b, *ra; unsigned *ctr, *xcc;
    CHK(hipMalloc(&a, (size_t)nwg * PER * 4)); CHK(hipMalloc(&b, (size_t)nwg * PER * 4)); CHK(hipMalloc(&ra, (size_t)nwg * PER * 4));
    CHK(hipMalloc(&ctr, 64 * 128)); CHK(hipMalloc(&xcc, nwg * 4));
    CHK(hipFuncSetAttribute((const void*)k_one, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    CHK(hipFuncSetAttribute((const void*)k_group, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    hipStream_t s; CHK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float ms;
    std::vector<float> ref((size_t)nwg * PER), got((size_t)nwg * PER);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            CHK(hipMemsetAsync(a, 0, (size_t)nwg * PER * 4, s)); CHK(hipMemsetAsync(b, 0, (size_t)nwg * PER * 4, s));
            CHK(hipEventRecord(e0, s));
            for (int l = 0; l < layers; ++l) hipLaunchKernelGGL(k_one, dim3(nwg), dim3(WG), smem, s, (l & 1) ? b : a, (l & 1) ? a : b, mode);
            CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
            printf("mode %d (a) separate launches          : %.2f us per layer\n", mode, ms * 1e3 / layers);
            CHK(hipMemcpy(ref.data(), a, ref.size() * 4, hipMemcpyDeviceToHost));
            for (int var = 0; var < (mode ? 3 : 2); ++var) {
                int gsize = var == 1 ? 256 : 4, light = var == 2;
                CHK(hipMemsetAsync(a, 0, (size_t)nwg * PER * 4, s)); CHK(hipMemsetAsync(b, 0, (size_t)nwg * PER * 4, s));
                CHK(hipMemsetAsync(ctr, 0, 64 * 128, s));
                int L = layers;
                void* args[] = {&a, &b, &L, &mode, &ctr, &gsize, &xcc, &light};
                CHK(hipEventRecord(e0, s));
                CHK(hipLaunchCooperativeKernel((const void*)k_group, dim3(nwg), dim3(WG), args, smem, s));
                CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
                CHK(hipMemcpy(got.data(), a, got.size() * 4, hipMemcpyDeviceToHost));
                size_t bad = 0;
                for (size_t i = 0; i < got.size(); ++i) bad += got[i] != ref[i];
                printf("mode %d (%s) %s barrier, %s: %.2f us per layer   mismatches vs (a): %zu\n", mode, light ? "e" : (gsize == 4 ? (mode ? "c" : "b") : "d"),
                       gsize == 4 ? "4-workgroup group" : "whole-grid       ", light ? "same-XCD light protocol" : "agent scope", ms * 1e3 / layers, bad);
            }
        }
    }
    std::vector<unsigned> hx(nwg);
    CHK(hipMemcpy(hx.data(), xcc, nwg * 4, hipMemcpyDeviceToHost));
    printf("XCC_ID of workgroups 0..15: ");
    for (int i = 0; i < 16; ++i) printf("%u ", hx[i]);
    int same = 0;
    for (int g = 0; g < 64; ++g) {
        const int base = (g & 7) | ((g >> 3) << 5);
        same += hx[base] == hx[base + 8] && hx[base] == hx[base + 16] && hx[base] == hx[base + 24];
    }
    printf("\nmode-1 groups whose four workgroups report one XCC_ID: %d of 64\n", same);
    return 0;
}
