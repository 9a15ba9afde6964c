// Probe (round 5): a barrier among the workgroups of ONE XCD that never leaves the XCD's L2 -- against a kernel boundary.
// tools/probes/groupsync.cpp (round 4) measured 8.5 us per layer for its "light" same-XCD hand-off, but its counter traffic was still AGENT-scope
// (sc1 atomics and polling loads are resolved behind the L2, at the fabric: > 1 us per round trip).  Here the counter is touched with WORKGROUP-scope
// operations only: the atomic add executes in the XCD's own L2 (all participants sit behind that L2: XCC_ID is checked), the polling load carries sc0
// (misses the vector L1), stores are drained with vmcnt(0) before the add and the readers drop their vector L1 afterwards.
//   256 workgroups (one per CU, 100 KB of LDS); per "layer" every workgroup reads the 16 KB a neighbour ON ITS XCD wrote one layer ago and writes 16 KB.
//   (a) separate launches, one per layer
//   (f) groups of 4 workgroups of one XCD (ids b, b+8, b+16, b+24), XCD-local protocol
//   (g) groups of 32 = all workgroups of one XCD, XCD-local protocol           [what a fused chain of layers over an XCD-contiguous work partition needs]
//   poll = atomic add of 0 (executes in the L2) | sc1 load | buffer_inv sc0 + sc0 load (a bare sc0 load hits the vector L1 for ever: measured, the spin times out)
//   inv = 0: buffer_inv sc1 after the barrier; 1: buffer_inv sc0; 2: none (expected: stale reads -> mismatches)
// Every spin is bounded (a flag is raised instead of hanging the GPU).   hipcc --offload-arch=gfx950 -O3 tools/probes/xcdsync.cpp -o tools/probes/xcdsync.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int WG = 256, PER = 4096;  // floats per workgroup and layer (16 KB)

__device__ __forceinline__ int partner(int b) {  // ids base + 8 k, k = 0..3: one XCD under round-robin dispatch
    const int base = (b & 7) | ((b >> 5) << 5), k = (b >> 3) & 3;
    return base + 8 * ((k + 1) & 3);
}
__device__ __forceinline__ void layer(const float* __restrict__ src, float* __restrict__ dst, int from, int work) {
    const float4* s = reinterpret_cast<const float4*>(src + (size_t)from * PER);
    float4* d = reinterpret_cast<float4*>(dst + (size_t)blockIdx.x * PER);
#pragma unroll
    for (int k = 0; k < PER / 4 / WG; ++k) {
        float4 v = s[threadIdx.x + k * WG];
        for (int w = 0; w < work; ++w) {  // optional dependent arithmetic: a layer that takes a few microseconds
            v.x = fmaf(v.x, 1.0000001f, 1e-9f); v.y = fmaf(v.y, 1.0000001f, 1e-9f); v.z = fmaf(v.z, 1.0000001f, 1e-9f); v.w = fmaf(v.w, 1.0000001f, 1e-9f);
        }
        v.x = v.x * 0.5f + 1.f; v.y = v.y * 0.5f + 2.f; v.z = v.z * 0.5f + 3.f; v.w = v.w * 0.5f + (float)(blockIdx.x & 7);
        d[threadIdx.x + k * WG] = v;
    }
}
__global__ __launch_bounds__(WG) void k_one(const float* src, float* dst, int work) { layer(src, dst, partner(blockIdx.x), work); }

__global__ __launch_bounds__(WG) void k_chain(float* a, float* b, int layers, unsigned* ctr, int gsize, unsigned* xcc, int inv, int work, int* err, int poll) {
    extern __shared__ float lds[];
    const unsigned my_xcc = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | ((4 - 1) << 11));
    if (threadIdx.x == 0 && xcc) xcc[blockIdx.x] = my_xcc;
    if (my_xcc != (blockIdx.x & 7u) && threadIdx.x == 0) *err = 2;  // the dispatch order this protocol relies on
    const int grp = gsize == 4 ? ((blockIdx.x & 7) | ((blockIdx.x >> 5) << 3)) : (blockIdx.x & 7);
    unsigned* c = ctr + grp * 32;  // one 128-byte line per group
    const int from = partner(blockIdx.x);
    for (int l = 0; l < layers; ++l) {
        layer((l & 1) ? b : a, (l & 1) ? a : b, from, work);
        const unsigned target = (unsigned)(l + 1) * (unsigned)gsize;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its stores are acknowledged by the L2
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            int spins = 0;
            for (;;) {
                unsigned v;
                if (poll == 0) v = __hip_atomic_fetch_add(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // an atomic executes in the L2
                else if (poll == 1) v = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);         // sc1 load
                else {
                    asm volatile("buffer_inv sc0" ::: "memory");
                    v = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);                      // sc0 load behind an L1 invalidate
                }
                if (v >= target) break;
                if (++spins > (1 << 18)) { *err = 1; break; }
            }
        }
        __syncthreads();
        if (inv == 0) asm volatile("buffer_inv sc1" ::: "memory");
        else if (inv == 1) asm volatile("buffer_inv sc0" ::: "memory");
    }
    lds[threadIdx.x] = 0.f;
}

// Does the invalidate really drop the vector L1 (and the scalar cache)?  Every workgroup rewrites ONE 4 KB block in place per round and, behind the barrier, reads its
// partner's block -- the same 4 KB every round, certainly resident in its L1 / scalar cache -- through plain vector loads and through a uniform (scalar) load.
__device__ __forceinline__ void xcd_bar(unsigned* c, unsigned target, int inv, int* err) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        int spins = 0;
        while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target)
            if (++spins > (1 << 18)) { *err = 1; break; }
    }
    __syncthreads();
    if (inv == 0) asm volatile("buffer_inv sc1" ::: "memory");
    else if (inv == 1) asm volatile("buffer_inv sc0" ::: "memory");
    else if (inv == 3) asm volatile("buffer_inv sc0\n\ts_dcache_inv" ::: "memory");
}
__global__ __launch_bounds__(WG) void k_stale(float* buf, int rounds, unsigned* ctr, int inv, int* err, unsigned* bad_v, unsigned* bad_s) {
    extern __shared__ float lds[];
    unsigned* c = ctr + (blockIdx.x & 7) * 32;
    float* mine = buf + (size_t)blockIdx.x * 1024;
    const float* theirs = buf + (size_t)partner(blockIdx.x) * 1024;
    unsigned nv = 0, ns = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int k = 0; k < 4; ++k) mine[threadIdx.x + k * WG] = (float)(r + 1);
        xcd_bar(c, (unsigned)(2 * r + 1) * 32u, inv, err);
        for (int k = 0; k < 4; ++k) nv += theirs[threadIdx.x + k * WG] != (float)(r + 1);
        const float u = theirs[__builtin_amdgcn_readfirstlane(r & 1023)];  // wave-uniform address: a scalar load
        ns += u != (float)(r + 1);
        xcd_bar(c, (unsigned)(2 * r + 2) * 32u, 2, err);  // nobody rewrites its block before every reader is done (no invalidate needed here)
    }
    if (nv) atomicAdd(bad_v, nv);
    if (ns) atomicAdd(bad_s, ns);
    lds[threadIdx.x] = 0.f;
}

int main() {
    const int nwg = 256, layers = 64, smem = 100 * 1024;
    float *a, *b; unsigned *ctr, *xcc; int* err;
    CHK(hipMalloc(&a, (size_t)nwg * PER * 4)); CHK(hipMalloc(&b, (size_t)nwg * PER * 4));
    CHK(hipMalloc(&ctr, 64 * 128)); CHK(hipMalloc(&xcc, nwg * 4)); CHK(hipMalloc(&err, 4));
    CHK(hipFuncSetAttribute((const void*)k_one, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    CHK(hipFuncSetAttribute((const void*)k_chain, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    hipStream_t s; CHK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float ms;
    std::vector<float> ref((size_t)nwg * PER), got((size_t)nwg * PER);
    for (int work : {0, 400}) {
        for (int rep = 0; rep < 2; ++rep) {
            CHK(hipMemsetAsync(a, 0, (size_t)nwg * PER * 4, s)); CHK(hipMemsetAsync(b, 0, (size_t)nwg * PER * 4, s));
            CHK(hipEventRecord(e0, s));
            for (int l = 0; l < layers; ++l) hipLaunchKernelGGL(k_one, dim3(nwg), dim3(WG), smem, s, (l & 1) ? b : a, (l & 1) ? a : b, work);
            CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
            printf("work %3d (a) separate launches                         : %6.2f us per layer\n", work, ms * 1e3 / layers);
            CHK(hipMemcpy(ref.data(), a, ref.size() * 4, hipMemcpyDeviceToHost));
            for (int gsize : {4, 32}) {
              for (int poll = 1; poll < 2; ++poll)
                for (int inv = 0; inv < 3; ++inv) {
                    CHK(hipMemsetAsync(a, 0, (size_t)nwg * PER * 4, s)); CHK(hipMemsetAsync(b, 0, (size_t)nwg * PER * 4, s));
                    CHK(hipMemsetAsync(ctr, 0, 64 * 128, s)); CHK(hipMemsetAsync(err, 0, 4, s));
                    int L = layers, w = work;
                    void* args[] = {&a, &b, &L, &ctr, &gsize, &xcc, &inv, &w, &err, &poll};
                    CHK(hipEventRecord(e0, s));
                    CHK(hipLaunchCooperativeKernel((const void*)k_chain, dim3(nwg), dim3(WG), args, smem, s));
                    CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
                    CHK(hipMemcpy(got.data(), a, got.size() * 4, hipMemcpyDeviceToHost));
                    int herr = 0; CHK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
                    size_t bad = 0;
                    for (size_t i = 0; i < got.size(); ++i) bad += got[i] != ref[i];
                    printf("work %3d (%s) %2d-workgroup XCD-local barrier, poll %-13s, %-15s: %6.2f us per layer   mismatches vs (a): %zu   flag %d\n", work, gsize == 4 ? "f" : "g", gsize,
                           poll == 0 ? "atomic add 0" : (poll == 1 ? "sc1 load" : "inv + sc0 load"), inv == 0 ? "buffer_inv sc1" : (inv == 1 ? "buffer_inv sc0" : "no invalidate"), ms * 1e3 / layers, bad, herr);
                }
            }
        }
    }
    {
        float* buf; unsigned *bv, *bs;
        CHK(hipMalloc(&buf, (size_t)nwg * 1024 * 4)); CHK(hipMalloc(&bv, 4)); CHK(hipMalloc(&bs, 4));
        CHK(hipFuncSetAttribute((const void*)k_stale, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        for (int inv : {2, 1, 3, 0}) {
            CHK(hipMemsetAsync(buf, 0, (size_t)nwg * 1024 * 4, s)); CHK(hipMemsetAsync(ctr, 0, 64 * 128, s)); CHK(hipMemsetAsync(err, 0, 4, s));
            CHK(hipMemsetAsync(bv, 0, 4, s)); CHK(hipMemsetAsync(bs, 0, 4, s));
            int rounds = 200;
            void* args[] = {&buf, &rounds, &ctr, &inv, &err, &bv, &bs};
            CHK(hipEventRecord(e0, s));
            CHK(hipLaunchCooperativeKernel((const void*)k_stale, dim3(nwg), dim3(WG), args, smem, s));
            CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
            unsigned hv = 0, hs = 0; int herr = 0;
            CHK(hipMemcpy(&hv, bv, 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(&hs, bs, 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            printf("staleness, %-28s: stale vector reads %u of %d, stale scalar reads %u of %d   %.2f us per round (two barriers)   flag %d\n",
                   inv == 2 ? "no invalidate" : (inv == 1 ? "buffer_inv sc0" : (inv == 3 ? "buffer_inv sc0 + s_dcache_inv" : "buffer_inv sc1")), hv, nwg * WG * 4 * rounds, hs,
                   nwg * WG * rounds, ms * 1e3 / rounds, herr);
        }
    }
    std::vector<unsigned> hx(nwg);
    CHK(hipMemcpy(hx.data(), xcc, nwg * 4, hipMemcpyDeviceToHost));
    int okx = 0;
    for (int i = 0; i < nwg; ++i) okx += hx[i] == (unsigned)(i & 7);
    printf("workgroups whose XCC_ID == blockIdx %% 8: %d of %d\n", okx, nwg);
    return 0;
}
