// Probe: how fast can a CU turn an fp32 NHWC tensor into the f16x2 operand planes of the conv kernels -- global load (prefetched DEPTH stages ahead),
// GroupNorm affine + SiLU + two-plane split, LDS writes, one barrier per stage -- WITHOUT any matrix work, as a function of the waves per workgroup?
// (The 3x3 conv of the 64 x 64 level stages 1.27 x 33.5 MB this way per launch; its launch takes 31 us.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I dif-pan_amd/csrc tools/probes/staging_rate.cpp -o tools/probes/staging_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ddif_dev.h"
using namespace ddif;
#define CK_(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// one "stage" = NTHR x NIT float4 (a 16-channel chunk of an 18 x 18 halo tile is 1296 float4)
template <int NTHR, int NIT, int DEPTH, bool SILU>
__global__ __launch_bounds__(NTHR) void stage_kernel(const float* in, size_t n4, int stages, float ga, float gb, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);  // [2][NTHR * NIT][5 floats: 2 planes x 8 B + pad]  (20 B per float4 item -> use 6 floats stride)
    const int tid = threadIdx.x;
    const size_t per = (size_t)NTHR * NIT;
    size_t base = (size_t)blockIdx.x * stages * per;
    float4 r[DEPTH][NIT];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int it = 0; it < NIT; ++it) r[d][it] = *reinterpret_cast<const float4*>(in + 4 * ((base + d * per + tid + it * NTHR) % n4));
    float acc = 0.f;
    for (int s = 0; s < stages; s += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            float* dst = As + ((s + d) & 1) * (per * 6);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float x = fmaf((&r[d][it].x)[i], ga, gb);
                    v[i] = SILU ? dd_silu_scaled(x, 1.0f / 16.0f) : x * 16.0f;
                }
                unsigned h01, l01, h23, l23;
                dd_split2_pair(v[0], v[1], &h01, &l01);
                dd_split2_pair(v[2], v[3], &h23, &l23);
                float* p = dst + (size_t)(tid + it * NTHR) * 6;
                *reinterpret_cast<uint2*>(p) = make_uint2(h01, h23);
                *reinterpret_cast<uint2*>(p + 2) = make_uint2(l01, l23);
                // prefetch the stage DEPTH ahead into the slot just consumed
                r[d][it] = *reinterpret_cast<const float4*>(in + 4 * ((base + (size_t)(s + d + DEPTH) * per + tid + it * NTHR) % n4));
            }
            __syncthreads();
            acc += dst[(tid * 7) % (per * 6)];  // something reads the tile
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

template <int NTHR, int NIT, int DEPTH, bool SILU>
void run(const char* name, const float* in, size_t n4, float* sink, int wg_per_cu) {
    const size_t per = (size_t)NTHR * NIT;
    const int grid = 256 * wg_per_cu;
    int stages = (int)(n4 / per / grid);
    stages -= stages % DEPTH;
    const size_t smem = 2 * per * 6 * 4;
    auto fn = stage_kernel<NTHR, NIT, DEPTH, SILU>;
    CK_(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipEvent_t e0, e1; CK_(hipEventCreate(&e0)); CK_(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(fn, dim3(grid), dim3(NTHR), smem, 0, in, n4, stages, 1.01f, 0.02f, sink);
    CK_(hipDeviceSynchronize());
    CK_(hipEventRecord(e0, 0));
    const int iters = 20;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(fn, dim3(grid), dim3(NTHR), smem, 0, in, n4, stages, 1.01f, 0.02f, sink);
    CK_(hipEventRecord(e1, 0)); CK_(hipEventSynchronize(e1));
    float ms; CK_(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, mb = (double)grid * stages * per * 16 / 1e6;
    printf("%-44s thr=%4d items=%d depth=%d wg/cu=%d smem=%6zu stages=%3d  %7.1f us for %6.1f MB = %5.0f GB/s\n", name, NTHR, NIT, DEPTH, wg_per_cu, smem, stages, us, mb, mb / us * 1e3);
}

int main() {
    const size_t n4 = (size_t)64 * 64 * 64 * 32 / 4 * 2;  // 2 x the 64 x 64 x 32-channel tensor of 64 tiles (67 MB), in float4s
    float *in, *sink;
    CK_(hipMalloc(&in, n4 * 16)); CK_(hipMalloc(&sink, 64));
    std::vector<float> h(n4 * 4); for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
    CK_(hipMemcpy(in, h.data(), n4 * 16, hipMemcpyHostToDevice));
    run<512, 3, 1, true>("GN+SiLU+split, 8 waves, 1 ahead", in, n4, sink, 1);
    run<512, 3, 2, true>("GN+SiLU+split, 8 waves, 2 ahead", in, n4, sink, 1);
    run<512, 3, 2, false>("scale+split (no SiLU), 8 waves, 2 ahead", in, n4, sink, 1);
    run<1024, 2, 2, true>("GN+SiLU+split, 16 waves, 2 ahead", in, n4, sink, 1);
    run<1024, 2, 4, true>("GN+SiLU+split, 16 waves, 4 ahead", in, n4, sink, 1);
    run<256, 3, 2, true>("GN+SiLU+split, 4 waves x 2 wg/cu, 2 ahead", in, n4, sink, 2);
    run<256, 3, 2, true>("GN+SiLU+split, 4 waves x 4 wg/cu, 2 ahead", in, n4, sink, 4);
    run<256, 6, 2, true>("GN+SiLU+split, 4 waves x 2 wg/cu, 6 items", in, n4, sink, 2);
    return 0;
}
