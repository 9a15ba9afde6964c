// Microbenchmark of the fused 8 x 8 linear-attention kernel (csrc/kernels_lafuse8.h): launch time and s_memtime stamps; development tool, not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffinite-math-only -I dif-pan_amd/csrc -I include tools/mbench_la8.cpp -o tools/mbench_la8.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ddif_net.h"
#include "kernels_lafuse8.h"
using namespace ddif;
namespace ddif { thread_local std::string g_err; int fail(int c, const char*, ...) { return c; } }
#define CK_(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NBQ>
void run(int B) {
    using G = LaFuse8Geom<NBQ>;
    const int fea = 32 * NBQ, c0 = 128, c1 = fea - c0, dout = 128;
    const size_t n0 = (size_t)B * 64 * c0, n1 = (size_t)B * 64 * c1, nout = (size_t)B * 64 * dout;
    const int nbq_pad = (NBQ + 3) & ~3;
    const size_t nwq = (size_t)nbq_pad * NBQ * 2 * 2 * 256, per = (size_t)4 * (2 * NBQ) * 2 * 3 * 256;
    float *in0, *in1, *out, *wq, *wmix, *vec; double* st; long long* dbg;
    CK_(hipMalloc(&in0, n0 * 4)); CK_(hipMalloc(&in1, n1 * 4)); CK_(hipMalloc(&out, nout * 4)); CK_(hipMalloc(&wq, nwq * 4)); CK_(hipMalloc(&wmix, per * B * 4));
    CK_(hipMalloc(&vec, 16384 * 4)); CK_(hipMalloc(&st, (size_t)B * 64 * 16)); CK_(hipMalloc(&dbg, 256 * 64 * 8)); CK_(hipMemset(dbg, 0, 256 * 64 * 8));
    std::vector<float> h(std::max(std::max(n0, n1), (size_t)16384));
    for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
    CK_(hipMemcpy(in0, h.data(), n0 * 4, hipMemcpyHostToDevice)); CK_(hipMemcpy(in1, h.data(), n1 * 4, hipMemcpyHostToDevice));
    CK_(hipMemcpy(vec, h.data(), 16384 * 4, hipMemcpyHostToDevice));
    CK_(hipMemset(wq, 0x11, nwq * 4)); CK_(hipMemset(wmix, 0x11, per * B * 4));
    std::vector<double> hs((size_t)B * 64 * 2);
    for (size_t i = 0; i < hs.size(); i += 2) { hs[i] = 10.0; hs[i + 1] = 5000.0; }
    CK_(hipMemcpy(st, hs.data(), hs.size() * 8, hipMemcpyHostToDevice));
    LaFuseArgs a{};
    a.in0 = in0; a.c0 = c0; a.in1 = in1; a.c1 = c1; a.B = B; a.H = 8; a.W = 8; a.st0 = st; a.np0 = 4; a.st1 = st; a.np1 = 4;
    a.gamma = vec; a.beta = vec + 512; a.dw_w = vec + 1024; a.wq = wq; a.nchq = NBQ; a.bq = vec + 4096; a.wmix = wmix; a.wmix_bstride = (long long)per;
    a.nch_mix = 2 * NBQ; a.bias = vec + 8192; a.out = out; a.dout = dout; a.dbg = dbg; a.xcd = 1;
    const int grid = 2 * B < 256 ? 2 * B : 256;
    auto fn = linattn8_fused_kernel<NBQ, 0>;
    auto fs = linattn8_fused_kernel<NBQ, 64>;
    CK_(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::smem));
    CK_(hipFuncSetAttribute((const void*)fs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::smem));
    hipEvent_t e0, e1; CK_(hipEventCreate(&e0)); CK_(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(fn, dim3(grid), dim3(512), G::smem, 0, a);
    CK_(hipDeviceSynchronize());
    const int iters = 50;
    CK_(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(fn, dim3(grid), dim3(512), G::smem, 0, a);
    CK_(hipEventRecord(e1, 0)); CK_(hipEventSynchronize(e1));
    float ms; CK_(hipEventElapsedTime(&ms, e0, e1));
    printf("linattn8_fused fea=%d B=%3d grid=%3d smem=%zu  %7.2f us per launch (back to back, warm L2)\n", fea, B, grid, (size_t)G::smem, ms * 1e3 / iters);
    hipLaunchKernelGGL(fs, dim3(grid), dim3(512), G::smem, 0, a);
    CK_(hipDeviceSynchronize());
    std::vector<long long> d(256 * 64);
    CK_(hipMemcpy(d.data(), dbg, d.size() * 8, hipMemcpyDeviceToHost));
    printf("  stamps (s_memtime ticks, deltas): entry -> tables done | chunk loop done | p in LDS | M_b contraction issued | stores issued\n");
    for (int wg : {0, grid / 2, grid - 1}) {
        printf("  wg %3d:", wg);
        for (int i = 1; i < 6; ++i) printf(" %lld", d[wg * 64 + i] - d[wg * 64 + i - 1]);
        printf("   total %lld\n", d[wg * 64 + 5] - d[wg * 64]);
    }
    {   // per-chunk anatomy: weights issued | GroupNorm stage | barrier | depthwise + split | barrier | contraction issued
        auto fc = linattn8_fused_kernel<NBQ, 64 | 128>;
        CK_(hipFuncSetAttribute((const void*)fc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::smem));
        CK_(hipMemset(dbg, 0, 256 * 64 * 8));
        hipLaunchKernelGGL(fc, dim3(grid), dim3(512), G::smem, 0, a);
        CK_(hipDeviceSynchronize());
        CK_(hipMemcpy(d.data(), dbg, d.size() * 8, hipMemcpyDeviceToHost));
        printf("  chunk anatomy of wg 0 (deltas; per chunk: fragments issued | GN stage | barrier | depthwise | barrier | contraction issued):\n   ");
        for (int i = 2; i < 2 + 6 * (NBQ / 2) + 1 && d[i]; ++i) printf(" %lld", d[i] - d[i - 1]);
        printf("\n");
    }
    hipFree(in0); hipFree(in1); hipFree(out); hipFree(wq); hipFree(wmix); hipFree(vec); hipFree(st); hipFree(dbg);
}

int main() {
    for (int B : {64, 8}) { run<8>(B); run<6>(B); }
    return 0;
}
