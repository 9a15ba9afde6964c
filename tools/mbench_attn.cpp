// Microbenchmark of the fused bottleneck attention block (csrc/kernels_attn.h): launch time at 4 / 8 wavefronts per sample and s_memtime stamps at
// the stage boundaries; development tool, not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I dif-pan_amd/csrc -I include tools/mbench_attn.cpp -o tools/mbench_attn.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ddif_net.h"
#include "kernels_attn.h"
using namespace ddif;
namespace ddif { thread_local std::string g_err; int fail(int c, const char*, ...) { return c; } }
#define CK_(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NW, int SPLIT = 1>
void run(int B) {
    const size_t n = (size_t)B * 64 * 128;
    float *x, *out, *wq, *wo, *vec; double *st, *sto; long long* dbg;
    const size_t nwq = (size_t)12 * 8 * 3 * 256, nwo = (size_t)4 * 8 * 3 * 256;
    CK_(hipMalloc(&x, n * 4)); CK_(hipMalloc(&out, n * 4)); CK_(hipMalloc(&wq, nwq * 4)); CK_(hipMalloc(&wo, nwo * 4)); CK_(hipMalloc(&vec, 4096 * 4));
    CK_(hipMalloc(&st, (size_t)B * 64 * 16)); CK_(hipMalloc(&sto, (size_t)B * 64)); CK_(hipMalloc(&dbg, 256 * 32 * 8)); CK_(hipMemset(dbg, 0, 256 * 32 * 8));
    std::vector<float> h(n); for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
    CK_(hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice)); CK_(hipMemcpy(vec, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    CK_(hipMemset(wq, 0x11, nwq * 4)); CK_(hipMemset(wo, 0x11, nwo * 4));
    std::vector<double> hs((size_t)B * 64 * 2); for (size_t i = 0; i < hs.size(); i += 2) { hs[i] = 10.0; hs[i + 1] = 5000.0; }
    CK_(hipMemcpy(st, hs.data(), hs.size() * 8, hipMemcpyHostToDevice));
    AttnBlockArgs a{}; a.x = x; a.st = st; a.np = 4; a.gamma = vec; a.beta = vec + 512; a.wqkv = wq; a.wout = wo; a.bout = vec + 1024; a.scale = 0.088f; a.out = out; a.st_out = sto; a.B = B;
    a.xcd = 1; a.dbg = dbg;
    auto fn = attn_block_kernel<NW, 0, SPLIT>;
    auto fs = attn_block_kernel<NW, 1, SPLIT>;
    CK_(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AttnBlockGeom::smem));
    CK_(hipFuncSetAttribute((const void*)fs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AttnBlockGeom::smem));
    hipEvent_t e0, e1; CK_(hipEventCreate(&e0)); CK_(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(fn, dim3(B * SPLIT), dim3(64 * NW), AttnBlockGeom::smem, 0, a);
    CK_(hipDeviceSynchronize());
    const int iters = 50;
    CK_(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(fn, dim3(B * SPLIT), dim3(64 * NW), AttnBlockGeom::smem, 0, a);
    CK_(hipEventRecord(e1, 0)); CK_(hipEventSynchronize(e1));
    float ms; CK_(hipEventElapsedTime(&ms, e0, e1));
    printf("attn_block NW=%d SPLIT=%d B=%3d  %7.2f us per launch (back to back, warm L2)\n", NW, SPLIT, B, ms * 1e3 / iters);
    hipLaunchKernelGGL(fs, dim3(B * SPLIT), dim3(64 * NW), AttnBlockGeom::smem, 0, a);
    CK_(hipDeviceSynchronize());
    std::vector<long long> d(256 * 32);
    CK_(hipMemcpy(d.data(), dbg, d.size() * 8, hipMemcpyDeviceToHost));
    printf("  stamps (s_memtime ticks; deltas): entry | burst issued | stats reduced | xn staged | barrier | qkv done | barrier | attention done | barrier | out issued | stores issued | barrier\n");
    for (int wg : {0, B * SPLIT / 2, B * SPLIT - 1}) {
        printf("  wg %3d:", wg);
        for (int i = 1; i < 12; ++i) printf(" %lld", d[wg * 32 + i] - d[wg * 32 + i - 1]);
        printf("   total %lld\n", d[wg * 32 + 11] - d[wg * 32]);
    }
    hipFree(x); hipFree(out); hipFree(wq); hipFree(wo); hipFree(vec); hipFree(st); hipFree(sto); hipFree(dbg);
}

int main() {
    for (int B : {64, 8}) { run<4>(B); run<4, 2>(B); run<4, 4>(B); run<8>(B); }
    return 0;
}
