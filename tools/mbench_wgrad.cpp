// Microbenchmark of the weight-gradient kernels (dif-pan_amd/csrc/kernels_bwd.h): the bf16x3 kernel (conv3x3_wgrad_x3_kernel) against the exact-fp32 one it replaces,
// results compared (the partial blocks summed over the splits), and the bf16x3 kernel's ablations (development tool, not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I dif-pan_amd/csrc -I include tools/mbench_wgrad.cpp -o tools/mbench_wgrad.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ddif_plan.h"
#include "kernels_bwd.h"
using namespace ddif;
namespace ddif { thread_local std::string g_err; int fail(int c, const char*, ...) { return c; } }
#define CK_(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static int stride8(int n) { return n + (((n / 8) % 2 == 0) ? 8 : 0); }
static hipEvent_t e0, e1;
template <class F>
static float time_it(F&& launch) {
    for (int i = 0; i < 3; ++i) launch();
    CK_(hipDeviceSynchronize());
    const int iters = 30;
    CK_(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) launch();
    CK_(hipEventRecord(e1, 0)); CK_(hipEventSynchronize(e1));
    float ms; CK_(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / iters;
}

template <int CENTRE>
static void run(int B, int H, int W, int Cin, int Cout, int rb_x3, int nsplit_x3, int rb_old, int nsplit_old) {
    const size_t nx = (size_t)B * H * W * Cin, ny = (size_t)B * H * W * Cout;
    std::vector<float> hx(nx), hy(ny);
    for (auto& v : hx) v = (rand() % 2001 - 1000) * 1e-3f;
    for (auto& v : hy) v = (rand() % 2001 - 1000) * 1e-4f;
    float *x, *dy, *p0, *p1;
    const int n_co = (Cout + 31) / 32, n_ci = (Cin + 31) / 32, nblk = n_co * n_ci;
    const size_t pfl0 = (size_t)nsplit_old * nblk * 9 * 1024, pfl1 = (size_t)nsplit_x3 * nblk * 9 * 1024;
    CK_(hipMalloc(&x, nx * 4)); CK_(hipMalloc(&dy, ny * 4)); CK_(hipMalloc(&p0, pfl0 * 4)); CK_(hipMalloc(&p1, pfl1 * 4));
    CK_(hipMemset(p0, 0, pfl0 * 4)); CK_(hipMemset(p1, 0, pfl1 * 4));
    CK_(hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice)); CK_(hipMemcpy(dy, hy.data(), ny * 4, hipMemcpyHostToDevice));
    WgradArgs a{}; a.x = x; a.dy = dy; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.n_ci = n_ci; a.centre_only = CENTRE; a.wshift = -1;
    for (int k = 0; k < 16; ++k) if ((1 << k) == W) a.wshift = k;
    // the fp32 kernel
    WgradArgs a0 = a; a0.rb = rb_old; a0.bands_y = (H + rb_old - 1) / rb_old; a0.partial = p0;
    const int hl = CENTRE ? 0 : 1;
    const size_t smem0 = ((size_t)rb_old * W * 32 + (size_t)(rb_old + 2) * (W + 2) * 32 + 4096) * 4;
    auto k0 = conv3x3_wgrad_kernel<0, 0>;  // (the batch-loading form: any band size; centre_only selects the 1x1 contraction)
    CK_(hipFuncSetAttribute((const void*)k0, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    const float t0 = time_it([&] { hipLaunchKernelGGL(k0, dim3(nblk, nsplit_old), dim3(256), smem0, 0, a0); });
    // the bf16x3 kernel
    WgradArgs a1 = a; a1.rb = rb_x3; a1.bands_y = (H + rb_x3 - 1) / rb_x3; a1.partial = p1;
    WgradX3Geom gm{rb_x3, CENTRE ? W : W + 16, stride8(rb_x3 * W), stride8((rb_x3 + 2 * hl) * (CENTRE ? W : W + 16))};
    size_t smem1 = (size_t)2 * 96 * (gm.ys + gm.xs); if (smem1 < 16384) smem1 = 16384;
    const int items = 2 * rb_x3 * (W / 8) * 8;
    printf("B=%d %dx%d %d->%d %s: fp32 kernel rb=%d nsplit=%d %.1f us | x3 rb=%d nsplit=%d items=%d smem=%zu:", B, H, W, Cin, Cout, CENTRE ? "1x1" : "3x3", rb_old, nsplit_old, t0, rb_x3, nsplit_x3,
           items, smem1);
    if (items > 256) { printf(" (more than 256 items: skipped)\n"); return; }
#define RUNX(ABL) { auto k1 = conv3x3_wgrad_x3_kernel<CENTRE, ABL>; CK_(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); \
        printf("  abl%-2d %.1f", ABL, time_it([&] { hipLaunchKernelGGL(k1, dim3(nblk, nsplit_x3), dim3(256), smem1, 0, a1, gm); })); }
    RUNX(0) RUNX(1) RUNX(4) RUNX(18) RUNX(31)
    printf(" us\n");
    // compare: sum of the partial blocks over the splits
    { auto k1 = conv3x3_wgrad_x3_kernel<CENTRE, 0>; hipLaunchKernelGGL(k1, dim3(nblk, nsplit_x3), dim3(256), smem1, 0, a1, gm); hipLaunchKernelGGL(k0, dim3(nblk, nsplit_old), dim3(256), smem0, 0, a0); }
    CK_(hipDeviceSynchronize());
    std::vector<float> h0(pfl0), h1(pfl1);
    CK_(hipMemcpy(h0.data(), p0, pfl0 * 4, hipMemcpyDeviceToHost)); CK_(hipMemcpy(h1.data(), p1, pfl1 * 4, hipMemcpyDeviceToHost));
    const size_t per = (size_t)nblk * 9 * 1024;
    double dmax = 0, vmax = 0;
    for (size_t i = 0; i < per; ++i) {
        if (CENTRE && (i / 1024) % 9 != 4) continue;
        double s0 = 0, s1 = 0;
        for (int k = 0; k < nsplit_old; ++k) s0 += h0[k * per + i];
        for (int k = 0; k < nsplit_x3; ++k) s1 += h1[k * per + i];
        dmax = std::max(dmax, fabs(s0 - s1)); vmax = std::max(vmax, fabs(s0));
    }
    printf("      max |dW(x3) - dW(fp32)| = %.3g  (max |dW| %.3g)\n", dmax, vmax);
    hipFree(x); hipFree(dy); hipFree(p0); hipFree(p1);
}

int main() {
    CK_(hipEventCreate(&e0)); CK_(hipEventCreate(&e1));
    run<0>(32, 64, 64, 32, 32, 1, 512, 1, 256);
    run<0>(32, 64, 64, 32, 32, 2, 256, 1, 256);
    run<0>(32, 64, 64, 32, 32, 2, 512, 1, 256);
    run<0>(32, 64, 64, 64, 32, 1, 256, 1, 256);
    run<0>(32, 64, 64, 64, 32, 2, 128, 1, 256);
    run<0>(32, 64, 64, 64, 32, 2, 256, 1, 256);
    run<0>(32, 32, 32, 64, 64, 3, 128, 4, 128);
    run<0>(32, 32, 32, 64, 64, 2, 128, 4, 128);
    run<0>(32, 32, 32, 64, 64, 4, 64, 4, 128);
    run<0>(32, 32, 32, 64, 64, 4, 128, 4, 128);
    run<0>(32, 16, 16, 64, 64, 6, 96, 8, 64);
    run<0>(32, 16, 16, 64, 64, 4, 128, 8, 64);
    run<0>(32, 16, 16, 64, 64, 8, 64, 8, 64);
    run<0>(32, 16, 16, 64, 64, 8, 128, 8, 64);
    run<0>(32, 8, 8, 128, 128, 8, 32, 8, 32);
    run<0>(32, 8, 8, 128, 128, 4, 32, 8, 32);
    run<1>(32, 64, 64, 64, 32, 2, 256, 1, 256);
    run<1>(32, 64, 64, 64, 32, 1, 256, 1, 256);
    run<1>(32, 64, 64, 64, 32, 2, 384, 1, 256);
    run<1>(32, 64, 64, 128, 64, 2, 64, 1, 64);
    run<1>(32, 64, 64, 128, 64, 1, 64, 1, 64);
    run<1>(32, 64, 64, 128, 64, 2, 96, 1, 64);
    run<1>(32, 32, 32, 64, 64, 4, 128, 4, 128);
    run<1>(32, 32, 32, 64, 64, 2, 128, 4, 128);
    run<1>(32, 16, 16, 128, 64, 8, 64, 8, 64);
    run<1>(32, 16, 16, 128, 64, 4, 64, 8, 64);
    run<1>(32, 8, 8, 128, 128, 8, 32, 8, 32);
    return 0;
}
