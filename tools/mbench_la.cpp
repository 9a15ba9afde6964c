// Microbenchmark / ablation harness for the fused linear-attention kernel (csrc/kernels_lafuse.h); development tool, not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I dif-pan_amd/csrc tools/mbench_la.cpp -o tools/mbench_la.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ddif_net.h"
#include "kernels_lafuse.h"
using namespace ddif;
namespace ddif { thread_local std::string g_err; int fail(int c, const char*, ...) { return c; } }
#define CK_(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int TH, int TW, int NBQ, int NBA, int ABL>
void run(const char* name, int B, int W, int c0, int grid_cap = 256) {
    using G = LaFuseGeom<TH, TW, NBQ, NBA>;
    const int H = TH, fea = 32 * NBQ, c1 = fea - c0, dout = 32 * NBA;
    const size_t n0 = (size_t)B * H * W * c0, n1 = (size_t)B * H * W * c1, nout = (size_t)B * H * W * dout;
    const int nbq_pad = (NBQ + 3) & ~3, nba_pad = (NBA + 3) & ~3;
    const size_t nwq = (size_t)nbq_pad * NBQ * 2 * 2 * 256, per = (size_t)nba_pad * (2 * NBQ) * 2 * 3 * 256;
    float *in0, *in1, *out, *wq, *wmix, *vec;
    double* st;
    CK_(hipMalloc(&in0, n0 * 4)); CK_(hipMalloc(&in1, n1 * 4)); CK_(hipMalloc(&out, nout * 4)); CK_(hipMalloc(&wq, nwq * 4)); CK_(hipMalloc(&wmix, per * B * 4));
    CK_(hipMalloc(&vec, 16384 * 4)); CK_(hipMalloc(&st, (size_t)B * 64 * 16));
    std::vector<float> h(std::max(std::max(n0, n1), std::max(per * B, (size_t)16384)));
    for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
    CK_(hipMemcpy(in0, h.data(), n0 * 4, hipMemcpyHostToDevice)); CK_(hipMemcpy(in1, h.data(), n1 * 4, hipMemcpyHostToDevice));
    CK_(hipMemcpy(vec, h.data(), 16384 * 4, hipMemcpyHostToDevice));
    CK_(hipMemset(wq, 0x11, nwq * 4)); CK_(hipMemset(wmix, 0x11, per * B * 4));  // small finite halves / bf16s: timing only
    std::vector<double> hs((size_t)B * 64 * 2);
    for (size_t i = 0; i < hs.size(); i += 2) { hs[i] = 10.0; hs[i + 1] = 5000.0; }
    CK_(hipMemcpy(st, hs.data(), hs.size() * 8, hipMemcpyHostToDevice));
    LaFuseArgs a{};
    a.in0 = in0; a.c0 = c0; a.in1 = in1; a.c1 = c1; a.B = B; a.H = H; a.W = W; a.st0 = st; a.np0 = 64; a.st1 = st; a.np1 = 64;
    a.gamma = vec; a.beta = vec + 512; a.dw_w = vec + 1024; a.wq = wq; a.nchq = NBQ; a.bq = vec + 4096; a.wmix = wmix; a.wmix_bstride = (long long)per;
    a.nch_mix = 2 * NBQ; a.bias = vec + 8192; a.out = out; a.dout = dout;
    const long nwork = (long)B * ((W + TW - 1) / TW);
    dim3 grid((unsigned)(nwork < grid_cap ? nwork : grid_cap));
    auto fn = linattn_fused_kernel<TH, TW, NBQ, NBA, ABL>;
    CK_(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::smem));
    hipEvent_t e0, e1; CK_(hipEventCreate(&e0)); CK_(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(fn, grid, dim3(512), G::smem, 0, a);
    CK_(hipDeviceSynchronize());
    const int iters = 30;
    CK_(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(fn, grid, dim3(512), G::smem, 0, a);
    CK_(hipEventRecord(e1, 0)); CK_(hipEventSynchronize(e1));
    float ms; CK_(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, mb = 4.0 * B * H * W * (fea + dout) / 1e6;
    printf("%-34s abl=%2d grid=%4u smem=%6zu  %7.2f us  %6.0f GB/s (algorithmic)\n", name, ABL, grid.x, (size_t)G::smem, us, mb / us * 1e3);
    hipFree(in0); hipFree(in1); hipFree(out); hipFree(wq); hipFree(wmix); hipFree(vec); hipFree(st);
}

template <int TH, int TW, int NBQ, int NBA>
void stamps(const char* name, int B, int W, int c0) {
    // one run with s_memtime stamps of thread 0 of a few workgroups: per-stage cycles of the first work item
    using G = LaFuseGeom<TH, TW, NBQ, NBA>;
    const int H = TH, fea = 32 * NBQ, c1 = fea - c0, dout = 32 * NBA;
    const size_t n0 = (size_t)B * H * W * c0, n1 = (size_t)B * H * W * c1, nout = (size_t)B * H * W * dout;
    const int nbq_pad = (NBQ + 3) & ~3, nba_pad = (NBA + 3) & ~3;
    const size_t nwq = (size_t)nbq_pad * NBQ * 2 * 2 * 256, per = (size_t)nba_pad * (2 * NBQ) * 2 * 3 * 256;
    float *in0, *in1, *out, *wq, *wmix, *vec; double* st; long long* dbg;
    CK_(hipMalloc(&in0, n0 * 4)); CK_(hipMalloc(&in1, n1 * 4)); CK_(hipMalloc(&out, nout * 4)); CK_(hipMalloc(&wq, nwq * 4)); CK_(hipMalloc(&wmix, per * B * 4));
    CK_(hipMalloc(&vec, 16384 * 4)); CK_(hipMalloc(&st, (size_t)B * 64 * 16)); CK_(hipMalloc(&dbg, 256 * 128 * 8)); CK_(hipMemset(dbg, 0, 256 * 128 * 8));
    CK_(hipMemset(in0, 0, n0 * 4)); CK_(hipMemset(in1, 0, n1 * 4)); CK_(hipMemset(vec, 0, 16384 * 4)); CK_(hipMemset(wq, 0x11, nwq * 4)); CK_(hipMemset(wmix, 0x11, per * B * 4));
    std::vector<double> hs((size_t)B * 64 * 2);
    for (size_t i = 0; i < hs.size(); i += 2) { hs[i] = 10.0; hs[i + 1] = 5000.0; }
    CK_(hipMemcpy(st, hs.data(), hs.size() * 8, hipMemcpyHostToDevice));
    LaFuseArgs a{};
    a.in0 = in0; a.c0 = c0; a.in1 = in1; a.c1 = c1; a.B = B; a.H = H; a.W = W; a.st0 = st; a.np0 = 64; a.st1 = st; a.np1 = 64;
    a.gamma = vec; a.beta = vec + 512; a.dw_w = vec + 1024; a.wq = wq; a.nchq = NBQ; a.bq = vec + 4096; a.wmix = wmix; a.wmix_bstride = (long long)per;
    a.nch_mix = 2 * NBQ; a.bias = vec + 8192; a.out = out; a.dout = dout; a.dbg = dbg;
    const long nwork = (long)B * ((W + TW - 1) / TW);
    dim3 grid((unsigned)(nwork < 256 ? nwork : 256));
    auto fn = linattn_fused_kernel<TH, TW, NBQ, NBA, 64>;
    CK_(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::smem));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(fn, grid, dim3(512), G::smem, 0, a);
    CK_(hipDeviceSynchronize());
    std::vector<long long> h(256 * 128);
    CK_(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
    printf("%s: s_memtime deltas (x10 ns at 100 MHz) of workgroups 0, 100: per chunk [start->GN done, ->barrier, ->dw done, ->barrier, ->mfma issued], then softmax+attn\n", name);
    for (int wg : {0, 100}) {
        printf("  wg %3d:", wg);
        for (int i = 1; i < 127 && h[wg * 128 + i]; ++i) printf(" %lld", h[wg * 128 + i] - h[wg * 128 + i - 1]);
        printf("\n");
    }
}

int main(int argc, char** argv) {
    const int B = 64;
    if (argc > 1 && argv[1][0] == 's') {
        stamps<64, 4, 2, 1>("64x64 32+32->32", B, 64, 32);
        stamps<32, 8, 4, 2>("32x32 64+64->64", B, 32, 64);
        return 0;
    }
#define ABLS(TH, TW, NBQ, NBA, NAME, W, C0) \
    run<TH, TW, NBQ, NBA, 0>(NAME, B, W, C0); run<TH, TW, NBQ, NBA, 1>(NAME " -inload", B, W, C0); run<TH, TW, NBQ, NBA, 2>(NAME " -wload", B, W, C0); \
    run<TH, TW, NBQ, NBA, 4>(NAME " -mfma", B, W, C0); run<TH, TW, NBQ, NBA, 8>(NAME " -dw", B, W, C0); run<TH, TW, NBQ, NBA, 16>(NAME " -softmax/attn", B, W, C0); \
    run<TH, TW, NBQ, NBA, 32>(NAME " -store", B, W, C0); run<TH, TW, NBQ, NBA, 3>(NAME " -in -w loads", B, W, C0); run<TH, TW, NBQ, NBA, 63>(NAME " -all", B, W, C0);
    ABLS(64, 4, 2, 1, "64x64 32+32->32", 64, 32)
    ABLS(32, 8, 4, 2, "32x32 64+64->64", 32, 64)
    run<64, 4, 3, 1, 0>("64x64 64+32->32", B, 64, 64);
    run<32, 8, 3, 2, 0>("32x32 64+32->64", B, 32, 64);
    return 0;
}
