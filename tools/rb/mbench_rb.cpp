// Microbenchmark + host check of the resident ResnetBlock kernel of the 8 x 8 level (kernels_rb.h) against the two conv_lr_kernel launches it
// replaces (development tool, not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I dif-pan_amd/csrc -I tools/rb -I include tools/rb/mbench_rb.cpp -o tools/mbench_rb.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "ddif_net.h"
#include "kernels_lr.h"
#include "kernels_rb.h"
using namespace ddif;
namespace ddif { thread_local std::string g_err; int fail(int c, const char*, ...) { return c; } }
#define CK_(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static float frand(float s) { return (rand() % 20001 - 10000) * 1e-4f * s; }
// the f16x2 pack of ddif_net.cpp pack_conv_f16 for a 3x3 conv with 16-channel chunks: [n-block][chunk][tap][plane][half h][cout j][8 halves]
static void pack_f16(const std::vector<float>& w, int cout, int cin, std::vector<_Float16>& o) {
    const int nch = cin / 16, nb = cout / 32;
    o.assign((size_t)nb * nch * 9 * 2 * 512, (_Float16)0.f);
    for (int nbi = 0; nbi < nb; ++nbi)
        for (int ch = 0; ch < nch; ++ch)
            for (int tap = 0; tap < 9; ++tap)
                for (int h = 0; h < 2; ++h)
                    for (int j = 0; j < 32; ++j)
                        for (int t = 0; t < 8; ++t) {
                            const int ci = ch * 16 + 8 * h + t, co = nbi * 32 + j;
                            const float val = w[((size_t)co * cin + ci) * 9 + tap] * 1024.0f;
                            const _Float16 hi = (_Float16)val, lo = (_Float16)(val - (float)hi);
                            const size_t fl = ((((size_t)nbi * nch + ch) * 9 + tap) * 2) * 256 + (size_t)(h * 32 + j) * 4;
                            o[fl * 2 + t] = hi;
                            o[(fl + 256) * 2 + t] = lo;
                        }
}
static double silu(double x) { return x / (1.0 + exp(-x)); }

int main(int argc, char** argv) {
    const int C = 128, NP = 64;
    for (int B : {64, 8, 256}) {
        srand(11);
        const size_t n = (size_t)B * NP * C;
        std::vector<float> hx(n), hw1((size_t)C * C * 9), hw2((size_t)C * C * 9), hg1(C), hb1(C), hg2(C), hb2(C), hbias1(C), hbias2(C), htb(C);
        for (auto& v : hx) v = frand(1.5f);
        for (auto& v : hw1) v = frand(0.05f);
        for (auto& v : hw2) v = frand(0.05f);
        for (int c = 0; c < C; ++c) { hg1[c] = 1.f + frand(0.3f); hb1[c] = frand(0.2f); hg2[c] = 1.f + frand(0.3f); hb2[c] = frand(0.2f); hbias1[c] = frand(0.1f); hbias2[c] = frand(0.1f); htb[c] = frand(0.3f); }
        std::vector<_Float16> p1, p2; pack_f16(hw1, C, C, p1); pack_f16(hw2, C, C, p2);
        // producer statistics of x: ONE partial per sample (np = 1), exact sums
        std::vector<double> hst((size_t)B * 2);
        for (int b = 0; b < B; ++b) { double s = 0, ss = 0; for (int i = 0; i < NP * C; ++i) { const double v = hx[(size_t)b * NP * C + i]; s += v; ss += v * v; } hst[2 * b] = s; hst[2 * b + 1] = ss; }
        float *x, *w1, *w2, *g1, *b1, *g2, *b2, *bias1, *bias2, *tb, *out, *h1, *out2; double *st, *sto, *sth, *sto2;
        CK_(hipMalloc(&x, n * 4)); CK_(hipMalloc(&out, n * 4)); CK_(hipMalloc(&h1, n * 4)); CK_(hipMalloc(&out2, n * 4));
        CK_(hipMalloc(&w1, p1.size() * 2)); CK_(hipMalloc(&w2, p2.size() * 2));
        for (float** p : {&g1, &b1, &g2, &b2, &bias1, &bias2, &tb}) CK_(hipMalloc(p, C * 4));
        CK_(hipMalloc(&st, B * 16)); CK_(hipMalloc(&sto, B * 16)); CK_(hipMalloc(&sth, (size_t)B * 4 * 16)); CK_(hipMalloc(&sto2, (size_t)B * 4 * 16));
        CK_(hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice)); CK_(hipMemcpy(w1, p1.data(), p1.size() * 2, hipMemcpyHostToDevice)); CK_(hipMemcpy(w2, p2.data(), p2.size() * 2, hipMemcpyHostToDevice));
        CK_(hipMemcpy(g1, hg1.data(), C * 4, hipMemcpyHostToDevice)); CK_(hipMemcpy(b1, hb1.data(), C * 4, hipMemcpyHostToDevice));
        CK_(hipMemcpy(g2, hg2.data(), C * 4, hipMemcpyHostToDevice)); CK_(hipMemcpy(b2, hb2.data(), C * 4, hipMemcpyHostToDevice));
        CK_(hipMemcpy(bias1, hbias1.data(), C * 4, hipMemcpyHostToDevice)); CK_(hipMemcpy(bias2, hbias2.data(), C * 4, hipMemcpyHostToDevice));
        CK_(hipMemcpy(tb, htb.data(), C * 4, hipMemcpyHostToDevice)); CK_(hipMemcpy(st, hst.data(), B * 16, hipMemcpyHostToDevice));
        RbArgs a{}; a.x = x; a.st = st; a.np = 1; a.g1 = g1; a.b1 = b1; a.g2 = g2; a.b2 = b2; a.w1 = w1; a.w2 = w2; a.bias1 = bias1; a.bias2 = bias2;
        a.tbias = tb; a.tbias_stride = 0; a.step_ptr = nullptr; a.tb_rowstride = 0; a.out = out; a.st_out = sto; a.B = B;
        // the two launches it replaces
        ConvArgs c1{}; c1.in0 = x; c1.c0 = C; c1.B = B; c1.Hin = c1.Win = c1.Hout = c1.Wout = 8; c1.Cout = C; c1.w = w1; c1.n_chunks = 8; c1.bias = bias1; c1.tbias = tb;
        c1.gamma = g1; c1.beta = b1; c1.out = h1; c1.tiles_x = c1.tiles_y = 1; c1.n_ct = 4; c1.st0 = st; c1.np0 = 1; c1.st_out = sth;
        ConvArgs c2 = c1; c2.in0 = h1; c2.w = w2; c2.bias = bias2; float* zeros; CK_(hipMalloc(&zeros, 4096)); CK_(hipMemset(zeros, 0, 4096)); c2.tbias = zeros;
        c2.gamma = g2; c2.beta = b2; c2.out = out2; c2.res = x; c2.st0 = sth; c2.np0 = 4; c2.st_out = sto2;
        auto k1 = conv_lr_kernel<3, 2, PRO_GN_SILU, 0, 0, true>;
        auto k2 = conv_lr_kernel<3, 2, PRO_GN_SILU, EPI_RES, 0, true>;
        using GL = LrGeom<3, 2, PRO_GN_SILU, false, true>;
        CK_(hipFuncSetAttribute((const void*)resblock8_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RbGeom::smem));
        CK_(hipFuncSetAttribute((const void*)resblock8_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RbGeom::smem));
        CK_(hipFuncSetAttribute((const void*)resblock8_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RbGeom::smem));
        CK_(hipFuncSetAttribute((const void*)resblock8_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RbGeom::smem));
        const int grid = B < 256 ? B : 256, gl = (B * 4 < 512) ? B * 4 : 512;
        hipEvent_t e0, e1; CK_(hipEventCreate(&e0)); CK_(hipEventCreate(&e1));
        auto time_it = [&](auto&& launch, const char* name) {
            for (int i = 0; i < 3; ++i) launch();
            CK_(hipDeviceSynchronize());
            const int iters = 50;
            CK_(hipEventRecord(e0, 0));
            for (int i = 0; i < iters; ++i) launch();
            CK_(hipEventRecord(e1, 0)); CK_(hipEventSynchronize(e1));
            float ms; CK_(hipEventElapsedTime(&ms, e0, e1));
            printf("B=%3d %-58s %7.2f us\n", B, name, ms * 1e3 / iters);
        };
        time_it([&] { hipLaunchKernelGGL(k1, dim3(gl), dim3(256), GL::smem, 0, c1); hipLaunchKernelGGL(k2, dim3(gl), dim3(256), GL::smem, 0, c2); }, "two conv_lr launches (res.conv1 + res.conv2)");
        time_it([&] { hipLaunchKernelGGL(resblock8_kernel<0>, dim3(grid), dim3(512), RbGeom::smem, 0, a); }, "resblock8 (one launch, one workgroup per sample)");
        time_it([&] { hipLaunchKernelGGL(resblock8_kernel<1>, dim3(grid), dim3(512), RbGeom::smem, 0, a); }, "resblock8 -wload");
        time_it([&] { hipLaunchKernelGGL(resblock8_kernel<2>, dim3(grid), dim3(512), RbGeom::smem, 0, a); }, "resblock8 -mfma");
        time_it([&] { hipLaunchKernelGGL(resblock8_kernel<3>, dim3(grid), dim3(512), RbGeom::smem, 0, a); }, "resblock8 -wload -mfma");
        // results: fused vs the pair, and both against fp64 on the host (samples 0 and B - 1)
        hipLaunchKernelGGL(resblock8_kernel<0>, dim3(grid), dim3(512), RbGeom::smem, 0, a);
        hipLaunchKernelGGL(k1, dim3(gl), dim3(256), GL::smem, 0, c1); hipLaunchKernelGGL(k2, dim3(gl), dim3(256), GL::smem, 0, c2);
        CK_(hipDeviceSynchronize());
        std::vector<float> ho(n), ho2(n); std::vector<double> hso(B * 2), hso2((size_t)B * 8);
        CK_(hipMemcpy(ho.data(), out, n * 4, hipMemcpyDeviceToHost)); CK_(hipMemcpy(ho2.data(), out2, n * 4, hipMemcpyDeviceToHost));
        CK_(hipMemcpy(hso.data(), sto, B * 16, hipMemcpyDeviceToHost)); CK_(hipMemcpy(hso2.data(), sto2, (size_t)B * 64, hipMemcpyDeviceToHost));
        double dmax = 0; for (size_t i = 0; i < n; ++i) dmax = std::max(dmax, (double)fabsf(ho[i] - ho2[i]));
        double smax = 0; for (int b = 0; b < B; ++b) for (int k = 0; k < 2; ++k) { const double t = hso2[b * 8 + k] + hso2[b * 8 + 2 + k] + hso2[b * 8 + 4 + k] + hso2[b * 8 + 6 + k]; smax = std::max(smax, fabs(t - hso[b * 2 + k]) / (1.0 + fabs(t))); }
        printf("B=%3d fused vs the two launches: max |d out| %.3g, max rel |d statistics| %.3g\n", B, dmax, smax);
        double emax = 0, emax2 = 0, vmax = 0;
        for (int b : {0, B - 1}) {
            auto conv = [&](const std::vector<double>& act, const std::vector<float>& w, std::vector<double>& o) {
                o.assign((size_t)NP * C, 0.0);
                for (int y = 0; y < 8; ++y) for (int xx = 0; xx < 8; ++xx) for (int co = 0; co < C; ++co) {
                    double s = 0;
                    for (int tap = 0; tap < 9; ++tap) { const int iy = y + tap / 3 - 1, ix = xx + tap % 3 - 1; if (iy < 0 || iy > 7 || ix < 0 || ix > 7) continue;
                        for (int ci = 0; ci < C; ++ci) s += act[(size_t)(iy * 8 + ix) * C + ci] * (double)w[((size_t)co * C + ci) * 9 + tap]; }
                    o[(size_t)(y * 8 + xx) * C + co] = s;
                }
            };
            auto gn_silu = [&](const std::vector<double>& in, const std::vector<float>& g, const std::vector<float>& bt, std::vector<double>& o) {
                double s = 0, ss = 0; for (double v : in) { s += v; ss += v * v; }
                const double mu = s / in.size(), var = ss / in.size() - mu * mu, rs = 1.0 / sqrt(var + 1e-5);
                o.resize(in.size()); for (size_t i = 0; i < in.size(); ++i) o[i] = silu((in[i] - mu) * rs * g[i % C] + bt[i % C]);
            };
            std::vector<double> xs((size_t)NP * C), a1, hh, a2, oo;
            for (size_t i = 0; i < xs.size(); ++i) xs[i] = hx[(size_t)b * NP * C + i];
            gn_silu(xs, hg1, hb1, a1); conv(a1, hw1, hh);
            for (size_t i = 0; i < hh.size(); ++i) hh[i] += (double)hbias1[i % C] + (double)htb[i % C];
            gn_silu(hh, hg2, hb2, a2); conv(a2, hw2, oo);
            for (size_t i = 0; i < oo.size(); ++i) { const double ref = oo[i] + hbias2[i % C] + xs[i]; vmax = std::max(vmax, fabs(ref));
                emax = std::max(emax, fabs(ref - ho[(size_t)b * NP * C + i])); emax2 = std::max(emax2, fabs(ref - ho2[(size_t)b * NP * C + i])); }
        }
        printf("B=%3d against fp64 (samples 0 and B-1, exact fp32 weights): fused max |err| %.3g, two launches %.3g, max |ref| %.3g\n", B, emax, emax2, vmax);
        for (float* p : {x, w1, w2, g1, b1, g2, b2, bias1, bias2, tb, out, h1, out2, zeros}) hipFree(p);
        hipFree(st); hipFree(sto); hipFree(sth); hipFree(sto2);
    }
    return 0;
}
