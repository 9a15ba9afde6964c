// Microbenchmark / ablation harness for the conv kernel (development tool, not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I dif-pan_amd/csrc tools/mbench.cpp -o /tmp/mbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <algorithm>
#include "ddif_net.h"
#include "kernels_conv.h"
using namespace ddif;
typedef void (*ConvKernelFnT)(ConvArgs);
namespace ddif { thread_local std::string g_err; int fail(int c, const char*, ...) { return c; } }

#define CK_(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static std::vector<float> g_last_out;  // output of the previous run (mode r compares two instantiations bit for bit)
static bool g_keep_out = false;
template <int KS, int S, int U, int TH, int TW, int CKc, int WM, int WN, int MB, int NB, int PRO, int ABL, int EPI = 0, int MATH = 0>
void run(const char* name, int B, int H, int W, int Cin, int Cout, int wg_per_cu) {
    const int Hout = S == 2 ? (H - 1) / 2 + 1 : (U ? 2 * H : H), Wout = S == 2 ? (W - 1) / 2 + 1 : (U ? 2 * W : W);
    const int n_chunks = (Cin + CKc - 1) / CKc, nb = (Cout + 31) / 32, nb_pad = (nb + 3) & ~3;
    const size_t nin = (size_t)B * H * W * Cin, nout = (size_t)B * Hout * Wout * Cout;
    const size_t nw = (size_t)nb_pad * n_chunks * (MATH >= 1 ? KS * KS * 3 * 256 : KS * KS * (CKc / 8) * 256);
    float *in, *w, *out, *gamma, *beta, *bias, *res; double *st, *sto;
    CK_(hipMalloc(&in, nin * 4)); CK_(hipMalloc(&w, nw * 4)); CK_(hipMalloc(&out, nout * 4)); CK_(hipMalloc(&res, nout * 4));
    CK_(hipMalloc(&gamma, Cin * 4)); CK_(hipMalloc(&beta, Cin * 4)); CK_(hipMalloc(&bias, Cout * 4));
    std::vector<float> h(std::max(std::max(nin, nw), nout)); for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
    CK_(hipMemcpy(in, h.data(), nin * 4, hipMemcpyHostToDevice)); CK_(hipMemcpy(w, h.data(), nw * 4, hipMemcpyHostToDevice));
    std::vector<_Float16> hw;
    if (MATH == 3 || MATH == 5) {  // valid halves (the float bit patterns above contain half NaNs): hi / lo planes of small numbers
        hw.resize(nw * 2); for (auto& v : hw) v = (_Float16)((rand() % 2001 - 1000) * 1e-3f);
        CK_(hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice));
    }
    CK_(hipMemcpy(res, h.data(), nout * 4, hipMemcpyHostToDevice));
    CK_(hipMemcpy(gamma, h.data(), Cin * 4, hipMemcpyHostToDevice)); CK_(hipMemcpy(beta, h.data(), Cin * 4, hipMemcpyHostToDevice));
    CK_(hipMemcpy(bias, h.data(), Cout * 4, hipMemcpyHostToDevice));
    ConvArgs a{}; a.in0 = in; a.c0 = Cin; a.B = B; a.Hin = H; a.Win = W; a.Hout = Hout; a.Wout = Wout; a.Cout = Cout; a.w = w; a.n_chunks = n_chunks;
    a.bias = bias; a.gamma = gamma; a.beta = beta; a.out = out; a.tiles_x = (Wout + TW - 1) / TW; a.tiles_y = (Hout + TH - 1) / TH;
    if (EPI & EPI_RES) a.res = res;
    float* dww; CK_(hipMalloc(&dww, 9 * Cin * 4)); CK_(hipMemcpy(dww, h.data(), 9 * Cin * 4, hipMemcpyHostToDevice)); a.dw_w = dww;
    float* xn; CK_(hipMalloc(&xn, nin * 4)); if (PRO == PRO_GN_DW) a.out_xn = xn;
    float* zeros; CK_(hipMalloc(&zeros, 4096)); CK_(hipMemset(zeros, 0, 4096)); a.tbias = zeros;
    constexpr int NT = 32 * NB * WN; a.n_ct = (Cout + NT - 1) / NT;
    const int np = a.tiles_x * a.tiles_y * a.n_ct;
    CK_(hipMalloc(&st, (size_t)B * 64 * 16)); CK_(hipMalloc(&sto, (size_t)B * np * 16));
    std::vector<double> hs((size_t)B * 64 * 2); for (size_t i = 0; i < hs.size(); i += 2) { hs[i] = 10.0; hs[i + 1] = 5000.0; }
    CK_(hipMemcpy(st, hs.data(), hs.size() * 8, hipMemcpyHostToDevice));
    a.st0 = st; a.np0 = 64; a.st_out = sto;
    long long* dbg = nullptr; if (ABL & 16) { CK_(hipMalloc(&dbg, 1024 * 128 * 8)); CK_(hipMemset(dbg, 0, 1024 * 128 * 8)); a.dbg = dbg; }
    const long nwork = (long)B * a.tiles_x * a.tiles_y * a.n_ct;
    const long cap = 256L * wg_per_cu;
    dim3 grid((unsigned)(nwork < cap ? nwork : cap));
    ConvKernelFnT fn = conv_mfma_kernel<KS, S, U, TH, TW, CKc, WM, WN, MB, NB, PRO, 1, EPI, ABL, MATH>;
    const size_t smem = conv_smem_bytes<KS, S, U, TH, TW, CKc, NB * WN, PRO, WM * WN, MATH>() + conv_smem_extra(PRO, n_chunks, CKc, a.n_ct * NT);
    if (smem > 65536) CK_(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipEvent_t e0, e1; CK_(hipEventCreate(&e0)); CK_(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(fn, grid, dim3(64 * WM * WN), smem, 0, a);
    CK_(hipDeviceSynchronize());
    const int iters = 20;
    CK_(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(fn, grid, dim3(64 * WM * WN), smem, 0, a);
    CK_(hipEventRecord(e1, 0)); CK_(hipEventSynchronize(e1));
    float ms; CK_(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, flop = 2.0 * B * Hout * Wout * Cout * Cin * KS * KS;
    if (ABL & 16) {
        std::vector<long long> hd(1024 * 128); CK_(hipMemcpy(hd.data(), dbg, hd.size() * 8, hipMemcpyDeviceToHost));
        for (int blk : {0, 1, 2, 300, 511}) { printf("  block %d stamps (cycles; entry, tables filled, first loads issued, first stage staged | 5 per stage: start, loads issued, mfma done, epilogue done, staged):\n   ", blk); long long t0 = hd[blk * 128];
            for (int i = 0; i < 126 && hd[blk * 128 + i]; ++i) printf(" %lld%s", hd[blk * 128 + i] - t0, (i % 5 == 3) ? " |" : ""); printf("\n"); }
    }
    printf("%-46s abl=%2d wg/cu=%d grid=%5u smem=%6zu  %8.1f us  %6.1f TF\n", name, ABL, wg_per_cu, grid.x, smem, us, flop / us / 1e6);
    if (g_keep_out) {
        std::vector<float> ho(nout); CK_(hipMemcpy(ho.data(), out, nout * 4, hipMemcpyDeviceToHost));
        if (g_last_out.size() == nout) {
            size_t nd = 0; double mx = 0; for (size_t i = 0; i < nout; ++i) { if (memcmp(&ho[i], &g_last_out[i], 4)) { ++nd; mx = std::max(mx, (double)fabsf(ho[i] - g_last_out[i])); } }
            std::vector<double> hst((size_t)B * np * 2); CK_(hipMemcpy(hst.data(), sto, hst.size() * 8, hipMemcpyDeviceToHost));
            double ss = 0; for (double v : hst) ss += v;
            printf("    vs previous run: %zu of %zu outputs differ (max |d| %.3g); out[1000] = %.6f, sum of partials %.6f\n", nd, nout, mx, ho[1000], ss);
        }
        if ((MATH == 3 || MATH == 5) && KS == 3 && S == 1 && !U && (PRO == PRO_GN_SILU || PRO == PRO_NONE)) {
            // host check of 256 outputs in double: weights = (hi + lo) / 2^10 of the packed halves ([n-block][16-channel chunk][tap][plane][half h][cout j][8 cin]),
            // activations rounded as the kernel stages them (x16, hi + lo halves)
            const int nch16 = (Cin + 15) / 16; double emax = 0, vmax = 0;
            const double N = (double)Cin * H * W, mean = 64 * 10.0 / N, var = 64 * 5000.0 / N - mean * mean, rstd = 1.0 / sqrt(var + 1e-5);
            for (int t = 0; t < 256; ++t) {
                const int b = (t * 7) % B, y = (t * 13 + (t >> 4)) % H, x = (t * 29 + 3) % W, co = (t * 5) % Cout;
                double acc = 0;
                for (int tap = 0; tap < 9; ++tap) {
                    const int iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
                    if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                    for (int ci = 0; ci < Cin; ++ci) {
                        double v = h[((size_t)(b * H + iy) * W + ix) * Cin + ci];
                        if (PRO == PRO_GN_SILU) { const float g = h[ci] * (float)rstd, bb = h[ci] - (float)mean * g; float xn = fmaf((float)v, g, bb); v = xn / (1.0 + exp(-(double)xn)); }
                        const float vs = (float)v * 16.0f; const _Float16 ah = (_Float16)vs; const _Float16 al = (_Float16)(vs - (float)ah);
                        const int ch = ci / 16, hh = (ci % 16) / 8, tt = ci % 8, nbi = co / 32, j = co % 32;
                        const size_t fl = ((((size_t)nbi * nch16 + ch) * 9 + tap) * 2) * 256 + (size_t)(hh * 32 + j) * 4;
                        const double wh = (double)(float)hw[fl * 2 + tt], wl = (double)(float)hw[(fl + 256) * 2 + tt];
                        acc += (wh * ((double)(float)ah + (double)(float)al) + wl * (double)(float)ah) ;  // hi*hi + hi*lo + lo*hi (lo*lo dropped, as the kernel does)
                    }
                }
                double ref = acc / 16384.0 + h[co];
                if (EPI & EPI_RES) ref += h[((size_t)(b * Hout + y) * Wout + x) * Cout + co];
                if (EPI & EPI_SILU) ref = ref / (1.0 + exp(-ref));
                const double got = g_last_out.size() ? 0 : 0; (void)got;
                const double e = fabs(ref - ho[((size_t)(b * Hout + y) * Wout + x) * Cout + co]);
                emax = std::max(emax, e); vmax = std::max(vmax, fabs(ref));
            }
            printf("    host check (256 outputs, fp64): max |err| %.3g, max |ref| %.3g\n", emax, vmax);
        }
        g_last_out.swap(ho);
    }
    hipFree(in); hipFree(w); hipFree(out); hipFree(res); hipFree(gamma); hipFree(beta); hipFree(bias); hipFree(st); hipFree(sto);
}

int main(int argc, char** argv) {
    const int B = 64;
    if (argc > 1 && argv[1][0] == 'q') {  // the fused q conv (GN + depthwise 3x3 + 1x1): 4 vs 8 waves, f32 vs bf16x3
        run<1, 1, 0, 8, 16, 32, 4, 1, 1, 2, PRO_GN_DW, 0, 0, 0>("1x1 gn_dw 64->64 @64^2 (8x16,NT64,4w) f32", B, 64, 64, 64, 64, 1);
        run<1, 1, 0, 8, 16, 32, 4, 1, 1, 2, PRO_GN_DW, 0, 0, 1>("1x1 gn_dw 64->64 @64^2 (8x16,NT64,4w) x3", B, 64, 64, 64, 64, 1);
        run<1, 1, 0, 8, 16, 32, 4, 2, 1, 1, PRO_GN_DW, 0, 0, 0>("1x1 gn_dw 64->64 @64^2 (8x16,NT64,8w) f32", B, 64, 64, 64, 64, 1);
        run<1, 1, 0, 8, 16, 32, 4, 2, 1, 1, PRO_GN_DW, 0, 0, 1>("1x1 gn_dw 64->64 @64^2 (8x16,NT64,8w) x3", B, 64, 64, 64, 64, 1);
        run<1, 1, 0, 8, 16, 32, 4, 1, 1, 4, PRO_GN_DW, 0, 0, 1>("1x1 gn_dw 128->128 @32^2 (8x16,NT128,4w) x3", B, 32, 32, 128, 128, 1);
        run<1, 1, 0, 8, 16, 32, 4, 2, 1, 2, PRO_GN_DW, 0, 0, 1>("1x1 gn_dw 128->128 @32^2 (8x16,NT128,8w) x3", B, 32, 32, 128, 128, 1);
        run<1, 1, 0, 8, 16, 32, 4, 2, 1, 2, PRO_GN_DW, 0, 0, 0>("1x1 gn_dw 128->128 @32^2 (8x16,NT128,8w) f32", B, 32, 32, 128, 128, 1);
        run<1, 1, 0, 8, 16, 32, 4, 2, 1, 1, PRO_NONE, 0, 0, 1>("1x1 64->64 @64^2 (8x16,NT64,8w) x3", B, 64, 64, 64, 64, 1);
        run<1, 1, 0, 8, 16, 32, 4, 1, 1, 2, PRO_NONE, 0, 0, 1>("1x1 64->64 @64^2 (8x16,NT64,4w) x3", B, 64, 64, 64, 64, 2);
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'm') {  // which memory stream costs what (bf16x3, 16x16 x 32, 8 waves)
#define RUNM(ABLV) run<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_GN_SILU, ABLV, 0, 1>("3x3 gn_silu 32->32 @64^2 x3 8w", B, 64, 64, 32, 32, 1)
        RUNM(0); RUNM(32); RUNM(96); RUNM(14); RUNM(46);
#define RUNM2(ABLV) run<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_NONE, ABLV, 0, 1>("3x3 64->64 @64^2 x3 8w", B, 64, 64, 64, 64, 1)
        RUNM2(0); RUNM2(32); RUNM2(96);
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'a') {  // round 2: what a bf16x3 3x3 launch is made of (8 waves, 16x16 x 32)
#define RUNA(ABLV) run<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_GN_SILU, ABLV, 0, 1>("3x3 gn_silu 32->32 @64^2 x3 8w", B, 64, 64, 32, 32, 1)
        RUNA(0); RUNA(1); RUNA(128); RUNA(129); RUNA(256); RUNA(14); RUNA(270); RUNA(143); RUNA(399);
        RUNA(2); RUNA(4); RUNA(8); RUNA(6); RUNA(10); RUNA(12);
#define RUNB(ABLV) run<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_NONE, ABLV, 0, 1>("3x3 64->32 @64^2 x3 8w (ffn.2)", B, 64, 64, 64, 32, 1)
        RUNB(0); RUNB(1); RUNB(128); RUNB(129); RUNB(256); RUNB(14); RUNB(399);
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'f') {  // round 4: what an f16x2 (MATH = 3) 3x3 launch is made of, and the 4-wave 8x16 tiling with two workgroups per CU
#define RUNF(ABLV) run<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_GN_SILU, ABLV, 0, 3>("3x3 gn_silu 32->32 @64^2 f16x2 8w 16x16", B, 64, 64, 32, 32, 1)
        RUNF(0); RUNF(1); RUNF(128); RUNF(129); RUNF(256); RUNF(2); RUNF(4); RUNF(8); RUNF(14); RUNF(270); RUNF(399);
        run<3, 1, 0, 8, 16, 16, 4, 1, 1, 1, PRO_GN_SILU, 0, 0, 3>("3x3 gn_silu 32->32 @64^2 f16x2 4w 8x16 2wg/cu", B, 64, 64, 32, 32, 2);
        run<3, 1, 0, 8, 16, 16, 4, 1, 1, 1, PRO_GN_SILU, 0, 0, 3>("3x3 gn_silu 32->32 @64^2 f16x2 4w 8x16 1wg/cu", B, 64, 64, 32, 32, 1);
        run<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_GN_SILU, 0, EPI_RES, 3>("3x3 gn_silu+res 32->32 @64^2 f16x2 8w", B, 64, 64, 32, 32, 1);
        run<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_GN_SILU, 0, 0, 3>("3x3 gn_silu 64->64 @32^2 f16x2 8w 16x16", B, 32, 32, 64, 64, 1);
        run<3, 1, 0, 8, 16, 16, 4, 1, 1, 1, PRO_GN_SILU, 0, 0, 3>("3x3 gn_silu 64->64 @32^2 f16x2 4w 8x16 2wg/cu", B, 32, 32, 64, 64, 2);
        run<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_NONE, 0, EPI_SILU, 3>("3x3 silu 32->64 @64^2 f16x2 8w (ffn.0)", B, 64, 64, 32, 64, 1);
        run<3, 1, 0, 16, 16, 16, 8, 1, 1, 2, PRO_NONE, 0, EPI_SILU, 3>("3x3 silu 32->64 @64^2 f16x2 8w NT64 (ffn.0)", B, 64, 64, 32, 64, 1);
        run<3, 1, 0, 8, 16, 16, 4, 1, 1, 1, PRO_NONE, 0, EPI_SILU, 3>("3x3 silu 32->64 @64^2 f16x2 4w 8x16 2wg/cu (ffn.0)", B, 64, 64, 32, 64, 2);
        run<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_NONE, 0, EPI_RES, 3>("3x3 res 64->32 @64^2 f16x2 8w (ffn.23)", B, 64, 64, 64, 32, 1);
        run<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_GN_SILU, 16, 0, 3>("3x3 gn_silu 32->32 @64^2 f16x2 8w stamps", B, 64, 64, 32, 32, 1);
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'r') {  // round 5: resident weights + one 32-channel stage per item (MATH = 5) against MATH = 3 with 16-channel stages
        g_keep_out = true;
        srand(7); run<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_GN_SILU, 0, 0, 3>("3x3 gn_silu 32->32 @64^2 f16x2 CK16", B, 64, 64, 32, 32, 1);
        srand(7); run<3, 1, 0, 16, 16, 32, 8, 1, 1, 1, PRO_GN_SILU, 0, 0, 5>("3x3 gn_silu 32->32 @64^2 f16x2 CK32 resident", B, 64, 64, 32, 32, 1);
        srand(9); run<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_GN_SILU, 0, EPI_RES, 3>("3x3 gn_silu+res 32->32 @64^2 f16x2 CK16", B, 64, 64, 32, 32, 1);
        srand(9); run<3, 1, 0, 16, 16, 32, 8, 1, 1, 1, PRO_GN_SILU, 0, EPI_RES, 5>("3x3 gn_silu+res 32->32 @64^2 f16x2 CK32 resident", B, 64, 64, 32, 32, 1);
        srand(11); run<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_NONE, 0, 0, 3>("3x3 32->32 @64^2 f16x2 CK16 (no prologue)", B, 64, 64, 32, 32, 1);
        srand(11); run<3, 1, 0, 16, 16, 32, 8, 1, 1, 1, PRO_NONE, 0, 0, 5>("3x3 32->32 @64^2 f16x2 CK32 resident (no prologue)", B, 64, 64, 32, 32, 1);
        g_keep_out = false;
        // B = 8 (the GF2 strong-scaling share): latency floor of one launch
        run<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_GN_SILU, 0, 0, 3>("B=8 3x3 gn_silu 32->32 @64^2 f16x2 CK16", 8, 64, 64, 32, 32, 1);
        run<3, 1, 0, 16, 16, 32, 8, 1, 1, 1, PRO_GN_SILU, 0, 0, 5>("B=8 3x3 gn_silu 32->32 @64^2 f16x2 CK32 resident", 8, 64, 64, 32, 32, 1);
#define RUNR(ABLV) run<3, 1, 0, 16, 16, 32, 8, 1, 1, 1, PRO_GN_SILU, ABLV, 0, 5>("3x3 gn_silu 32->32 @64^2 f16x2 CK32 resident", B, 64, 64, 32, 32, 1)
        RUNR(1); RUNR(128); RUNR(129); RUNR(256); RUNR(2); RUNR(4); RUNR(14); RUNR(270); RUNR(399);
        run<3, 1, 0, 16, 16, 32, 8, 1, 1, 1, PRO_GN_SILU, 16, 0, 5>("3x3 gn_silu 32->32 @64^2 f16x2 CK32 resident stamps", B, 64, 64, 32, 32, 1);
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'i') {  // round 5: instruction-cache effect of ALTERNATING kernels (the step never runs the same instantiation twice in a row)
        const int H = 64, W = 64, Cin = 32, Cout = 32;
        const size_t nin = (size_t)B * H * W * Cin, nout = (size_t)B * H * W * Cout, nw = 4 * 2 * 9 * 3 * 256;
        float *in, *w, *out, *gamma, *beta, *bias, *res, *zeros; double *st, *sto;
        CK_(hipMalloc(&in, nin * 4)); CK_(hipMalloc(&w, nw * 4)); CK_(hipMalloc(&out, nout * 4)); CK_(hipMalloc(&res, nout * 4)); CK_(hipMalloc(&gamma, 4096)); CK_(hipMalloc(&beta, 4096));
        CK_(hipMalloc(&bias, 4096)); CK_(hipMalloc(&zeros, 4096)); CK_(hipMemset(zeros, 0, 4096)); CK_(hipMalloc(&st, (size_t)B * 64 * 16)); CK_(hipMalloc(&sto, (size_t)B * 64 * 16));
        std::vector<float> hh(nin); for (auto& v : hh) v = (rand() % 2001 - 1000) * 1e-3f;
        CK_(hipMemcpy(in, hh.data(), nin * 4, hipMemcpyHostToDevice)); CK_(hipMemcpy(res, hh.data(), nout * 4, hipMemcpyHostToDevice));
        CK_(hipMemcpy(gamma, hh.data(), 4096, hipMemcpyHostToDevice)); CK_(hipMemcpy(beta, hh.data(), 4096, hipMemcpyHostToDevice)); CK_(hipMemcpy(bias, hh.data(), 4096, hipMemcpyHostToDevice));
        std::vector<_Float16> hw(nw * 2); for (auto& v : hw) v = (_Float16)((rand() % 2001 - 1000) * 1e-3f);
        CK_(hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice));
        std::vector<double> hs((size_t)B * 64 * 2); for (size_t i = 0; i < hs.size(); i += 2) { hs[i] = 10.0; hs[i + 1] = 5000.0; }
        CK_(hipMemcpy(st, hs.data(), hs.size() * 8, hipMemcpyHostToDevice));
        ConvArgs a{}; a.in0 = in; a.c0 = Cin; a.B = B; a.Hin = H; a.Win = W; a.Hout = H; a.Wout = W; a.Cout = Cout; a.w = w; a.n_chunks = 2; a.bias = bias; a.gamma = gamma; a.beta = beta;
        a.out = out; a.tiles_x = 4; a.tiles_y = 4; a.res = res; a.tbias = zeros; a.n_ct = 1; a.st0 = st; a.np0 = 64; a.st_out = sto;
        ConvKernelFnT ks[4] = {conv_mfma_kernel<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_GN_SILU, 1, 0, 0, 3>, conv_mfma_kernel<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_GN_SILU, 1, EPI_RES, 0, 3>,
                               conv_mfma_kernel<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_NONE, 1, EPI_SILU, 0, 3>, conv_mfma_kernel<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_NONE, 1, EPI_RES, 0, 3>};
        size_t sm[4] = {conv_smem_bytes<3, 1, 0, 16, 16, 16, 1, PRO_GN_SILU, 8, 3>() + conv_smem_extra(PRO_GN_SILU, 2, 16, 32), 0, conv_smem_bytes<3, 1, 0, 16, 16, 16, 1, PRO_NONE, 8, 3>() + conv_smem_extra(PRO_NONE, 2, 16, 32), 0};
        sm[1] = sm[0]; sm[3] = sm[2];
        for (int k = 0; k < 4; ++k) CK_(hipFuncSetAttribute((const void*)ks[k], hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm[k]));
        hipEvent_t e0, e1; CK_(hipEventCreate(&e0)); CK_(hipEventCreate(&e1));
        auto timeseq = [&](const char* name, std::vector<int> seq) {
            for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(ks[seq[i % seq.size()]], dim3(256), dim3(512), sm[seq[i % seq.size()]], 0, a);
            CK_(hipDeviceSynchronize());
            const int iters = 120;
            CK_(hipEventRecord(e0, 0));
            for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(ks[seq[i % seq.size()]], dim3(256), dim3(512), sm[seq[i % seq.size()]], 0, a);
            CK_(hipEventRecord(e1, 0)); CK_(hipEventSynchronize(e1));
            float ms; CK_(hipEventElapsedTime(&ms, e0, e1));
            printf("%-40s %7.2f us per launch\n", name, ms * 1e3 / iters);
        };
        for (int rep = 0; rep < 2; ++rep) {
            timeseq("A only (gn_silu)", {0}); timeseq("B only (gn_silu + res)", {1}); timeseq("C only (silu epilogue)", {2}); timeseq("D only (res, no prologue)", {3});
            timeseq("A B alternating", {0, 1}); timeseq("A B C D round robin", {0, 1, 2, 3}); timeseq("A A B B C C D D", {0, 0, 1, 1, 2, 2, 3, 3});
        }
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'x') {  // bf16x3 (MATH = 1) against the exact-fp32 MFMA
        run<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_GN_SILU, 0>("3x3 gn_silu 32->32 @64^2 (16x16,NT32,8w) f32", B, 64, 64, 32, 32, 1);
        run<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_GN_SILU, 0, 0, 1>("3x3 gn_silu 32->32 @64^2 (16x16,NT32,8w) bf16x3", B, 64, 64, 32, 32, 1);
        run<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_GN_SILU, 14, 0, 1>("3x3 gn_silu 32->32 @64^2 (16x16,NT32,8w) bf16x3", B, 64, 64, 32, 32, 1);
        run<3, 1, 0, 8, 16, 16, 4, 1, 1, 1, PRO_GN_SILU, 0, 0, 1>("3x3 gn_silu 32->32 @64^2 (8x16,NT32,4w) bf16x3", B, 64, 64, 32, 32, 1);
        run<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_GN_SILU, 0, 0, 1>("3x3 gn_silu 64->64 @32^2 (16x16,NT32,8w) bf16x3", B, 32, 32, 64, 64, 1);
        run<3, 1, 0, 16, 16, 16, 8, 1, 1, 2, PRO_NONE, 0>("3x3 64->64 @64^2 (16x16,NT64,8w) f32", B, 64, 64, 64, 64, 1);
        run<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_NONE, 0, 0, 1>("3x3 64->64 @64^2 (16x16,NT32,8w) bf16x3", B, 64, 64, 64, 64, 1);
        run<3, 1, 0, 8, 16, 16, 4, 1, 1, 1, PRO_GN_SILU, 0, 0, 1>("3x3 gn_silu 64->64 @16^2 (8x16,NT32,4w) bf16x3", B, 16, 16, 64, 64, 1);
        run<3, 1, 0, 8, 8, 16, 2, 2, 1, 1, PRO_GN_SILU, 0, 0, 1>("3x3 gn_silu 128->128 @8^2 (8x8,NT64,4w) bf16x3", B, 8, 8, 128, 128, 1);
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'b') {  // big per-wave register tiles, one wave per SIMD
        run<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_GN_SILU, 0>("3x3 gn_silu 32->32 @64^2 (16x16,NT32,8w)", B, 64, 64, 32, 32, 1);
        run<3, 1, 0, 16, 16, 16, 4, 1, 2, 1, PRO_GN_SILU, 0>("3x3 gn_silu 32->32 @64^2 (16x16,NT32,4w MB2)", B, 64, 64, 32, 32, 1);
        run<3, 1, 0, 16, 16, 16, 8, 1, 1, 2, PRO_GN_SILU, 0>("3x3 gn_silu 64->64 @32^2 (16x16,NT64,8w)", B, 32, 32, 64, 64, 1);
        run<3, 1, 0, 16, 16, 16, 4, 1, 2, 2, PRO_GN_SILU, 0>("3x3 gn_silu 64->64 @32^2 (16x16,NT64,4w MB2NB2)", B, 32, 32, 64, 64, 1);
        run<3, 1, 0, 16, 16, 16, 8, 1, 1, 2, PRO_NONE, 0>("3x3 64->64 @64^2 (16x16,NT64,8w)", B, 64, 64, 64, 64, 1);
        run<3, 1, 0, 16, 16, 16, 4, 1, 2, 2, PRO_NONE, 0>("3x3 64->64 @64^2 (16x16,NT64,4w MB2NB2)", B, 64, 64, 64, 64, 1);
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'd') {  // delayed-start (anti-phase) experiment
        run<3, 1, 0, 8, 16, 16, 4, 1, 1, 1, PRO_GN_SILU, 0>("3x3 gn_silu 32->32 @64^2 (8x16,NT32)", B, 64, 64, 32, 32, 2);
        run<3, 1, 0, 8, 16, 16, 4, 1, 1, 1, PRO_GN_SILU, 32>("3x3 gn_silu 32->32 @64^2 (8x16,NT32) delay40", B, 64, 64, 32, 32, 2);
        run<3, 1, 0, 8, 16, 16, 4, 1, 1, 1, PRO_GN_SILU, 96>("3x3 gn_silu 32->32 @64^2 (8x16,NT32) delay60", B, 64, 64, 32, 32, 2);
        run<3, 1, 0, 8, 16, 16, 4, 1, 1, 1, PRO_GN_SILU, 48>("3x3 gn_silu 32->32 @64^2 (8x16,NT32) delay40 stamps", B, 64, 64, 32, 32, 2);
        run<3, 1, 0, 8, 16, 16, 4, 1, 1, 1, PRO_NONE, 0>("3x3 64->64 @64^2 (8x16,NT32)", B, 64, 64, 64, 64, 2);
        run<3, 1, 0, 8, 16, 16, 4, 1, 1, 1, PRO_NONE, 32>("3x3 64->64 @64^2 (8x16,NT32) delay40", B, 64, 64, 64, 64, 2);
        run<3, 1, 0, 8, 16, 16, 4, 1, 1, 1, PRO_NONE, 96>("3x3 64->64 @64^2 (8x16,NT32) delay60", B, 64, 64, 64, 64, 2);
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'n') {  // nontemporal-store experiment (ABL 64 = plain stores)
        run<3, 1, 0, 8, 16, 16, 4, 1, 1, 1, PRO_GN_SILU, 0>("3x3 gn_silu 32->32 @64^2 (8x16,NT32) nt", B, 64, 64, 32, 32, 2);
        run<3, 1, 0, 8, 16, 16, 4, 1, 1, 1, PRO_GN_SILU, 64>("3x3 gn_silu 32->32 @64^2 (8x16,NT32) plain", B, 64, 64, 32, 32, 2);
        run<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_GN_SILU, 0>("3x3 gn_silu 32->32 @64^2 (16x16,NT32,8w) nt", B, 64, 64, 32, 32, 1);
        run<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_GN_SILU, 64>("3x3 gn_silu 32->32 @64^2 (16x16,NT32,8w) plain", B, 64, 64, 32, 32, 1);
        run<1, 1, 0, 8, 16, 32, 4, 1, 1, 2, PRO_NONE, 0>("1x1 64->64 @64^2 (8x16,NT64) nt", B, 64, 64, 64, 64, 2);
        run<1, 1, 0, 8, 16, 32, 4, 1, 1, 2, PRO_NONE, 64>("1x1 64->64 @64^2 (8x16,NT64) plain", B, 64, 64, 64, 64, 2);
        run<3, 1, 0, 16, 16, 16, 8, 1, 1, 2, PRO_NONE, 0>("3x3 64->64 @64^2 (16x16,NT64,8w) nt", B, 64, 64, 64, 64, 1);
        run<3, 1, 0, 16, 16, 16, 8, 1, 1, 2, PRO_NONE, 64>("3x3 64->64 @64^2 (16x16,NT64,8w) plain", B, 64, 64, 64, 64, 1);
        return 0;
    }
    if (argc > 1 && argv[1][0] == 's') {  // stamp mode
        run<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_GN_SILU, 16, 0, 1>("3x3 gn_silu 32->32 @64^2 x3 8w", B, 64, 64, 32, 32, 1);
        run<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_GN_SILU, 30, 0, 1>("3x3 gn_silu 32->32 @64^2 x3 8w", B, 64, 64, 32, 32, 1);
        return 0;
    }
    if (argc > 1) {  // PMC mode: few kernels, distinct template instantiations
        run<3, 1, 0, 8, 16, 16, 4, 1, 1, 1, PRO_GN_SILU, 0>("3x3 gn_silu 32->32 @64^2 (8x16,NT32)", B, 64, 64, 32, 32, 2);
        run<3, 1, 0, 8, 16, 16, 4, 1, 1, 1, PRO_GN_SILU, 14>("3x3 gn_silu 32->32 @64^2 (8x16,NT32)", B, 64, 64, 32, 32, 2);
        run<3, 1, 0, 8, 16, 16, 4, 1, 1, 1, PRO_NONE, 0>("3x3 64->64 @64^2 (8x16,NT32)", B, 64, 64, 64, 64, 2);
        run<1, 1, 0, 8, 16, 32, 4, 1, 1, 2, PRO_GN_DW, 0>("1x1 gn_dw 64->64 @64^2 (8x16,NT64)", B, 64, 64, 64, 64, 2);
        return 0;
    }
    run<3, 1, 0, 8, 16, 16, 4, 1, 1, 1, PRO_GN_SILU, 0>("3x3 gn_silu 32->32 @64^2 (8x16,NT32)", B, 64, 64, 32, 32, 2);
    run<3, 1, 0, 8, 16, 16, 4, 1, 1, 1, PRO_GN_SILU, 14>("3x3 gn_silu 32->32 @64^2 (8x16,NT32)", B, 64, 64, 32, 32, 2);
    run<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_GN_SILU, 0>("3x3 gn_silu 32->32 @64^2 (16x16,NT32,8w)", B, 64, 64, 32, 32, 1);
    run<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_GN_SILU, 14>("3x3 gn_silu 32->32 @64^2 (16x16,NT32,8w)", B, 64, 64, 32, 32, 1);
    run<3, 1, 0, 8, 16, 16, 4, 1, 1, 1, PRO_GN_SILU, 0>("3x3 gn_silu 64->64 @32^2 (8x16,NT32)", B, 32, 32, 64, 64, 2);
    run<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_GN_SILU, 0>("3x3 gn_silu 64->64 @32^2 (16x16,NT32,8w)", B, 32, 32, 64, 64, 1);
    run<3, 1, 0, 16, 16, 16, 8, 1, 1, 2, PRO_GN_SILU, 0>("3x3 gn_silu 64->64 @32^2 (16x16,NT64,8w)", B, 32, 32, 64, 64, 1);
    run<3, 1, 0, 8, 16, 16, 4, 1, 1, 1, PRO_NONE, 0>("3x3 64->64 @64^2 (8x16,NT32)", B, 64, 64, 64, 64, 2);
    run<3, 1, 0, 16, 16, 16, 8, 1, 1, 1, PRO_NONE, 0>("3x3 64->64 @64^2 (16x16,NT32,8w)", B, 64, 64, 64, 64, 1);
    run<3, 1, 0, 16, 16, 16, 8, 1, 1, 2, PRO_NONE, 0>("3x3 64->64 @64^2 (16x16,NT64,8w)", B, 64, 64, 64, 64, 1);
    run<3, 1, 0, 8, 8, 16, 2, 2, 1, 1, PRO_GN_SILU, 0>("3x3 gn_silu 128->128 @8^2 (8x8,NT64)", B, 8, 8, 128, 128, 2);
    run<3, 1, 0, 8, 16, 16, 4, 1, 1, 1, PRO_GN_SILU, 0>("3x3 gn_silu 64->64 @16^2 (8x16,NT32)", B, 16, 16, 64, 64, 2);
    run<1, 1, 0, 8, 16, 32, 4, 1, 1, 2, PRO_NONE, 0>("1x1 64->64 @64^2 (8x16,NT64)", B, 64, 64, 64, 64, 2);
    run<1, 1, 0, 8, 16, 32, 4, 1, 1, 2, PRO_GN_DW, 0>("1x1 gn_dw 64->64 @64^2 (8x16,NT64)", B, 64, 64, 64, 64, 2);
    run<1, 1, 0, 8, 16, 32, 4, 1, 1, 4, PRO_GN_DW, 0>("1x1 gn_dw 128->128 @32^2 (8x16,NT128)", B, 32, 32, 128, 128, 1);
    run<1, 1, 0, 8, 16, 32, 4, 1, 1, 4, PRO_GN_SILU, 0>("1x1 gn_silu 256->128 @32^2 (8x16,NT128)", B, 32, 32, 256, 128, 2);
    return 0;
}
