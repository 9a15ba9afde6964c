// Instantiations + launcher of the fused linear-attention kernel (kernels_lafuse.h) -- a translation unit of its own so that it
// builds in parallel with the conv kernel families.
#include "ddif_plan.h"
#include "kernels_lafuse.h"
#include "kernels_lafuse8.h"

namespace ddif {

namespace {
template <int TH, int TW, int NBQ, int NBA>
int la_launch1(const LaFuseArgs& a, int grid, hipStream_t s, bool prepare_only) {
    using G = LaFuseGeom<TH, TW, NBQ, NBA>;
    auto fn = linattn_fused_kernel<TH, TW, NBQ, NBA>;
    if (prepare_only) {
        DDIF_HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::smem));
        return 0;
    }
    hipLaunchKernelGGL(fn, dim3(grid), dim3(2 * TH * TW), G::smem, s, a);
    return 0;
}
template <int TH, int TW>
int la_launch_t(const LaFuseArgs& a, int nbq, int nba, int grid, hipStream_t s, bool prep) {
    switch (nbq * 10 + nba) {
        case 21: return la_launch1<TH, TW, 2, 1>(a, grid, s, prep);
        case 31: return la_launch1<TH, TW, 3, 1>(a, grid, s, prep);
        case 41: return la_launch1<TH, TW, 4, 1>(a, grid, s, prep);
        case 22: return la_launch1<TH, TW, 2, 2>(a, grid, s, prep);
        case 32: return la_launch1<TH, TW, 3, 2>(a, grid, s, prep);
        case 42: return la_launch1<TH, TW, 4, 2>(a, grid, s, prep);
        case 62:  // 128 + 64 channels -> 64: the first decoder block of the 16 x 16 level (round 6; 42 spilled registers, still one launch instead of three)
            if constexpr (TH == 16) return la_launch1<TH, TW, 6, 2>(a, grid, s, prep);
            [[fallthrough]];
        default: return fail(DDIF_ERR_INVALID, "linattn_fused: no instantiation for %d q blocks / %d output blocks", nbq, nba);
    }
}
}  // namespace

// shapes the fused kernel carries: whole columns of 64 or 32 rows, 64 / 96 / 128 feature channels, 32 / 64 output channels
bool lafuse_supported(int H, int fea, int dout) {
    static const bool la6 = [] { const char* e = getenv("DDIF_LA6"); return !e || atoi(e) != 0; }();  // DDIF_LA6=0: the 192-channel block of the 16 x 16 level as three launches (rounds 4-5)
    if (H == 16 && fea == 192 && dout == 64 && la6) return true;
    return (H == 64 || H == 32 || H == 16) && fea % 32 == 0 && fea >= 64 && fea <= 128 && dout % 32 == 0 && dout >= 32 && dout <= 64;
}
// image columns per workgroup: 256 pixels on eight wavefronts (nw = 8) or 128 on four (nw = 4, round 6: chosen by the plan when the eight-wave grid would leave CUs idle)
int lafuse_strip(int H, int nw) { return 32 * nw / H; }
int lafuse_launch(const LaFuseArgs& a, int grid, hipStream_t s, bool prepare_only, int nw) {
    const int nbq = (a.c0 + a.c1) / 32, nba = a.dout / 32;
    if (nw == 4) {
        if (a.H == 64) return la_launch_t<64, 2>(a, nbq, nba, grid, s, prepare_only);
        if (a.H == 32) return la_launch_t<32, 4>(a, nbq, nba, grid, s, prepare_only);
        if (a.H == 16) return la_launch_t<16, 8>(a, nbq, nba, grid, s, prepare_only);
    } else {
        if (a.H == 64) return la_launch_t<64, 4>(a, nbq, nba, grid, s, prepare_only);
        if (a.H == 32) return la_launch_t<32, 8>(a, nbq, nba, grid, s, prepare_only);
        if (a.H == 16) return la_launch_t<16, 16>(a, nbq, nba, grid, s, prepare_only);  // a 16 x 16 sample = one workgroup
    }
    return fail(DDIF_ERR_INVALID, "linattn_fused: H = %d", a.H);
}

// the 8 x 8 level (kernels_lafuse8.h): half a sample per workgroup, 256 (128 + 128) or 192 (128 + 64) feature channels, 128 output channels
bool lafuse8_supported(int H, int W, int c0, int c1, int dout) {
    return H == 8 && W == 8 && c0 % 64 == 0 && c1 % 64 == 0 && (c0 + c1 == 256 || c0 + c1 == 192) && dout == 128;
}
int lafuse8_launch(const LaFuseArgs& a, int grid, hipStream_t s, bool prepare_only) {
    const int nbq = (a.c0 + a.c1) / 32;
    if (nbq == 8) {
        auto fn = linattn8_fused_kernel<8>;
        if (prepare_only) {
            DDIF_HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LaFuse8Geom<8>::smem));
            return 0;
        }
        hipLaunchKernelGGL(fn, dim3(grid), dim3(512), LaFuse8Geom<8>::smem, s, a);
        return 0;
    }
    if (nbq == 6) {
        auto fn = linattn8_fused_kernel<6>;
        if (prepare_only) {
            DDIF_HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LaFuse8Geom<6>::smem));
            return 0;
        }
        hipLaunchKernelGGL(fn, dim3(grid), dim3(512), LaFuse8Geom<6>::smem, s, a);
        return 0;
    }
    return fail(DDIF_ERR_INVALID, "linattn8_fused: %d feature channels", a.c0 + a.c1);
}

}  // namespace ddif
