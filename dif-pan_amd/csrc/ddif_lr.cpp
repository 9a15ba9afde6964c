// Instantiations of the low-resolution conv kernel (kernels_lr.h) -- a translation unit of its own so that it builds
// in parallel with the main conv kernel family (ddif_plan.cpp).
#include "ddif_plan.h"
#include "kernels_lr.h"
#include "kernels_attn.h"

namespace ddif {

namespace {
template <int KS, int MB, int PRO, int EPI, bool F16 = false, bool B1 = false, bool ROWS = false>
ConvVariant lr_variant1(const char* name) {
    ConvVariant v;
    v.fn = conv_lr_kernel<KS, MB, PRO, EPI, 0, F16, B1, ROWS>;
    using G = LrGeom<KS, MB, PRO, (EPI & EPI_COLST) != 0, F16, B1>;
    v.smem = G::smem;
    v.th = G::TH;
    v.tw = G::TW;
    v.nt = 32;
    v.nthr = 256;
    v.x3 = true;
    v.f16 = F16;
    v.b1 = B1;
    v.lr = true;
    v.name = name;
    return v;
}
// the whole-width staging of the 3x3 convs (kernels_lr.h ROWS; f16x2 and bf16x3): same tile, same LDS footprint, same results
template <int MB, bool F16>
ConvVariant lr_rows_for(int pro, int epi) {
    if (pro == PRO_GN_SILU && epi == 0) return lr_variant1<3, MB, PRO_GN_SILU, 0, F16, false, true>("lr3x3_gn_silu_rows");
    if (pro == PRO_GN_SILU && epi == EPI_RES) return lr_variant1<3, MB, PRO_GN_SILU, EPI_RES, F16, false, true>("lr3x3_gn_silu_res_rows");
    if (pro == PRO_NONE && epi == EPI_SILU) return lr_variant1<3, MB, PRO_NONE, EPI_SILU, F16, false, true>("lr3x3_silu_rows");
    if (pro == PRO_NONE && epi == 0) return lr_variant1<3, MB, PRO_NONE, 0, F16, false, true>("lr3x3_rows");
    if (pro == PRO_NONE && epi == EPI_RES) return lr_variant1<3, MB, PRO_NONE, EPI_RES, F16, false, true>("lr3x3_res_rows");
    return ConvVariant();
}
template <int MB>
ConvVariant lr_for(int ks, int pro, int epi, int math) {
#define LRV(KS, PRO, EPI, NAME) (math == MATH_F16X2 ? lr_variant1<KS, MB, PRO, EPI, true>(NAME) : (math == MATH_BF16X1 ? lr_variant1<KS, MB, PRO, EPI, false, true>(NAME) : lr_variant1<KS, MB, PRO, EPI, false>(NAME)))
    if (ks == 3) {
        if (pro == PRO_GN_SILU && epi == 0) return LRV(3, PRO_GN_SILU, 0, "lr3x3_gn_silu");
        if (pro == PRO_GN_SILU && epi == EPI_RES) return LRV(3, PRO_GN_SILU, EPI_RES, "lr3x3_gn_silu_res");
        if (pro == PRO_NONE && epi == EPI_SILU) return LRV(3, PRO_NONE, EPI_SILU, "lr3x3_silu");
        if (pro == PRO_NONE && epi == 0) return LRV(3, PRO_NONE, 0, "lr3x3");
        if (pro == PRO_NONE && epi == EPI_RES) return LRV(3, PRO_NONE, EPI_RES, "lr3x3_res");  // merged ffn[3] o ffn[2] + residual
    } else if (ks == 1) {
        if (pro == PRO_NONE && epi == EPI_FILM) return LRV(1, PRO_NONE, EPI_FILM, "lr1x1_film");
        if (pro == PRO_NONE && epi == EPI_RES) return LRV(1, PRO_NONE, EPI_RES, "lr1x1_res");
        if (pro == PRO_NONE && epi == 0) return LRV(1, PRO_NONE, 0, "lr1x1");
        if (pro == PRO_NONE && epi == EPI_COLST) return LRV(1, PRO_NONE, EPI_COLST, "lr1x1_colstats");
        if (pro == PRO_GN && epi == 0) return LRV(1, PRO_GN, 0, "lr1x1_gn");
        if (pro == PRO_GN_SILU && epi == 0) return LRV(1, PRO_GN_SILU, 0, "lr1x1_gn_silu");
        if (pro == PRO_COLSM && epi == 0)  // never f16x2 (probabilities far below the half range)
            return math == MATH_BF16X1 ? lr_variant1<1, MB, PRO_COLSM, 0, false, true>("lr1x1_colsoftmax") : lr_variant1<1, MB, PRO_COLSM, 0, false>("lr1x1_colsoftmax");
    }
    return ConvVariant();
#undef LRV
}
}  // namespace

// Fused bottleneck attention block (kernels_attn.h): launch helper (the kernel lives in this translation unit)
size_t attn_block_smem() { return AttnBlockGeom::smem; }
// DDIF_ATTN_NW=8: eight wavefronts per sample instead of four; same values either way (kernels_attn.h), tests/test_env_switches.py.  Measured in round 6
// (profiles/r06/attn_block_stamps.txt): 22.3 vs 23.3 us per launch -- the block is bound by the matrix pipe of its ONE CU (qkv: 288 bf16 MFMAs per wave = 9.2 k of the
// 12.8 k cycles of that stage; attention core on the exact fp32 MFMA: 8.2 k of 22.4 k), a second wave per SIMD has no idle pipe to fill.  Four stays the default.
static int attn_nw() {
    static const int v = [] { const char* e = getenv("DDIF_ATTN_NW"); return (e && atoi(e) == 8) ? 8 : 4; }();
    return v;
}
// DDIF_ATTN_SPLIT = 1 / 2: one / two workgroups per sample instead of four (round 6, kernels_attn.h SPLIT: the query tokens of a sample on up to four CUs, k and v
// recomputed by each, no exchange); same values.  Measured at B = 64: 22.5 / 16.6 / 15.8 us per launch, 3.418 / 3.366 / 3.344 ms per step (profiles/r06/attn_split_ab.txt)
int attn_block_split() {
    static const int v = [] { const char* e = getenv("DDIF_ATTN_SPLIT"); const int x = e ? atoi(e) : 4; return (x == 1 || x == 2) ? x : 4; }();
    return attn_nw() == 8 ? 1 : v;
}
int attn_block_prepare() {
    DDIF_HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_block_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)AttnBlockGeom::smem));
    DDIF_HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_block_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)AttnBlockGeom::smem));
    DDIF_HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_block_kernel<4, 0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)AttnBlockGeom::smem));
    DDIF_HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_block_kernel<4, 0, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)AttnBlockGeom::smem));
    DDIF_HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_block_kernel<4, 0, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)AttnBlockGeom::smem));
    return 0;
}
// grid: workgroups to launch (<= a.B * attn_block_split(); each walks samples / parts with that stride).  a.wqkv_f16 != null (set by the plan when the f16x2
// conditions hold): the qkv conv on f16x2 -- instantiated for the default form (four waves, four workgroups per sample) only
void attn_block_launch(const AttnBlockArgs& a, int grid, hipStream_t s) {
    if (attn_nw() == 8) hipLaunchKernelGGL(attn_block_kernel<8>, dim3(grid), dim3(512), AttnBlockGeom::smem, s, a);
    else if (attn_block_split() == 2) hipLaunchKernelGGL((attn_block_kernel<4, 0, 2>), dim3(grid), dim3(256), AttnBlockGeom::smem, s, a);
    else if (attn_block_split() == 4 && a.wqkv_f16) hipLaunchKernelGGL((attn_block_kernel<4, 0, 4, true>), dim3(grid), dim3(256), AttnBlockGeom::smem, s, a);
    else if (attn_block_split() == 4) hipLaunchKernelGGL((attn_block_kernel<4, 0, 4>), dim3(grid), dim3(256), AttnBlockGeom::smem, s, a);
    else hipLaunchKernelGGL(attn_block_kernel<4>, dim3(grid), dim3(256), AttnBlockGeom::smem, s, a);
}

// mb = 2: 8x8 pixel tiles, mb = 4: 8x16.  The per-sample time bias needs no variant of its own here (the epilogue reads
// bias and time-bias rows straight from memory), so EPI_TBS is accepted and ignored.
ConvVariant get_lr_variant(int ks, int mb, int pro, int epi, int math, bool rows) {
    epi &= ~EPI_TBS;
    if (rows) {
        if (ks != 3 || math == MATH_BF16X1) return ConvVariant();
        if (math == MATH_F16X2) return mb == 2 ? lr_rows_for<2, true>(pro, epi) : lr_rows_for<4, true>(pro, epi);
        return mb == 2 ? lr_rows_for<2, false>(pro, epi) : lr_rows_for<4, false>(pro, epi);
    }
    return mb == 2 ? lr_for<2>(ks, pro, epi, math) : lr_for<4>(ks, pro, epi, math);
}

}  // namespace ddif
