// EPI_XF instantiations of the conv kernel (kernels_conv.h): a 3x3 conv with 32 output channels whose epilogue ALSO computes the next block's
// CondInjection -- y = (x_conv(out) + b) * (1 + scale) + shift (reference models/sr3_dwt.py:376-396) -- from its accumulator registers and writes it
// as a second output with its own GroupNorm partials.  One translation unit of its own so that it builds beside the main kernel family.
// Only the f16x2 tilings the inference plans of the engine configuration pick for these producers are instantiated:
//   the stem (cat[self_cond, x], VEC = 2), ResnetBlock.block2 (GroupNorm + SiLU prologue, + residual) and the Downsample conv (stride 2).
#include "ddif_plan.h"
#include "kernels_conv.h"

namespace ddif {

namespace {
template <int S, int TH, int TW, int CK, int NW, int PRO, int VEC, int EPI, int MATH>
ConvVariant xf1(const char* name) {
    ConvVariant v;
    v.fn = conv_mfma_kernel<3, S, 0, TH, TW, CK, NW, 1, 1, 1, PRO, VEC, EPI, 0, MATH>;
    v.smem = conv_smem_bytes<3, S, 0, TH, TW, CK, 1, PRO, NW, MATH>();
    v.th = TH;
    v.tw = TW;
    v.nt = 32;
    v.nthr = 64 * NW;
    v.x3 = v.f16 = true;
    v.wr = MATH == 5;
    v.name = name;
    return v;
}
}  // namespace

// cfg: the tiling add_conv() chose for the conv WITHOUT the fold (27 = 16x16 pixels on eight waves, 28 = 8x16 on four, 37 = 27 with resident weights);
// epi: its epilogue bits without the fold (0 or EPI_RES); nbx: 32-cout blocks of the x_conv (1 or 2).  Null fn: no such instantiation -> no fold.
ConvVariant get_xf_variant(int stride, int pro, int cfg, int vec, int epi, int nbx) {
    if (stride == 1 && pro == PRO_GN_SILU && vec == 1 && epi == EPI_RES && nbx == 1) {  // ResnetBlock.block2 -> next block's x_conv 32 -> 32
        if (cfg == 37) return xf1<1, 16, 16, 32, 8, PRO_GN_SILU, 1, EPI_RES | EPI_XF1, 5>("conv3x3_gn_silu_res_xfilm");
        if (cfg == 27) return xf1<1, 16, 16, 16, 8, PRO_GN_SILU, 1, EPI_RES | EPI_XF1, 3>("conv3x3_gn_silu_res_xfilm");
        if (cfg == 28) return xf1<1, 8, 16, 16, 4, PRO_GN_SILU, 1, EPI_RES | EPI_XF1, 3>("conv3x3_gn_silu_res_xfilm");
    }
    if (stride == 1 && pro == PRO_NONE && vec == 2 && epi == 0 && nbx == 1) {  // stem -> first block's x_conv 32 -> 32
        if (cfg == 27) return xf1<1, 16, 16, 16, 8, PRO_NONE, 2, EPI_XF1, 3>("conv3x3_cat_xfilm");
        if (cfg == 28) return xf1<1, 8, 16, 16, 4, PRO_NONE, 2, EPI_XF1, 3>("conv3x3_cat_xfilm");
    }
    if (stride == 2 && pro == PRO_NONE && vec == 1 && epi == 0 && nbx == 2) {  // Downsample 32 -> 32 -> next block's x_conv 32 -> 64
        if (cfg == 28) return xf1<2, 8, 16, 16, 4, PRO_NONE, 1, EPI_XF2, 3>("conv3x3_s2_xfilm");
    }
    return ConvVariant();
}

}  // namespace ddif
