// 1x1 instantiations of the general conv kernel (kernels_conv.h); see conv_variants.h.
#include "conv_variants.h"

namespace ddif {

ConvVariant get_conv_variant_k1(int stride, int ups, int ck, int pro, int cfg, int vec, int epi) {
    ConvVariant v;
    const bool plain = stride == 1 && !ups;
    if (!plain) return v;
    if (epi == EPI_FILM) {
        if (ck == 32 && vec == 1 && pro == PRO_NONE) { v = variant_for_cfg<1, 1, 0, 32, PRO_NONE, 1, EPI_FILM>(cfg); v.name = "conv1x1_film"; }
    } else if (epi == EPI_RES) {
        if (vec != 1) return v;
        if (ck == 32 && pro == PRO_NONE) { v = variant_for_cfg<1, 1, 0, 32, PRO_NONE, 1, EPI_RES>(cfg); v.name = "conv1x1_res"; }
        else if (ck == 32 && pro == PRO_COLSM) { v = variant_for_cfg<1, 1, 0, 32, PRO_COLSM, 1, EPI_RES>(cfg); v.name = "conv1x1_colsoftmax_res"; }
    } else if (epi != 0) {
        return v;
    } else if (!vec) {
        if (ck == 32 && pro == PRO_NONE) { v = variant_for_cfg<1, 1, 0, 32, PRO_NONE, 0>(cfg); v.name = "conv1x1_scalar"; }
        else if (ck == 16 && pro == PRO_NONE) { v = variant_for_cfg<1, 1, 0, 16, PRO_NONE, 0>(cfg); v.name = "conv1x1_ck16_scalar"; }
    } else if (vec == 1) {
        if (ck == 32 && pro == PRO_NONE) { v = variant_for_cfg<1, 1, 0, 32, PRO_NONE, 1>(cfg); v.name = "conv1x1"; }
        else if (ck == 16 && pro == PRO_NONE) { v = variant_for_cfg<1, 1, 0, 16, PRO_NONE, 1>(cfg); v.name = "conv1x1_ck16"; }
        else if (ck == 32 && pro == PRO_GN) { v = variant_for_cfg<1, 1, 0, 32, PRO_GN, 1>(cfg); v.name = "conv1x1_gn"; }
        else if (ck == 32 && pro == PRO_GN_SILU) { v = variant_for_cfg<1, 1, 0, 32, PRO_GN_SILU, 1>(cfg); v.name = "conv1x1_gn_silu"; }
        else if (ck == 32 && pro == PRO_COLSM) { v = variant_for_cfg<1, 1, 0, 32, PRO_COLSM, 1>(cfg); v.name = "conv1x1_colsoftmax"; }
        else if (ck == 32 && pro == PRO_GN_DW) { v = variant_for_cfg<1, 1, 0, 32, PRO_GN_DW, 1>(cfg); v.name = "conv1x1_gn_dw3x3"; }
    }
    return v;
}

}  // namespace ddif
