// 3x3 instantiations of the general conv kernel (kernels_conv.h) with an epilogue variant (residual, SiLU, per-sample time bias, scalar output, sampler update);
// see conv_variants.h.
#include "conv_variants.h"

namespace ddif {

ConvVariant get_conv_variant_k3e(int stride, int ups, int ck, int pro, int cfg, int vec, int epi) {
    ConvVariant v;
    const bool plain = stride == 1 && !ups;
    if (ck != 16 || vec != 1 || !plain) return v;
    if (epi == EPI_SOUT) {
        if (pro == PRO_GN_SILU) { v = variant_for_cfg<3, 1, 0, 16, PRO_GN_SILU, 1, EPI_SOUT>(cfg); v.name = "conv3x3_gn_silu_sout"; }
    } else if (epi == EPI_RES) {
        if (pro == PRO_GN_SILU) { v = variant_for_cfg<3, 1, 0, 16, PRO_GN_SILU, 1, EPI_RES>(cfg); v.name = "conv3x3_gn_silu_res"; }
        else if (pro == PRO_NONE) { v = variant_for_cfg<3, 1, 0, 16, PRO_NONE, 1, EPI_RES>(cfg); v.name = "conv3x3_res"; }
    } else if (epi == EPI_TBS) {
        if (pro == PRO_GN_SILU) { v = variant_for_cfg<3, 1, 0, 16, PRO_GN_SILU, 1, EPI_TBS>(cfg); v.name = "conv3x3_gn_silu_tbs"; }
    } else if (epi == EPI_SAMP) {
        if (pro == PRO_GN_SILU && cfg >= 7 && cfg != 20 && cfg != 21) { v = variant_for_cfg<3, 1, 0, 16, PRO_GN_SILU, 1, EPI_SAMP>(cfg); v.name = "conv3x3_gn_silu_sampler"; }
    } else if (epi == EPI_SILU) {
        if (pro == PRO_NONE) { v = variant_for_cfg<3, 1, 0, 16, PRO_NONE, 1, EPI_SILU>(cfg); v.name = "conv3x3_silu"; }
    }
    return v;
}

}  // namespace ddif
