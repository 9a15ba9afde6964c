// 3x3 instantiations of the general conv kernel (kernels_conv.h) with the plain epilogue; see conv_variants.h.
#include "conv_variants.h"

namespace ddif {

ConvVariant get_conv_variant_k3(int stride, int ups, int ck, int pro, int cfg, int vec) {
    ConvVariant v;
    const bool plain = stride == 1 && !ups;
    if (ck != 16) return v;  // 3x3 convs always use 16-channel chunks
    if (vec == 2) {
        if (plain && pro == PRO_NONE) { v = variant_for_cfg<3, 1, 0, 16, PRO_NONE, 2>(cfg); v.name = "conv3x3_cat"; }
    } else if (!vec) {
        if (plain && pro == PRO_NONE) { v = variant_for_cfg<3, 1, 0, 16, PRO_NONE, 0>(cfg); v.name = "conv3x3_scalar"; }
    } else if (plain && pro == PRO_GN_SILU) { v = variant_for_cfg<3, 1, 0, 16, PRO_GN_SILU, 1>(cfg); v.name = "conv3x3_gn_silu"; }
    else if (plain && pro == PRO_NONE) { v = variant_for_cfg<3, 1, 0, 16, PRO_NONE, 1>(cfg); v.name = "conv3x3"; }
    else if (stride == 2 && !ups && pro == PRO_NONE) { v = variant_small_tiles<3, 2, 0, 16, PRO_NONE, 1>(cfg); v.name = "conv3x3_s2"; }
    else if (stride == 1 && ups && pro == PRO_NONE) { v = variant_for_cfg<3, 1, 1, 16, PRO_NONE, 1>(cfg); v.name = "conv3x3_up2"; }
    return v;
}

}  // namespace ddif
