// Tiling table of the general conv kernel (kernels_conv.h): which instantiation a configuration number means.  Internal to the two translation units that
// instantiate the family -- ddif_conv_k3.cpp (3x3 convs) and ddif_conv_k1.cpp (1x1 convs) -- so that the kernels build in parallel and the plan builder
// (ddif_plan.cpp) no longer recompiles them.
#pragma once
#include "ddif_plan.h"
#include "kernels_conv.h"

namespace ddif {
namespace {

// cfg: 0 = 8x16 pixels x 32 couts, 1 = 8x16 x 64, 3 = 8x16 x 128 (1x1 convs only: a wide cout tile stages -- and for
// PRO_GN_DW recomputes -- the input once instead of once per 32 couts), 2 = 8x8 x 64, 4 = 8x8 x 128 (1x1 only);
// 5 = 16x16 x 32 and 6 = 16x16 x 64 with EIGHT wavefronts (plain 3x3 convs at the high-resolution levels: twice the
// MFMAs per staged input / weight element, see kernels_conv.h)
template <int KS, int S, int U, int CK, int PRO, int VEC, int EPI = 0>
ConvVariant variant_for_cfg(int cfg) {
    ConvVariant v;
    switch (cfg) {
        case 0: v.fn = conv_mfma_kernel<KS, S, U, 8, 16, CK, 4, 1, 1, 1, PRO, VEC, EPI>; v.smem = conv_smem_bytes<KS, S, U, 8, 16, CK, 1, PRO>(); v.th = 8; v.tw = 16; v.nt = 32; break;
        case 2: v.fn = conv_mfma_kernel<KS, S, U, 8, 8, CK, 2, 2, 1, 1, PRO, VEC, EPI>; v.smem = conv_smem_bytes<KS, S, U, 8, 8, CK, 2, PRO>(); v.th = 8; v.tw = 8; v.nt = 64; break;
        default: break;
    }
    if constexpr (KS == 1 && VEC == 1) {
        switch (cfg) {
            case 1: v.fn = conv_mfma_kernel<KS, S, U, 8, 16, CK, 4, 1, 1, 2, PRO, VEC, EPI>; v.smem = conv_smem_bytes<KS, S, U, 8, 16, CK, 2, PRO>(); v.th = 8; v.tw = 16; v.nt = 64; break;
            case 3: v.fn = conv_mfma_kernel<KS, S, U, 8, 16, CK, 4, 1, 1, 4, PRO, VEC, EPI>; v.smem = conv_smem_bytes<KS, S, U, 8, 16, CK, 4, PRO>(); v.th = 8; v.tw = 16; v.nt = 128; break;
            case 4: v.fn = conv_mfma_kernel<KS, S, U, 8, 8, CK, 2, 2, 1, 2, PRO, VEC, EPI>; v.smem = conv_smem_bytes<KS, S, U, 8, 8, CK, 4, PRO>(); v.th = 8; v.tw = 8; v.nt = 128; break;
            default: break;
        }
        if constexpr (CK == 32) {  // bf16x3 instantiations of the same five tilings (cfg + 12)
            switch (cfg) {
                case 12: v.fn = conv_mfma_kernel<KS, S, U, 8, 16, CK, 4, 1, 1, 1, PRO, VEC, EPI, 0, 1>; v.smem = conv_smem_bytes<KS, S, U, 8, 16, CK, 1, PRO, 4, 1>(); v.th = 8; v.tw = 16; v.nt = 32; v.x3 = true; break;
                case 13: v.fn = conv_mfma_kernel<KS, S, U, 8, 16, CK, 4, 1, 1, 2, PRO, VEC, EPI, 0, 1>; v.smem = conv_smem_bytes<KS, S, U, 8, 16, CK, 2, PRO, 4, 1>(); v.th = 8; v.tw = 16; v.nt = 64; v.x3 = true; break;
                case 15: v.fn = conv_mfma_kernel<KS, S, U, 8, 16, CK, 4, 1, 1, 4, PRO, VEC, EPI, 0, 1>; v.smem = conv_smem_bytes<KS, S, U, 8, 16, CK, 4, PRO, 4, 1>(); v.th = 8; v.tw = 16; v.nt = 128; v.x3 = true; break;
                case 14: v.fn = conv_mfma_kernel<KS, S, U, 8, 8, CK, 2, 2, 1, 1, PRO, VEC, EPI, 0, 1>; v.smem = conv_smem_bytes<KS, S, U, 8, 8, CK, 2, PRO, 4, 1>(); v.th = 8; v.tw = 8; v.nt = 64; v.x3 = true; break;
                case 16: v.fn = conv_mfma_kernel<KS, S, U, 8, 8, CK, 2, 2, 1, 2, PRO, VEC, EPI, 0, 1>; v.smem = conv_smem_bytes<KS, S, U, 8, 8, CK, 4, PRO, 4, 1>(); v.th = 8; v.tw = 8; v.nt = 128; v.x3 = true; break;
                // bf16x1 (MATH = 4, the throughput variant): the same five tilings (cfg + 52); the 32-cout tile too (no split to pay for)
                case 52: v.fn = conv_mfma_kernel<KS, S, U, 8, 16, CK, 4, 1, 1, 1, PRO, VEC, EPI, 0, 4>; v.smem = conv_smem_bytes<KS, S, U, 8, 16, CK, 1, PRO, 4, 4>(); v.th = 8; v.tw = 16; v.nt = 32; v.x3 = v.b1 = true; break;
                case 53: v.fn = conv_mfma_kernel<KS, S, U, 8, 16, CK, 4, 1, 1, 2, PRO, VEC, EPI, 0, 4>; v.smem = conv_smem_bytes<KS, S, U, 8, 16, CK, 2, PRO, 4, 4>(); v.th = 8; v.tw = 16; v.nt = 64; v.x3 = v.b1 = true; break;
                case 55: v.fn = conv_mfma_kernel<KS, S, U, 8, 16, CK, 4, 1, 1, 4, PRO, VEC, EPI, 0, 4>; v.smem = conv_smem_bytes<KS, S, U, 8, 16, CK, 4, PRO, 4, 4>(); v.th = 8; v.tw = 16; v.nt = 128; v.x3 = v.b1 = true; break;
                case 54: v.fn = conv_mfma_kernel<KS, S, U, 8, 8, CK, 2, 2, 1, 1, PRO, VEC, EPI, 0, 4>; v.smem = conv_smem_bytes<KS, S, U, 8, 8, CK, 2, PRO, 4, 4>(); v.th = 8; v.tw = 8; v.nt = 64; v.x3 = v.b1 = true; break;
                case 56: v.fn = conv_mfma_kernel<KS, S, U, 8, 8, CK, 2, 2, 1, 2, PRO, VEC, EPI, 0, 4>; v.smem = conv_smem_bytes<KS, S, U, 8, 8, CK, 4, PRO, 4, 4>(); v.th = 8; v.tw = 8; v.nt = 128; v.x3 = v.b1 = true; break;
                default: break;
            }
        }
    }
    if constexpr (KS == 3 && S == 1 && VEC == 1) {
        switch (cfg) {
            // bf16x3 (MATH = 1): 7 = 16x16 x 32 on eight waves, 8 = 8x16 x 32 on four, 9 = 8x8 x 64 on four
            case 7: v.fn = conv_mfma_kernel<KS, S, U, 16, 16, CK, 8, 1, 1, 1, PRO, VEC, EPI, 0, 1>; v.smem = conv_smem_bytes<KS, S, U, 16, 16, CK, 1, PRO, 8, 1>(); v.th = 16; v.tw = 16; v.nt = 32; v.nthr = 512; v.x3 = true; break;
            case 8: v.fn = conv_mfma_kernel<KS, S, U, 8, 16, CK, 4, 1, 1, 1, PRO, VEC, EPI, 0, 1>; v.smem = conv_smem_bytes<KS, S, U, 8, 16, CK, 1, PRO, 4, 1>(); v.th = 8; v.tw = 16; v.nt = 32; v.x3 = true; break;
            case 9: v.fn = conv_mfma_kernel<KS, S, U, 8, 8, CK, 2, 2, 1, 1, PRO, VEC, EPI, 0, 1>; v.smem = conv_smem_bytes<KS, S, U, 8, 8, CK, 2, PRO, 4, 1>(); v.th = 8; v.tw = 8; v.nt = 64; v.x3 = true; break;
            // bf16x1 (MATH = 4, the throughput variant): the same three tilings (+ 40)
            case 47: v.fn = conv_mfma_kernel<KS, S, U, 16, 16, CK, 8, 1, 1, 1, PRO, VEC, EPI, 0, 4>; v.smem = conv_smem_bytes<KS, S, U, 16, 16, CK, 1, PRO, 8, 4>(); v.th = 16; v.tw = 16; v.nt = 32; v.nthr = 512; v.x3 = v.b1 = true; break;
            case 48: v.fn = conv_mfma_kernel<KS, S, U, 8, 16, CK, 4, 1, 1, 1, PRO, VEC, EPI, 0, 4>; v.smem = conv_smem_bytes<KS, S, U, 8, 16, CK, 1, PRO, 4, 4>(); v.th = 8; v.tw = 16; v.nt = 32; v.x3 = v.b1 = true; break;
            case 49: v.fn = conv_mfma_kernel<KS, S, U, 8, 8, CK, 2, 2, 1, 1, PRO, VEC, EPI, 0, 4>; v.smem = conv_smem_bytes<KS, S, U, 8, 8, CK, 2, PRO, 4, 4>(); v.th = 8; v.tw = 8; v.nt = 64; v.x3 = v.b1 = true; break;
            default: break;
        }
    }
    if constexpr (KS == 3 && S == 1 && VEC == 1 && (PRO == PRO_NONE || PRO == PRO_GN_SILU)) {
        switch (cfg) {
            // f16x2 (MATH = 3): the bf16x3 tilings 7 / 8 / 9 (+ 20).  (A 16x16 x 64-cout tiling on eight waves measured 0.5 % faster on the
            // step at B = 64 -- 4.75 vs 4.775 ms -- but a cout tile that depends on the item count regroups the GroupNorm partials, and
            // tiles of a batch are then no longer bit-equal to single-tile runs: not kept.)
            case 27: v.fn = conv_mfma_kernel<KS, S, U, 16, 16, CK, 8, 1, 1, 1, PRO, VEC, EPI, 0, 3>; v.smem = conv_smem_bytes<KS, S, U, 16, 16, CK, 1, PRO, 8, 3>(); v.th = 16; v.tw = 16; v.nt = 32; v.nthr = 512; v.x3 = v.f16 = true; break;
            case 28: v.fn = conv_mfma_kernel<KS, S, U, 8, 16, CK, 4, 1, 1, 1, PRO, VEC, EPI, 0, 3>; v.smem = conv_smem_bytes<KS, S, U, 8, 16, CK, 1, PRO, 4, 3>(); v.th = 8; v.tw = 16; v.nt = 32; v.x3 = v.f16 = true; break;
            case 29: v.fn = conv_mfma_kernel<KS, S, U, 8, 8, CK, 2, 2, 1, 1, PRO, VEC, EPI, 0, 3>; v.smem = conv_smem_bytes<KS, S, U, 8, 8, CK, 2, PRO, 4, 3>(); v.th = 8; v.tw = 8; v.nt = 64; v.x3 = v.f16 = true; break;
            default: break;
        }
        if constexpr (U == 0) {
            // 37 = tiling 27 with RESIDENT weights (MATH = 5, round 5): 32 input channels as ONE stage per work item, the cout tile's weights copied into
            // LDS once per workgroup.  Same pack, same accumulation order, same partials as 27 -- bit-identical results, 3-6 % faster in isolation
            // (profiles/r05/a_mbench_resident.txt).  Chosen by add_conv for 32 -> <= 32 channel convs.
            if (cfg == 37) { v.fn = conv_mfma_kernel<KS, S, U, 16, 16, 32, 8, 1, 1, 1, PRO, VEC, EPI, 0, 5>; v.smem = conv_smem_bytes<KS, S, U, 16, 16, 32, 1, PRO, 8, 5>(); v.th = 16; v.tw = 16; v.nt = 32; v.nthr = 512; v.x3 = v.f16 = v.wr = true; }
        }
    }
    if constexpr (KS == 3 && S == 1 && U == 0 && VEC == 2 && PRO == PRO_NONE && EPI == 0) {
        // the stem (cat[self_cond, x] inside ONE 16-channel chunk: float4 staging with a per-thread source select) on the f16x2 tilings 27 / 28 (round 5; until then the
        // exact-fp32 tilings: 38.8 us per launch at B = 64 for 2.4 GFLOP)
        switch (cfg) {
            case 27: v.fn = conv_mfma_kernel<KS, S, U, 16, 16, CK, 8, 1, 1, 1, PRO, VEC, EPI, 0, 3>; v.smem = conv_smem_bytes<KS, S, U, 16, 16, CK, 1, PRO, 8, 3>(); v.th = 16; v.tw = 16; v.nt = 32; v.nthr = 512; v.x3 = v.f16 = true; break;
            case 28: v.fn = conv_mfma_kernel<KS, S, U, 8, 16, CK, 4, 1, 1, 1, PRO, VEC, EPI, 0, 3>; v.smem = conv_smem_bytes<KS, S, U, 8, 16, CK, 1, PRO, 4, 3>(); v.th = 8; v.tw = 16; v.nt = 32; v.x3 = v.f16 = true; break;
            default: break;
        }
    }
    if constexpr (KS == 3 && S == 2 && U == 0 && VEC == 1 && PRO == PRO_NONE && EPI == 0) {
        // the Downsample convs on the f16x2 path (round 5; until then the exact-fp32 8 x 8 tiling 2: 43.6 / 26.5 us per launch at the 32 x 32 / 16 x 16 outputs for
        // 1.2 GFLOP each): the 8 x 16 x 32 and 8 x 8 x 64 tilings 28 / 29 with a stride-2 halo tile (17 x 33 / 17 x 17 staged pixels)
        switch (cfg) {
            case 28: v.fn = conv_mfma_kernel<KS, S, U, 8, 16, CK, 4, 1, 1, 1, PRO, VEC, EPI, 0, 3>; v.smem = conv_smem_bytes<KS, S, U, 8, 16, CK, 1, PRO, 4, 3>(); v.th = 8; v.tw = 16; v.nt = 32; v.x3 = v.f16 = true; break;
            case 29: v.fn = conv_mfma_kernel<KS, S, U, 8, 8, CK, 2, 2, 1, 1, PRO, VEC, EPI, 0, 3>; v.smem = conv_smem_bytes<KS, S, U, 8, 8, CK, 2, PRO, 4, 3>(); v.th = 8; v.tw = 8; v.nt = 64; v.x3 = v.f16 = true; break;
            default: break;
        }
    }
    if constexpr (KS == 3 && S == 1 && U == 0 && VEC == 1) {
        switch (cfg) {
            case 5: v.fn = conv_mfma_kernel<KS, S, U, 16, 16, CK, 8, 1, 1, 1, PRO, VEC, EPI>; v.smem = conv_smem_bytes<KS, S, U, 16, 16, CK, 1, PRO, 8>(); v.th = 16; v.tw = 16; v.nt = 32; v.nthr = 512; break;
            case 6: v.fn = conv_mfma_kernel<KS, S, U, 16, 16, CK, 8, 1, 1, 2, PRO, VEC, EPI>; v.smem = conv_smem_bytes<KS, S, U, 16, 16, CK, 2, PRO, 8>(); v.th = 16; v.tw = 16; v.nt = 64; v.nthr = 512; break;
            default: break;
        }
    }
    return v;
}
template <int KS, int S, int U, int CK, int PRO, int VEC>
ConvVariant variant_small_tiles(int cfg) {  // stride-2: the 8x16 halo would not fit comfortably in LDS
    return (cfg >= 2) ? variant_for_cfg<KS, S, U, CK, PRO, VEC>(cfg) : ConvVariant();
}
}  // namespace
}  // namespace ddif
