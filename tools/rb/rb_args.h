// Launch arguments of the resident ResnetBlock kernel of the 8 x 8 level (kernels_rb.h); shared with the plan builder.
#pragma once

namespace ddif {

struct RbArgs {
    const float* x;        // [B, 64, 128] NHWC input (also the residual)
    const double* st;      // GroupNorm partials of x, [B][np][2]
    int np;
    const float *g1, *b1;  // block1.block.0 weight / bias (GroupNorm 1)
    const float *g2, *b2;  // block2.block.0
    const float *w1, *w2;  // f16x2 packs of block1.block.3 / block2.block.3 (ddif_net.cpp pack_conv_f16, 16-channel chunks): [4 cout blocks][8 slabs][9 taps][hi | lo][1 KiB]
    const float *bias1, *bias2;  // [128]
    const float* tbias;    // time-bias rows of conv1 (FeatureWiseAffine): row of sample b = tbias + step * tb_rowstride + b * tbias_stride
    int tbias_stride;
    const int* step_ptr;   // device step counter of the running sampler (null: step 0)
    int tb_rowstride;
    float* out;            // [B, 64, 128]
    double* st_out;        // [B][1][2] or null
    int B;
};

}  // namespace ddif
