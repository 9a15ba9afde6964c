// Launch arguments of the fused bottleneck attention block (kernels_attn.h); shared with the plan builder.
#pragma once

namespace ddif {

struct AttnBlockArgs {
    const float* x;       // [B, 64, 128] NHWC input (also the residual)
    const double* st;     // GroupNorm partials of x, [B][np][2]
    int np;
    const float* gamma;   // norm.weight / norm.bias
    const float* beta;
    const float* wqkv;    // packed bf16x3 1x1 weights, 12 cout blocks x 8 slabs x 3 planes x 1 KiB
    const float* wqkv_f16; // the same as f16x2 planes (12 x 8 slabs x 2 planes x 1 KiB; ddif_net.cpp pack_conv_f16) or null: the qkv conv on three products per fp32
                          // product instead of six -- its input is GroupNorm output, bounded on the host like every f16x2 conv (round 6)
    const float* wout;    // packed bf16x3 1x1 weights, 4 cout blocks x 8 slabs x 3 planes x 1 KiB
    const float* bout;    // [128]
    float scale;          // 1 / sqrt(C)  (NOT 1 / sqrt(d): sr3_dwt.py:352)
    float* out;           // [B, 64, 128]
    double* st_out;       // [B][1][2] or null
    int B;
    int b0;               // first sample
    int xcd;              // samples of one XCD contiguous (workgroup w runs on XCD w % 8; ddif_dev.h wg_work_range)
    long long* dbg;       // microbenchmark instrumentation (ABL & 1, tools/mbench_attn.cpp) only
};

}  // namespace ddif
