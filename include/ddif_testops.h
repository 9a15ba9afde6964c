/* TEST-ONLY entry points of libddif.so (split off include/ddif.h in round 6: that header is the product surface).
 *
 * The stateless per-op forward / backward ops round 2's op-by-op training tape was built from.  The product trains through ONE call
 * (ddif_plan_train_step, include/ddif.h; csrc/ddif_train.cpp: a reverse launch program over NHWC activations); these ops remain exported as an independent
 * cross-check of that program -- tests/ddif_testops.py binds them, tests/train_tape.py strings them into a training step, tests/test_backward_ops.py checks each
 * against torch autograd, tests/test_train_graph.py compares the tape with the native step.  Nothing under dif-pan_amd/ddif/ calls them.
 * Conventions as in include/ddif.h (status codes, ddif_last_error, device pointers, `stream` = hipStream_t or NULL). */
#ifndef DDIF_TESTOPS_H_
#define DDIF_TESTOPS_H_

#include "ddif.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Backward of nn.Conv2d(Cin, Cout, 3, padding=1) as autograd computes it under loss.backward() (diffusion_engine.py:233):
 *   dx = conv_transpose(dy, w), dw[co,ci,ky,kx] = sum dy[b,co,y,x] * x[b,ci,y+ky-1,x+kx-1], db[co] = sum dy[b,co,y,x].
 * All pointers are DEVICE pointers in the reference's layouts: x (B,Cin,H,W), w (Cout,Cin,3,3), dy (B,Cout,H,W),
 * dx (B,Cin,H,W), dw (Cout,Cin,3,3), db (Cout); dx / dw / db may be NULL (skipped).  4 | Cin, 4 | Cout.
 * dgrad = the forward implicit-GEMM kernel on flipped / transposed weights packed on the device (bf16x3 split products like the
 * inference path; DDIF_TRAIN_X3=0: exact fp32 MFMA); wgrad = split-K exact-fp32 MFMA kernel with a fixed-order two-level
 * reduction; bitwise reproducible. */
typedef struct ddif_convbwd* ddif_convbwd_t;
DDIF_API int ddif_convbwd_create(ddif_convbwd_t* out, int B, int Cin, int Cout, int H, int W, int device);
DDIF_API void ddif_convbwd_destroy(ddif_convbwd_t h);
DDIF_API int ddif_convbwd_run(ddif_convbwd_t h, const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, void* stream);

/* Backward of one `Block` (models/sr3_dwt.py:288-300: GroupNorm(1 group, eps 1e-5, affine) -> x*sigmoid(x) -> Dropout -> conv3x3 pad 1 + bias),
 * i.e. what autograd does for it inside `loss.backward()` (diffusion_engine.py:233).  NCHW fp32 device pointers:
 *   x (B,Cin,H,W) the Block's input; gamma, beta (Cin); mask (B,Cin,H,W) = the dropout site's mask holding 0 or 1/(1-p)
 *   (ddif_plan_train_site / _set_dropout), NULL in eval mode; w (Cout,Cin,3,3); dy (B,Cout,H,W) the gradient of the Block's output.
 * Outputs (each nullable): dx (B,Cin,H,W), dgamma, dbeta (Cin), dw (Cout,Cin,3,3), db (Cout), and dy_plane_sums (B,Cout) =
 * sum over pixels of dy -- the gradient of the per-sample time bias FeatureWiseAffine adds to block1's output
 * (models/sr3_dwt.py:241-258, 322).  The activation in front of the conv is recomputed from x (nothing but x is kept
 * from the forward pass).  fp64 fixed-order reductions; bitwise reproducible. */
typedef struct ddif_blockbwd* ddif_blockbwd_t;
DDIF_API int ddif_blockbwd_create(ddif_blockbwd_t* out, int B, int Cin, int Cout, int H, int W, int device);
/* The same op for the other conv-with-prologue shapes of the network: ks = 3 or 1 (w is then (Cout,Cin,1,1)); pro = what sits in
 * front of the conv: NONE (CondInjection.x_conv, ffn.0, ffn.3, attention output convs, stem: gamma = beta = mask = NULL, dx = the
 * conv's dgrad), GN (SelfAttention.norm -> qkv, the attention prenorms), GN_SILU (Block; CondInjection.body's tail), SILU (ffn.2
 * behind ffn.0's SiLU).  resample: PLAIN; DOWN2 = Downsample, conv3x3 stride 2 pad 1 (models/sr3_dwt.py:276-282): x is (B,Cin,H,W),
 * dy (B,Cout,(H-1)/2+1,(W-1)/2+1); UP2 = Upsample, nearest x2 then conv3x3 (:266-273): x (B,Cin,H,W), dy (B,Cout,2H,2W).
 * Correctness first: the dgrad of a 1x1 conv runs on the 3x3 kernel with the weights in the centre tap (9x the matrix work; its
 * wgrad contracts the centre tap only), DOWN2 on the stride-1 kernels with zeros inserted into dy (4x). */
enum { DDIF_BWD_PRO_NONE = 0, DDIF_BWD_PRO_GN = 1, DDIF_BWD_PRO_GN_SILU = 2, DDIF_BWD_PRO_SILU = 3 };
enum { DDIF_BWD_PLAIN = 0, DDIF_BWD_DOWN2 = 1, DDIF_BWD_UP2 = 2 };
DDIF_API int ddif_blockbwd_create_ex(ddif_blockbwd_t* out, int B, int Cin, int Cout, int H, int W, int ks, int pro, int resample, int device);
DDIF_API void ddif_blockbwd_destroy(ddif_blockbwd_t h);
DDIF_API int ddif_blockbwd_run(ddif_blockbwd_t h, const float* x, const float* gamma, const float* beta, const float* mask, const float* w, const float* dy,
                               float* dx, float* dgamma, float* dbeta, float* dw, float* db, float* dy_plane_sums, void* stream);

/* ---- stateless backward ops of the non-convolution pieces (a15) ------------------------------------------------------------
 * NCHW fp32 device pointers on the CURRENT device; every output pointer nullable unless noted; correctness-first kernels
 * (csrc/kernels_bwd_ops.h), fixed-order reductions.  Each mirrors what autograd does for the named reference lines. */
/* depthwise conv3x3, groups = C, pad 1, no bias (FastAttnCondInjection.q[0] / kv[0], models/sr3_dwt.py:507-520): w (C,1,3,3) */
DDIF_API int ddif_dwconv3x3_bwd(const float* x, const float* w, const float* dy, int B, int C, int H, int W, float* dx, float* dw, void* stream);
/* CondInjection's out = xc * (1 + scale) + shift (:395-396): scale_shift (B,2C,H,W) = [scale | shift]; d(scale_shift) same layout */
DDIF_API int ddif_film_bwd(const float* xc, const float* scale_shift, const float* dout, int B, int C, int H, int W, float* dxc, float* dscale_shift,
                           void* stream);
/* SelfAttention core (:345-358): qkv (B,3C,H,W) in the reference's per-head [q|k|v] interleave, dout = gradient of the (B,C,H,W)
 * weighted sum in front of `out`; scale 1/sqrt(C).  H*W <= 64, head dim <= 32. */
DDIF_API int ddif_selfattn_core_bwd(const float* qkv, const float* dout, int B, int C, int H, int W, int heads, float* dqkv, void* stream);
/* FastAttnCondInjection core (:545-566): q_pre (B,qd,H,W) and kv_pre (B,2qd,H,W) = [k | v] BEFORE their softmaxes (over H and
 * over W); dout = gradient of the (B,qd,H,W) attention output in front of attn_out.  Head dim <= 32, W <= 64. */
DDIF_API int ddif_linattn_core_bwd(const float* q_pre, const float* kv_pre, const float* dout, int B, int qd, int H, int W, int heads, float* dq_pre,
                                   float* dkv_pre, void* stream);
/* GroupNorm(1 group, eps 1e-5, affine) alone, for a normalised tensor with several consumers (FastAttnCondInjection.prenorm_x feeds
 * q[0] AND attn_res, :540-573; the caller adds the consumers' gradients into dy).  workspace: B * (2 C + 4) doubles (device). */
DDIF_API int ddif_groupnorm_bwd(const float* x, const float* gamma, const float* dy, int B, int C, int H, int W, float* dx, float* dgamma, float* dbeta,
                                double* workspace, void* stream);
/* nn.Linear (noise_level_mlp, FeatureWiseAffine; :59-64,241-258): x (B,in), w (out,in), dy (B,out) */
DDIF_API int ddif_linear_bwd(const float* x, const float* w, const float* dy, int B, int nin, int nout, float* dx, float* dw, float* db, void* stream);
/* Swish x*sigmoid(x) between the two Linear layers of noise_level_mlp (:61-63) */
DDIF_API int ddif_swish_bwd(const float* x, const float* dy, int64_t n, float* dx, void* stream);
/* F.l1_loss(pred, target) with mean reduction (diffusion/diffusion_ddpm_pan.py:742-749): dpred = sign(pred - target) * upstream / n */
DDIF_API int ddif_l1_loss_bwd(const float* pred, const float* target, int64_t n, float upstream, float* dpred, void* stream);

/* ---- forward ops of the TRAINING graph (a15) --------------------------------------------------------------------------------
 * Stateless NCHW ops of round 2's op-by-op training tape (tests/train_tape.py, kept as an independent cross-check of the native
 * step; the product trains through ddif_plan_train_step, csrc/ddif_train.cpp: a reverse launch program over NHWC activations).
 * Same arithmetic as models/sr3_dwt.py; the convs run on the bf16x3 split products (fp32-class; DDIF_TRAIN_X3=0: the exact-fp32 MFMA). */
typedef struct ddif_convfwd* ddif_convfwd_t;
/* nn.Conv2d(Cin, Cout, ks, stride, padding = ks / 2) [+ nearest x2 upsampling in front: Upsample, models/sr3_dwt.py:266-273];
 * x (B,Cin,H,W), w (Cout,Cin,ks,ks), bias (Cout) or NULL, y (B,Cout,Ho,Wo) */
DDIF_API int ddif_convfwd_create(ddif_convfwd_t* out, int B, int Cin, int Cout, int H, int W, int ks, int stride, int up2, int device);
DDIF_API void ddif_convfwd_destroy(ddif_convfwd_t h);
DDIF_API int ddif_convfwd_run(ddif_convfwd_t h, const float* x, const float* w, const float* bias, float* y, void* stream);
DDIF_API int ddif_dwconv3x3_fwd(const float* x, const float* w, int B, int C, int H, int W, float* y, void* stream);
/* GroupNorm(1 group, eps 1e-5, affine) [-> x*sigmoid(x)] [-> * mask] */
DDIF_API int ddif_groupnorm_fwd(const float* x, const float* gamma, const float* beta, const float* mask, int B, int C, int H, int W, int silu, float* y,
                                void* stream);
DDIF_API int ddif_swish_fwd(const float* x, int64_t n, float* y, void* stream);
DDIF_API int ddif_film_fwd(const float* xc, const float* scale_shift, int B, int C, int H, int W, float* out, void* stream);
/* out = a + alpha[b] * f (alpha NULL: 1): residual adds and DropPath (models/sr3_dwt.py:576) */
DDIF_API int ddif_add_scaled(const float* a, const float* f, const float* alpha, int B, int64_t per_sample, float* out, void* stream);
DDIF_API int ddif_linear_fwd(const float* x, const float* w, const float* bias, int B, int nin, int nout, float* y, void* stream);
DDIF_API int ddif_selfattn_core_fwd(const float* qkv, int B, int C, int H, int W, int heads, float* out, void* stream);
DDIF_API int ddif_linattn_core_fwd(const float* q_pre, const float* kv_pre, int B, int qd, int H, int W, int heads, float* out, void* stream);
/* The same core in the layout and form the native training step runs it (csrc/kernels_linattn.h): q_pre (B,H,W,qd), kv_pre (B,H,W,2qd) = [k | v],
 * out / dout (B,H,W,qd), all NHWC.  workspace (device floats, ddif_linattn_nhwc_workspace of them): the forward leaves the per-sample contexts in
 * its head, the backward of the SAME inputs reads them there.  max(H, W) * qd <= 8192, head dim <= 32. */
DDIF_API int64_t ddif_linattn_nhwc_workspace(int B, int qd, int H, int W, int heads);
DDIF_API int ddif_linattn_nhwc_fwd(const float* q_pre, const float* kv_pre, int B, int qd, int H, int W, int heads, float* out, float* workspace, void* stream);
DDIF_API int ddif_linattn_nhwc_bwd(const float* q_pre, const float* kv_pre, const float* dout, int B, int qd, int H, int W, int heads, float* dq_pre, float* dkv_pre,
                                   float* workspace, void* stream);
/* q_sample (diffusion/diffusion_ddpm_pan.py:668-681): out = a[b] * x0 + s[b] * noise; a, s = B device floats */
DDIF_API int ddif_q_sample(const float* x0, const float* noise, const float* a, const float* s, int B, int64_t per_sample, float* out, void* stream);
/* F.l1_loss(pred, target), mean reduction: out = one device float */
DDIF_API int ddif_l1_loss_fwd(const float* pred, const float* target, int64_t n, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DDIF_TESTOPS_H_ */
