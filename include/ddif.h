/* libddif -- C ABI of the MI355X-native DDIF denoising hot path.
 *
 * The reference (294coder/Dif-PAN) has no FFI: its boundary for this path is a set of Python call signatures.
 * Each entry point below names the reference interface it stands in for (paths relative to the reference root).
 * The Python package dif-pan_amd/ddif binds these with ctypes and re-exposes the reference's
 * own classes on top: UNetSR3 (models/sr3_dwt.py:30-219), GaussianDiffusion (diffusion/diffusion_ddpm_pan.py:143-778),
 * NoiseScheduleVP / model_wrapper / DPM_Solver (solver/dpm_solver.py).
 *
 * Conventions
 *   - every function returns 0 on success or a negative ddif_status; it never throws and never aborts;
 *     ddif_last_error() returns the message of the calling thread's last failure.
 *   - tensors crossing the boundary are fp32, contiguous, NCHW (the reference's layout), device pointers unless a
 *     parameter says "host".  The library converts to its internal NHWC layout.
 *   - all device work is enqueued on the caller's stream (a hipStream_t passed as void*; NULL = default stream) and
 *     is asynchronous; the library never calls hipDeviceSynchronize.  Pointers are borrowed until that work ends.
 *   - handles are not thread-safe; one host thread per handle, one process per GPU.
 */
#ifndef DDIF_H_
#define DDIF_H_

#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__)
#define DDIF_API __attribute__((visibility("default")))
#else
#define DDIF_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef enum ddif_status {
    DDIF_OK = 0,
    DDIF_ERR_INVALID = -1,     /* bad argument / unsupported configuration */
    DDIF_ERR_HIP = -2,         /* a HIP runtime call failed */
    DDIF_ERR_MISSING = -3,     /* a required weight was never loaded */
    DDIF_ERR_STATE = -4,       /* call order violated (e.g. sample before set_cond) */
    DDIF_ERR_RANGE = -5        /* an activation left the range of the plan's arithmetic (ddif_plan_range_status): the result of that call is not valid */
} ddif_status;

typedef struct ddif_net* ddif_net_t;   /* weights (repacked for the kernels) of one UNetSR3 */
typedef struct ddif_plan* ddif_plan_t; /* per-(B,H,W) workspaces, cond caches, launch program */

/* Constructor arguments of UNetSR3 (models/sr3_dwt.py:31-51) that shape the network. */
typedef struct ddif_net_cfg {
    int32_t in_channel, out_channel, inner_channel, lms_channel, pan_channel, norm_groups;
    int32_t n_channel_mults;
    int32_t channel_mults[8];
    int32_t n_attn_res;
    int32_t attn_res[8];
    int32_t res_blocks, image_size, self_condition;
} ddif_net_cfg;

/* ---- network ------------------------------------------------------------------------------------------------ */

/* UNetSR3.__init__ (models/sr3_dwt.py:31-167).  Unsupported configurations (norm_groups != 1, fourier features,
 * pred_var, inner_channel != 32, head dim != 16) fail with DDIF_ERR_INVALID -- there is no fallback path. */
DDIF_API int ddif_net_create(ddif_net_t* out, const ddif_net_cfg* cfg, int device);
DDIF_API void ddif_net_destroy(ddif_net_t net);

/* nn.Module.load_state_dict, one tensor at a time (checkpoint layout: SURVEY.md appendix C; utils/misc.py:89-122).
 * `key` is the reference state-dict key, `data` a HOST pointer to fp32 in the reference layout (OIHW / (out,in)).
 * The pseudo key "noise_level_mlp.0.freqs" (inner_channel/2 floats) overrides the positional-encoding
 * frequencies exp(-ln(1e4) * j / count) (models/sr3_dwt.py:229-236) so they are bit-identical to torch's. */
DDIF_API int ddif_net_load(ddif_net_t net, const char* key, const float* data, const int64_t* shape, int ndim);
/* Repack everything loaded so far for the kernels and upload.  Must follow the last ddif_net_load. */
DDIF_API int ddif_net_commit(ddif_net_t net, void* stream);
/* Training: rewrite the packed weights IN PLACE from DEVICE parameter tensors (reference layouts, fp32; n (key, pointer) pairs covering
 * every learnable tensor of the state dict) -- one launch on `stream`, no host copy.  Plans of this net stay valid.  The eval-only
 * merged decoder-FFN weights are NOT refreshed: after a refresh only train-mode plans (ddif_plan_create_train) may run until the next
 * ddif_net_load + ddif_net_commit (inference plans are refused).  Replaces nothing in the reference: there `optimizer.step()` writes
 * the tensors the next forward reads (diffusion_engine.py:238); here the kernels read a packed copy. */
DDIF_API int ddif_net_refresh(ddif_net_t net, int n, const char* const* keys, const float* const* params_dev, void* stream);
DDIF_API int64_t ddif_net_num_params(ddif_net_t net);

/* ---- plan ----------------------------------------------------------------------------------------------------- */

/* Workspaces and launch program for batches of B tiles of H x W (H, W multiples of 8). */
DDIF_API int ddif_plan_create(ddif_plan_t* out, ddif_net_t net, int B, int H, int W);
DDIF_API void ddif_plan_destroy(ddif_plan_t plan);

/* Everything in UNetSR3.forward that depends only on `cond` (models/sr3_dwt.py:389-391,540-546,563,661-663):
 * bilinear resizes, CondInjection.body -> FiLM scale/shift, FastAttnCondInjection kv -> softmax -> context.
 * cond: (B, 2C+4P, H, W) = cat[lms, pan, up(wavelets)] (diffusion_engine.py:221-228). */
DDIF_API int ddif_plan_set_cond(ddif_plan_t plan, const float* cond, void* stream);

/* UNetSR3.forward(x, time, cond, self_cond) (models/sr3_dwt.py:169-219) with cond as given to set_cond.
 * time: B floats on the HOST (long timesteps are passed as their float value); self_cond may be NULL (-> x). */
DDIF_API int ddif_plan_forward(ddif_plan_t plan, const float* x, const float* time_host, const float* self_cond, float* out,
                      void* stream);

/* Per-step coefficient tables of one sampling run, HOST arrays of n_steps floats in EXECUTION order
 * (first executed step first).  The caller derives them from the schedule buffers exactly as the reference does. */
typedef struct ddif_ddpm_tables {
    int32_t n_steps;
    const float* t_model;   /* value fed to the network as `time` */
    const float* coef_x0;   /* posterior_mean_coef1[t]                       (diffusion_ddpm_pan.py:316-320) */
    const float* coef_xt;   /* posterior_mean_coef2[t] */
    const float* coef_z;    /* [t != 0] * exp(0.5 * posterior_log_variance_clipped[t])          (:441-442) */
} ddif_ddpm_tables;

/* GaussianDiffusion.p_sample_loop (diffusion/diffusion_ddpm_pan.py:445-507) for pred_mode="x_start":
 *   img = x_T; for each step: x0 = net(img, t, cond, self_cond=img); x0 = clamp(x0 + lms, lo, hi) - lms (if do_clamp);
 *   img = coef_x0*x0 + coef_xt*img + coef_z*z.
 * x_T: (B,C,H,W) or NULL; noise: (n_steps,B,C,H,W) standard normals in execution order or NULL.  NULL draws from the
 * on-device counter-based generator keyed by (seed, draw index, tile0 + tile index, element), so a batch split
 * over several GPUs reproduces the single-GPU result.  out: (B,C,H,W) = final img (the residual to lms). */
DDIF_API int ddif_plan_sample_ddpm(ddif_plan_t plan, const ddif_ddpm_tables* tabs, const float* x_T, const float* noise,
                          uint64_t seed, uint64_t tile0, float clamp_lo, float clamp_hi, int do_clamp, float* out,
                          void* stream);

typedef struct ddif_ddim_tables {
    int32_t n_steps;
    const float* t_model;      /* the RESPACED index j (diffusion_ddpm_pan.py:661) */
    const float* sqrt_recip;   /* sqrt_recip_alphas_cumprod[j]      (:284-287) */
    const float* sqrt_recipm1; /* sqrt_recipm1_alphas_cumprod[j] */
    const float* sqrt_ap;      /* sqrt(alphas_cumprod_prev[j])      (:615-618) */
    const float* dir_coef;     /* sqrt(1 - alphas_cumprod_prev[j] - sigma^2) */
    const float* sigma;        /* [j != 0] * eta-sigma              (:609-613,619-620) */
} ddif_ddim_tables;

/* GaussianDiffusion.ddim_sample_loop after respacing (diffusion/diffusion_ddpm_pan.py:624-666,594-621);
 * self_cond is never passed (-> x); no clamp unless do_clamp. */
DDIF_API int ddif_plan_sample_ddim(ddif_plan_t plan, const ddif_ddim_tables* tabs, const float* x_T, const float* noise,
                          uint64_t seed, uint64_t tile0, float clamp_lo, float clamp_hi, int do_clamp, float* out,
                          void* stream);

/* One model evaluation of DPM-Solver++ (solver/dpm_solver.py:279-300,441-450): net(x, t_model, cond) -> eps ->
 * x0 -> image-space clamp corrector; x, x0_out: (B,C,H,W) in the library's INTERNAL layout handles below. */
typedef struct ddif_dpm_tables {
    int32_t n_evals;           /* model evaluations = steps */
    int32_t order;             /* 1..3 */
    const float* t_model;      /* (t - 1/N) * 1000, float                              (dpm_solver.py:285-286) */
    const float* alpha;        /* marginal alpha at each evaluation time */
    const float* sigma;        /* marginal std   at each evaluation time */
    /* update k (k = 0 .. n_evals-1) advances x from evaluation time k to time k+1 with order ord[k]: */
    const int32_t* ord;
    const float* cx;           /* sigma_t / sigma_prev                                  (:836,897) */
    const float* a_phi1;       /* alpha_t * expm1(-h) */
    const float* inv_r0;       /* 1 / r0      (orders 2,3) */
    const float* inv_r1;       /* 1 / r1      (order 3) */
    const float* r0_frac;      /* r0 / (r0 + r1) */
    const float* inv_r01;      /* 1 / (r0 + r1) */
    const float* a_phi2;       /* alpha_t * phi_2 */
    const float* a_phi3;       /* alpha_t * phi_3 */
} ddif_dpm_tables;

/* DPM_Solver.sample(method="multistep", algorithm_type="dpmsolver++") (solver/dpm_solver.py:1179-1221). */
DDIF_API int ddif_plan_sample_dpmpp(ddif_plan_t plan, const ddif_dpm_tables* tabs, const float* x_T, float clamp_lo,
                           float clamp_hi, int do_clamp, float* out, void* stream);

/* GaussianDiffusion.p_losses forward half (diffusion/diffusion_ddpm_pan.py:692-732), eval mode, pred_mode x_start:
 * x_t = a[b]*x0 + s[b]*noise; pred = net(x_t, t, cond, self_cond); returns pred (recon_x0). a, s, time: B floats each, HOST or DEVICE
 * memory (copied on `stream`: a training loop that draws t on the device hands them over without a host round trip). */
DDIF_API int ddif_plan_q_sample_forward(ddif_plan_t plan, const float* x0, const float* noise, const float* sqrt_ac_host,
                               const float* sqrt_1mac_host, const float* time_host, const float* self_cond,
                               float* pred, void* stream);

/* ---- either side of the denoising loop ---------------------------------------------------------------------------- */

/* Cond assembly of the engine, fused with the level-1 Haar analysis the reference does with PyWavelets on the CPU at
 * dataset construction (dataset/pan_dataset.py:73-81,127-142; dataset/hisr.py:48-59; diffusion_engine.py:221-228,441-444):
 *   cond = cat[lms, pan, bilinear_up2([LL(lms), details(pan)])] / division        -> (B, 2C+4P, H, W)
 * lms_raw (B,C,H,W), pan_raw (B,P,H,W): device, raw sensor units; H, W even.
 * wavelet_order 0 = [LL, H, D, V] (PanCollection sets), 1 = [LL, H, V, D] (CAVE / Harvard). */
DDIF_API int ddif_cond_assemble(const float* lms_raw, const float* pan_raw, float division, int B, int C, int P, int H, int W,
                                int wavelet_order, float* cond_out, void* stream);

/* Validation metrics per image (utils/_metric_legacy.py:299-379 analysis_accu(choices=5) as called by
 * utils/metric.py:24-98 AnalysisPanAcc): out[b] = {SAM, ERGAS, PSNR (reference sign), CC}; gt, pred (B,C,H,W) device;
 * out: device, B*4 floats.  The reference's quirks are kept: last row / column dropped, pi = 3.14159256, SAM rounded to
 * 6 digits, PSNR = -20 log10(1/rmse). */
DDIF_API int ddif_metrics(const float* gt, const float* pred, int B, int C, int H, int W, float ergas_ratio, float* out, void* stream);
/* SSIM per image (out: B floats) as `skimage.metrics.structural_similarity(gt, pred, channel_axis=0)` computes it with library defaults
 * (reference utils/metric.py:153-166): 7x7 uniform window, K1 0.01, K2 0.03, sample covariance, mean over the image cropped by 3 pixels
 * per side and over the channels.  data_range: the reference passes none, for which skimage takes the float dtype range (-1, 1) -> 2.0.
 * PARITY UNPINNED: skimage is absent from the build image; the oracle restates the published algorithm. */
DDIF_API int ddif_ssim(const float* gt, const float* pred, int B, int C, int H, int W, float data_range, float* out, void* stream);

/* Fused optimizer step of the training loop (diffusion_engine.py:237-241): global-norm gradient clipping
 * (utils/misc.py:25-36 -> clip_grad_norm_), torch.optim.AdamW.step and EmaUpdater.update (utils/optim_utils.py:43-58) as
 * three launches over a chunk table, whatever the number of tensors.  The handle owns the moment buffers (zero-initialised);
 * params / grads / ema are borrowed device pointers that must stay valid for the handle's lifetime (ema or its entries may
 * be NULL). */
typedef struct ddif_optim* ddif_optim_t;
DDIF_API int ddif_optim_create(ddif_optim_t* out, int n_tensors, const int64_t* sizes, float* const* params, const float* const* grads,
                               float* const* ema, int device);
/* Same with CALLER-OWNED moment buffers (exp_avg / exp_avg_sq of torch.optim.AdamW, one device array per tensor, borrowed for the handle's
 * lifetime and not zeroed): the optimizer state then lives where a checkpoint can save and restore it (reference diffusion_engine.py:333-340
 * saves weights only; full-state resume is SURVEY 8f-4). */
DDIF_API int ddif_optim_create_ex(ddif_optim_t* out, int n_tensors, const int64_t* sizes, float* const* params, const float* const* grads,
                                  float* const* ema, float* const* exp_avg, float* const* exp_avg_sq, int device);
DDIF_API void ddif_optim_destroy(ddif_optim_t h);
/* step: 1-based AdamW step count; max_grad_norm <= 0 disables clipping; ema_mode 0 = leave ema alone, 1 = ema <- p
 * (iteration <= start_iter), 2 = ema <- ema*decay + p*(1-decay); grad_norm_host (nullable) receives the pre-clip global
 * norm and makes the call synchronous. */
DDIF_API int ddif_optim_step(ddif_optim_t h, float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                             float max_grad_norm, int ema_mode, float ema_decay, float* grad_norm_host, void* stream);

/* ---- training: train-mode plans, the native training step (ddif_plan_train_step below), building blocks ---- */

/* Train-mode network (UNetSR3 under .train(): nn.Dropout(p) between SiLU and conv of every ResnetBlock Block,
 * models/sr3_dwt.py:295, and DropPath(0.2) on every decoder FFN, :534,576).  A train-mode plan runs the same entry points
 * (ddif_plan_set_cond / ddif_plan_forward / ddif_plan_q_sample_forward) with the masks below applied; it starts with identity
 * masks.  Masks hold 0 or 1/(1-p) (what nn.Dropout multiplies by) / per-sample DropPath scales in {0, 1/(1-p)}. */
DDIF_API int ddif_plan_create_train(ddif_plan_t* out, ddif_net_t net, int B, int H, int W);
/* ---- native training step (reference diffusion_engine.py:230-233: `diff_loss, recon = diffusion(res, cond=cond); diff_loss.backward()`).
 * A train-mode plan also holds the REVERSE launch program of the whole denoiser.  Per iteration:
 *   ddif_net_refresh (weights changed)  ->  ddif_plan_set_cond  ->  masks (ddif_plan_train_random_masks / _set_dropout / _set_droppath)
 *   ->  ddif_plan_train_step: x_t = a x0 + s noise, train-mode forward, L1 loss against x0, backward.
 * ddif_plan_train_bind names the gradient tensors once: n (state-dict key, device pointer) pairs covering every learnable tensor, reference
 * layouts (conv (Cout,Cin,k,k), Linear (out,in), vectors); the step WRITES them (no accumulation).  loss_dev: one device float (mean
 * absolute error, F.l1_loss); pred (nullable): the network output (B,C,H,W).  a / s / t: arrays of B floats (sqrt(alpha_bar_t),
 * sqrt(1 - alpha_bar_t), t) in HOST or DEVICE memory (device: no synchronisation per iteration); self_cond nullable (B,C,H,W).  Deterministic: fixed-order reductions, no atomics. */
DDIF_API int ddif_plan_train_bind(ddif_plan_t plan, int n, const char* const* keys, float* const* grads_dev);
DDIF_API int ddif_plan_train_num_grads(ddif_plan_t plan, int* n);
/* the same on a GIVEN network input and target (no q_sample): what the gradient parity tests drive with the reference's own tensors */
DDIF_API int ddif_plan_train_forward_backward(ddif_plan_t plan, const float* x, const float* time_host, const float* self_cond, const float* target,
                                              float* loss_dev, float* pred, void* stream);
DDIF_API int ddif_plan_train_step(ddif_plan_t plan, const float* x0, const float* noise, const float* sqrt_ac_host, const float* sqrt_1mac_host,
                                  const float* time_host, const float* self_cond, float* loss_dev, float* pred, void* stream);
DDIF_API int ddif_plan_train_info(ddif_plan_t plan, int* n_dropout_sites, int* n_droppath_sites);  /* sites in execution order */
DDIF_API int ddif_plan_train_site(ddif_plan_t plan, int site, int* C, int* H, int* W);             /* mask shape (B, C, H, W) */
/* explicit masks (parity with a reference run whose masks were captured): mask (B,C,H,W) device; scales HOST [n_droppath][B] */
DDIF_API int ddif_plan_train_set_dropout(ddif_plan_t plan, int site, const float* mask, void* stream);
DDIF_API int ddif_plan_train_set_droppath(ddif_plan_t plan, const float* scales_host, void* stream);
/* fresh masks from the counter-based generator, keyed by (seed, site, tile0 + sample, element): independent of the batch split */
DDIF_API int ddif_plan_train_random_masks(ddif_plan_t plan, uint64_t seed, uint64_t tile0, float p_dropout, float p_droppath, void* stream);
/* read the masks in force back (what nn.Dropout / DropPath would have drawn): mask (B,C,H,W) device; scales DEVICE [n_droppath][B] */
DDIF_API int ddif_plan_train_get_dropout(ddif_plan_t plan, int site, float* mask, void* stream);
DDIF_API int ddif_plan_train_get_droppath(ddif_plan_t plan, float* scales_dev, void* stream);

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

/* ---- measurement -------------------------------------------------------------------------------------------- */

/* Bracket EVERY launch of one denoising step out of `every_n_steps` with HIP events on the launch stream.  A step is either
 * profiled completely or not at all: when fewer than one step's worth of the `max_events` event pairs are left, profiling
 * stops, and `steps_recorded` says how many whole steps the sums cover.  Collect after the stream has been synchronised. */
DDIF_API int ddif_prof_begin(ddif_plan_t plan, int every_n_steps, int max_events);
typedef struct ddif_prof_result {
    int64_t launches;        /* timed launches */
    double total_ms;         /* sum of their durations */
    double total_flop;       /* algorithmic flops of the timed launches (2*M*N*K, unpadded) */
    double total_bytes;      /* algorithmic activation bytes read + written by them */
    char kernel_name[128];
    int64_t steps_recorded;  /* whole denoising steps the sums (and ddif_prof_classes) cover */
    int64_t launches_per_step; /* launches of the step program (event pairs one profiled step consumes) */
    double total_mfma_flop;  /* the same flops weighted by what the kernel ISSUES on the matrix pipe, in units of the dense 16-bit MFMA rate:
                              * x3 on the f16x2 path (three half products per fp32 product), x6 on bf16x3, x16 on the exact fp32 MFMA (which
                              * runs at 1/16 of that rate): total_mfma_flop / time / 2516.8 TF = the fraction of the matrix pipe's peak in use */
} ddif_prof_result;
DDIF_API int ddif_prof_collect(ddif_plan_t plan, ddif_prof_result* out);
/* Per-class breakdown of the same profiled steps (EVERY launch of a profiled step is bracketed): six entries -- 3x3 convs and
 * 1x1 convs of the levels with more than 256 pixels per sample, everything at the low-resolution levels, bottleneck
 * attention, softmax statistics, other.  Valid after ddif_prof_collect. */
typedef struct ddif_prof_class {
    int64_t launches;
    double total_ms, total_flop, total_bytes;
    char name[64];
    double total_mfma_flop;  /* as in ddif_prof_result */
} ddif_prof_class;
DDIF_API int ddif_prof_classes(ddif_plan_t plan, ddif_prof_class* out6);
/* launches of the step program (one denoising step = this many event pairs when profiled) and of the set_cond program */
DDIF_API int ddif_plan_num_launches(ddif_plan_t plan, int* step_launches, int* cond_launches);

/* flops / bytes of one denoising step and of set_cond for this plan (algorithmic, SURVEY.md 8d accounting) */
DDIF_API int ddif_plan_cost(ddif_plan_t plan, double* step_flop, double* step_bytes, double* cond_flop, double* cond_bytes);

/* Device memory of a plan: everything it allocated, of which `arena_bytes` hold the activations of the denoising-step program
 * (placed by liveness: a buffer is reused as soon as the last launch that touches its tensor has been enqueued); `unaliased_bytes`
 * is what the same activations would take with one buffer per tensor. */
DDIF_API int ddif_plan_memory(ddif_plan_t plan, int64_t* total_bytes, int64_t* arena_bytes, int64_t* unaliased_bytes);

/* Arithmetic of the convolutions of INFERENCE plans created afterwards (process-wide; also DDIF_MATH=bf16 in the environment):
 *   DDIF_MATH_SPLIT (default): fp32-class results -- f16x2 / bf16x3 split products or the exact fp32 MFMA; the parity configuration.
 *   DDIF_MATH_BF16: the THROUGHPUT variant BASELINE configs[1] names ("bf16"; the reference would get it from torch.autocast around
 *     models/sr3_dwt.py's nn.Conv2d calls): conv operands rounded once to bf16, one v_mfma_f32_32x32x16_bf16 product, fp32 accumulation;
 *     tensors in memory, GroupNorm statistics, attention and the sampler update stay fp32.  Not a parity configuration: bench.py reports
 *     its drift against the split path next to its rate.  Training plans ignore it.
 * Returns DDIF_ERR_INVALID for another value. */
#define DDIF_MATH_SPLIT 0
#define DDIF_MATH_BF16 1
DDIF_API int ddif_set_math_mode(int mode);
DDIF_API int ddif_get_math_mode(void);

/* Range guard of the f16x2 split (the default arithmetic of inference plans).  An IEEE half carries 2^4 x the activation, so the path is exact
 * only for |x| < 4094.  Behind a GroupNorm that bound is proven on the host from gamma / beta; a conv WITHOUT a GroupNorm in front (the decoder's
 * feed-forward pair, up-sampling convs, CondInjection.x_conv: models/sr3_dwt.py:528-533, 266-273, 380-396) stages its RAW input, which a trained
 * checkpoint (utils/misc.py:89-122 loads any) may push past it -- the half becomes inf and the output NaN.  Those launches therefore watch what
 * they stage and set a sticky per-plan flag.
 *   ddif_plan_range_status: synchronises `stream`, writes 1 to *overflow if a launch of this plan staged a value outside the range since the
 *     last call (and clears the flag), else 0, and returns DDIF_OK; called once per sampler / forward call by the Python layer, which then
 *     rebuilds the plan under ddif_set_f16_raw(0) and repeats the call.  A caller that does not want to fall back treats 1 as DDIF_ERR_RANGE.
 *   ddif_set_f16_raw(0): plans created afterwards keep convs on raw inputs on bf16x3 (full fp32 exponent range, six products instead of three);
 *     1 (default; DDIF_F16_RAW=0 in the environment for 0) = f16x2 with the watch.  GroupNorm-prologue convs are unaffected. */
DDIF_API int ddif_plan_range_status(ddif_plan_t plan, void* stream, int* overflow);
DDIF_API int ddif_set_f16_raw(int on);
DDIF_API int ddif_get_f16_raw(void);

/* TEST HOOK: cap the persistent grid (workgroups per conv launch) of plans created afterwards; 0 removes the cap.
 * Results do not depend on the cap (work items are walked in a fixed order per workgroup and every reduction has a
 * fixed order); tests use it to make small cases walk many work items per workgroup, across sample boundaries,
 * which is the regime the B=64 benchmark configuration runs in. */
DDIF_API int ddif_debug_set_grid_cap(int max_workgroups);

DDIF_API const char* ddif_last_error(void);
DDIF_API const char* ddif_version(void);
/* 1 when built for the test-only host emulator (never shipped), 0 for the gfx950 build */
DDIF_API int ddif_is_emulated(void);

#ifdef __cplusplus
}
#endif
#endif /* DDIF_H_ */
