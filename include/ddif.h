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

/* The per-op entry points of round 2's op-by-op training tape (stateless forward / backward ops, ddif_convfwd_*, ddif_convbwd_*, ddif_blockbwd_*) are TEST-ONLY since
 * round 6: they stay in libddif.so as an independent cross-check of the reverse program above and are declared in include/ddif_testops.h. */

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
