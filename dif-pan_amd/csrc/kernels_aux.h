// Kernels either side of the denoising loop: cond assembly + level-1 Haar analysis, the validation
// metrics (SAM / ERGAS / PSNR / CC) and the fused optimizer step (global-norm clip + AdamW + EMA).  All are
// HBM-bound streaming / reduction kernels: NCHW fp32 at the boundary, float4 where the layout allows, every reduction
// in a fixed order (no atomics) so results are bitwise reproducible.
#pragma once
#include "ddif_dev.h"

namespace ddif {

// ----------------------------------------------------------------------------------------------------------------
// cond = cat[lms, pan, bilinear_up2(wavelets)] / division  (reference diffusion_engine.py:221-228, 441-444) with
// wavelets = level-1 Haar ("db1") analysis: LL of lms, (H, V, D) details of pan (dataset/pan_dataset.py:73-81,139-142;
// dataset/hisr.py:48-59).  Haar as documented by PyWavelets: LL = (a+b+c+d)/2, cH = (a+b-c-d)/2 (rows axis),
// cV = (a-b+c-d)/2 (cols axis), cD = (a-b-c+d)/2 for the 2x2 block [[a, b], [c, d]].
// order 0 (PanCollection): [LL, H, D, V];  order 1 (CAVE / Harvard): [LL, H, V, D].
// The bilinear x2 up-sampling is F.interpolate(mode="bilinear", align_corners=False): src = (dst + 0.5)/2 - 0.5, clamped.
// One thread per output element of cond (B, 2C+4P, H, W); H and W even.
__global__ void cond_assemble_kernel(const float* lms, const float* pan, float div, int B, int C, int P, int H, int W,
                                     int order, float* cond) {
    const int CC = 2 * C + 4 * P, h2 = H / 2, w2 = W / 2;
    const size_t total = (size_t)B * CC * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const int ch = (int)((i / ((size_t)W * H)) % CC);
        const int b = (int)(i / ((size_t)W * H * CC));
        float v;
        if (ch < C) {
            v = lms[(((size_t)b * C + ch) * H + y) * W + x] / div;  // true division, as norm_func does (dataset/pan_dataset.py:127-134)
        } else if (ch < C + P) {
            v = pan[(((size_t)b * P + (ch - C)) * H + y) * W + x] / div;
        } else {
            const int wch = ch - C - P;  // channel of the wavelet stack: [LL x C | band1 x P | band2 x P | band3 x P]
            const float* src;
            int band;                    // 0 LL, 1 H, 2 V, 3 D
            if (wch < C) {
                src = lms + ((size_t)b * C + wch) * H * W;
                band = 0;
            } else {
                const int k = (wch - C) / P, pc = (wch - C) % P;
                src = pan + ((size_t)b * P + pc) * H * W;
                band = order == 0 ? (k == 0 ? 1 : (k == 1 ? 3 : 2)) : k + 1;
            }
            float fy = 0.5f * (y + 0.5f) - 0.5f, fx = 0.5f * (x + 0.5f) - 0.5f;
            if (fy < 0.f) fy = 0.f;
            if (fx < 0.f) fx = 0.f;
            const int y0 = (int)fy, x0 = (int)fx;
            const int y1 = y0 + (y0 < h2 - 1 ? 1 : 0), x1 = x0 + (x0 < w2 - 1 ? 1 : 0);
            const float ly = fy - y0, lx = fx - x0;
            auto wv = [&](int yy, int xx) {
                const float* p = src + (size_t)(2 * yy) * W + 2 * xx;
                const float a = p[0], bb = p[1], c = p[W], d = p[W + 1];  // wavelets of the RAW data, then / division (as the datasets do)
                return (band == 0 ? (a + bb + c + d) * 0.5f : (band == 1 ? (a + bb - c - d) * 0.5f : (band == 2 ? (a - bb + c - d) * 0.5f : (a - bb - c + d) * 0.5f))) / div;
            };
            v = (1.f - ly) * ((1.f - lx) * wv(y0, x0) + lx * wv(y0, x1)) + ly * ((1.f - lx) * wv(y1, x0) + lx * wv(y1, x1));
        }
        cond[i] = v;
    }
}

// ----------------------------------------------------------------------------------------------------------------
// Validation metrics of one (gt, pred) pair per image (utils/_metric_legacy.py:299-379 analysis_accu with
// flag_cut_bounds=True, dim_cut=1, choices=5 -- what AnalysisPanAcc runs, utils/metric.py:24-98): the LAST row and
// column are dropped (`img[0:-1, 0:-1]`), then
//   SAM   = mean over pixels with |a||b| > 0 of acos(<a,b> / (|a||b|)) (NaN -> 0), rounded to 6 digits, * 180 / 3.14159256
//   ERGAS = 100 / ratio * sqrt(mean_c( mean((a_c - b_c)^2) / mean(a_c)^2 ))
//   PSNR  = mean_c( -20 log10(1 / rmse_c) )          (the reference's sign: NEGATIVE of the conventional PSNR)
//   CC    = mean_c( cov(a_c, b_c) / sqrt(var(a_c) var(b_c)) )   (sums form, as written there)
// Pass 1 (metric_channel_sums_kernel): grid (C, B): six fp64 sums per (image, channel) over the cropped region.
// Pass 2 (metric_sam_kernel): grid (chunks, B): per-pixel angle sum and count, one fp64 pair per workgroup.
// Pass 3 (metric_finalize_kernel): one thread per image.
__device__ __forceinline__ double block_sum_256(double v, double* red) {  // all 256 threads; result valid in every thread
    v = wave_sum(v);
    const int tid = threadIdx.x;
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void metric_channel_sums_kernel(const float* gt, const float* pred, int C, int H, int W, double* sums) {
    DDIF_DYN_SMEM(smem_);  // 64 bytes of dynamic LDS (the host emulator has no static __shared__)
    double* red = reinterpret_cast<double*>(smem_);
    const int c = blockIdx.x, b = blockIdx.y, hc = H - 1, wc = W - 1;
    const float* a = gt + ((size_t)b * C + c) * H * W;
    const float* p = pred + ((size_t)b * C + c) * H * W;
    double s[6] = {0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < hc * wc; i += 256) {
        const int y = i / wc, x = i - y * wc;
        const double av = a[(size_t)y * W + x], pv = p[(size_t)y * W + x];
        s[0] += av;
        s[1] += pv;
        s[2] += av * pv;
        s[3] += av * av;
        s[4] += pv * pv;
        s[5] += (av - pv) * (av - pv);
    }
    for (int k = 0; k < 6; ++k) {
        const double t = block_sum_256(s[k], red);
        if (threadIdx.x == 0) sums[((size_t)b * C + c) * 6 + k] = t;
    }
}

__global__ __launch_bounds__(256) void metric_sam_kernel(const float* gt, const float* pred, int C, int H, int W, int nchunk, double* part) {
    DDIF_DYN_SMEM(smem_);  // 64 bytes of dynamic LDS (the host emulator has no static __shared__)
    double* red = reinterpret_cast<double*>(smem_);
    const int b = blockIdx.y, hc = H - 1, wc = W - 1, n = hc * wc;
    const size_t plane = (size_t)H * W;
    const float* a = gt + (size_t)b * C * plane;
    const float* p = pred + (size_t)b * C * plane;
    double ang = 0.0, cnt = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += nchunk * 256) {
        const int y = i / wc, x = i - y * wc;
        const size_t o = (size_t)y * W + x;
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;  // fp32 like torch.sum(img_base * img_out, 2)
        for (int c = 0; c < C; ++c) {
            const float av = a[c * plane + o], pv = p[c * plane + o];
            s1 += av * pv;
            s2 += av * av;
            s3 += pv * pv;
        }
        const float t = sqrtf(s2 * s3);
        if (t > 0.f) cnt += 1.0;
        const float an = acosf(s1 / t);
        if (an == an) ang += (double)an;  // NaN -> 0
    }
    const double ta = block_sum_256(ang, red), tc = block_sum_256(cnt, red);
    if (threadIdx.x == 0) {
        part[((size_t)b * nchunk + blockIdx.x) * 2 + 0] = ta;
        part[((size_t)b * nchunk + blockIdx.x) * 2 + 1] = tc;
    }
}

__global__ void metric_finalize_kernel(const double* sums, const double* part, int B, int C, int H, int W, int nchunk, float ratio, float* out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double n = (double)(H - 1) * (W - 1);
    double ang = 0, cnt = 0;
    for (int k = 0; k < nchunk; ++k) {
        ang += part[((size_t)b * nchunk + k) * 2];
        cnt += part[((size_t)b * nchunk + k) * 2 + 1];
    }
    double av = cnt == 0 ? ang : ang / cnt;
    av = rint(av * 1e6) / 1e6;
    const double sam = av * 180.0 / 3.14159256;  // (sic: the reference's constant)
    double summ = 0, psnr = 0, cc = 0;
    for (int c = 0; c < C; ++c) {
        const double* s = sums + ((size_t)b * C + c) * 6;
        const double ma = s[0] / n, mp = s[1] / n, mse = s[5] / n;
        summ += mse / (ma * ma);
        psnr += -20.0 * (log(1.0 / sqrt(mse)) / log(10.0));
        const double c1 = s[2] - n * ma * mp, c2 = s[4] - n * mp * mp, c3 = s[3] - n * ma * ma;
        cc += c1 / sqrt(c2 * c3);
    }
    out[b * 4 + 0] = (float)sam;
    out[b * 4 + 1] = (float)(100.0 / ratio * sqrt(summ / C));
    out[b * 4 + 2] = (float)(psnr / C);
    out[b * 4 + 3] = (float)(cc / C);
}

// ----------------------------------------------------------------------------------------------------------------
// SSIM of the validation pass (reference utils/metric.py:153-166: skimage.metrics.structural_similarity(gt, pred, channel_axis=0) with
// library defaults -- 7x7 uniform window, K1 = 0.01, K2 = 0.03, sample covariance (NP / (NP - 1)), mean over the image cropped by
// (win - 1) / 2 = 3 pixels per side and over the channels; for float images without data_range the library takes dmax - dmin of the dtype
// range (-1, 1), i.e. 2.0 -- the caller passes it).  The crop removes exactly the pixels whose window touches the border, so the filter's
// boundary rule never enters.  skimage is NOT in the build image: this restatement is parity-unpinned against the library itself.
//   ssim_window_kernel   grid (row chunks, C, B): one thread per interior pixel, 49-tap window sums in fp64, fp64 partial per workgroup
//   ssim_finalize_kernel per image: fixed-order sum of the partials / (C * (H-6) * (W-6))
__global__ __launch_bounds__(256) void ssim_window_kernel(const float* gt, const float* pred, int C, int H, int W, int nchunk, float data_range, double* part) {
    DDIF_DYN_SMEM(smem_);
    double* red = reinterpret_cast<double*>(smem_);
    const int b = blockIdx.z, c = blockIdx.y, hc = H - 6, wc = W - 6, n = hc * wc;
    const size_t plane = (size_t)H * W;
    const float* a = gt + ((size_t)b * C + c) * plane;
    const float* p = pred + ((size_t)b * C + c) * plane;
    const double c1 = (0.01 * (double)data_range) * (0.01 * (double)data_range), c2 = (0.03 * (double)data_range) * (0.03 * (double)data_range);
    const double np_ = 49.0, cov_norm = np_ / (np_ - 1.0);
    double acc = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += nchunk * 256) {
        const int y = i / wc, x = i - y * wc;  // top-left corner of the window = centre (y + 3, x + 3)
        double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
        for (int dy = 0; dy < 7; ++dy) {
            const float* ar = a + (size_t)(y + dy) * W + x;
            const float* pr = p + (size_t)(y + dy) * W + x;
            for (int dx = 0; dx < 7; ++dx) {
                const double av = ar[dx], pv = pr[dx];
                sx += av;
                sy += pv;
                sxx += av * av;
                syy += pv * pv;
                sxy += av * pv;
            }
        }
        const double ux = sx / np_, uy = sy / np_;
        const double vx = cov_norm * (sxx / np_ - ux * ux), vy = cov_norm * (syy / np_ - uy * uy), vxy = cov_norm * (sxy / np_ - ux * uy);
        acc += ((2.0 * ux * uy + c1) * (2.0 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2));
    }
    const double t = block_sum_256(acc, red);
    if (threadIdx.x == 0) part[((size_t)b * C + c) * nchunk + blockIdx.x] = t;
}

__global__ void ssim_finalize_kernel(const double* part, int B, int C, int H, int W, int nchunk, float* out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double s = 0.0;
    for (int c = 0; c < C; ++c) {
        double sc = 0.0;
        for (int k = 0; k < nchunk; ++k) sc += part[((size_t)b * C + c) * nchunk + k];
        s += sc / ((double)(H - 6) * (W - 6));
    }
    out[b] = (float)(s / C);
}

// ----------------------------------------------------------------------------------------------------------------
// Fused optimizer step of the training loop (reference diffusion_engine.py:237-241): clip_grad_norm_(0.003)
// (utils/misc.py:25-36 -> torch.nn.utils.clip_grad_norm_), torch.optim.AdamW.step, EmaUpdater.update
// (utils/optim_utils.py:43-58).  Multi-tensor: a device table of chunks (tensor pointers + offset + length, <= 4096
// elements each) covers every parameter; three launches per step whatever the number of tensors:
//   optim_gradnorm_kernel   one fp64 sum of squares per chunk
//   optim_clipcoef_kernel   total norm (fixed-order sum over the chunk partials) -> coef = min(1, max_norm / (norm + 1e-6))
//   optim_update_kernel     g *= coef;  p *= 1 - lr*wd;  m, v moments;  p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps);
//                           EMA: copy (iteration <= start_iter) or ema = ema*decay + p*(1-decay)
struct OptimChunk {
    float* p;
    const float* g;
    float* m;
    float* v;
    float* ema;  // may be null
    int n;
    int pad;
};
struct OptimHyper {
    float lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt;  // bc1 = 1 - beta1^t, bc2_sqrt = sqrt(1 - beta2^t)
    float max_norm;                                            // <= 0: no clipping
    float ema_decay;
    int ema_mode;                                              // 0 none, 1 copy, 2 lerp
};

__global__ __launch_bounds__(256) void optim_gradnorm_kernel(const OptimChunk* chunks, double* partial) {
    DDIF_DYN_SMEM(smem_);  // 64 bytes of dynamic LDS (the host emulator has no static __shared__)
    double* red = reinterpret_cast<double*>(smem_);
    const OptimChunk c = chunks[blockIdx.x];
    double s = 0.0;
    for (int i = threadIdx.x; i < c.n; i += 256) {
        const double g = c.g[i];
        s += g * g;
    }
    const double t = block_sum_256(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void optim_clipcoef_kernel(const double* partial, int nchunks, float max_norm, float* out /* [0] = total norm, [1] = coef */) {
    DDIF_DYN_SMEM(smem_);  // 64 bytes of dynamic LDS (the host emulator has no static __shared__)
    double* red = reinterpret_cast<double*>(smem_);
    double s = 0.0;
    for (int i = threadIdx.x; i < nchunks; i += 256) s += partial[i];
    const double t = block_sum_256(s, red);
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(t);
        float coef = 1.f;
        if (max_norm > 0.f) {
            coef = max_norm / (norm + 1e-6f);
            if (coef > 1.f) coef = 1.f;
        }
        out[0] = norm;
        out[1] = coef;
    }
}

__global__ __launch_bounds__(256) void optim_update_kernel(const OptimChunk* chunks, OptimHyper h, const float* normcoef) {
#pragma clang fp contract(off)
    const OptimChunk c = chunks[blockIdx.x];
    const float coef = normcoef[1];
    const float step_size = h.lr / h.bc1;
    for (int i = threadIdx.x; i < c.n; i += 256) {
        const float g = c.g[i] * coef;
        float p = c.p[i];
        p = p * (1.f - h.lr * h.weight_decay);            // decoupled weight decay (AdamW)
        const float m = c.m[i] + (g - c.m[i]) * (1.f - h.beta1);  // torch: exp_avg.lerp_(grad, 1 - beta1)
        const float v = c.v[i] * h.beta2 + (1.f - h.beta2) * g * g;
        const float denom = sqrtf(v) / h.bc2_sqrt + h.eps;
        p = p - step_size * (m / denom);
        c.p[i] = p;
        c.m[i] = m;
        c.v[i] = v;
        if (c.ema) {
            if (h.ema_mode == 1) c.ema[i] = p;
            else if (h.ema_mode == 2) c.ema[i] = c.ema[i] * h.ema_decay + p * (1.f - h.ema_decay);
        }
    }
}

}  // namespace ddif
