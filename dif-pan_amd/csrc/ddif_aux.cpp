// Host side of the kernels around the denoising loop (kernels_aux.h): cond assembly + Haar, validation metrics, fused
// optimizer step.  extern "C" entry points are declared in include/ddif.h.
#include "ddif_net.h"
#include "kernels_aux.h"

namespace ddif {
namespace {
inline dim3 grid_for(size_t n) {
    size_t g = (n + 255) / 256;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return dim3((unsigned)g);
}
// scratch for the metric partials, grown on demand, one per device (handles are single-threaded by contract)
struct Scratch {
    double* p = nullptr;
    size_t n = 0;
};
Scratch g_scratch[64];
int scratch(size_t n, double** out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    Scratch& s = g_scratch[dev];
    if (s.n < n) {
        if (s.p) DDIF_HIPCHK(hipFree(s.p));
        s.p = nullptr;
        s.n = 0;
        DDIF_HIPCHK(hipMalloc((void**)&s.p, n * sizeof(double)));
        s.n = n;
    }
    *out = s.p;
    return 0;
}
}  // namespace

struct Optim {
    int device = 0;
    std::vector<void*> allocs;
    OptimChunk* d_chunks = nullptr;
    double* d_partial = nullptr;
    float* d_normcoef = nullptr;
    int nchunks = 0;
    int64_t nparams = 0;
    ~Optim() {
        for (void* p : allocs) (void)hipFree(p);
    }
};
}  // namespace ddif

struct ddif_optim {
    ddif::Optim o;
};

extern "C" {

int ddif_cond_assemble(const float* lms_raw, const float* pan_raw, float division, int B, int C, int P, int H, int W, int wavelet_order,
                       float* cond_out, void* stream) {
    if (!lms_raw || !pan_raw || !cond_out) return ddif::fail(DDIF_ERR_INVALID, "ddif_cond_assemble: NULL argument");
    if (B < 1 || C < 1 || P < 1 || H < 2 || W < 2 || (H & 1) || (W & 1)) return ddif::fail(DDIF_ERR_INVALID, "ddif_cond_assemble: H=%d W=%d must be even and >= 2", H, W);
    if (!(division > 0.f) || wavelet_order < 0 || wavelet_order > 1) return ddif::fail(DDIF_ERR_INVALID, "ddif_cond_assemble: bad division / wavelet_order");
    const size_t total = (size_t)B * (2 * C + 4 * P) * H * W;
    hipLaunchKernelGGL(ddif::cond_assemble_kernel, ddif::grid_for(total), dim3(256), 0, (hipStream_t)stream, lms_raw, pan_raw, division, B, C, P, H, W,
                       wavelet_order, cond_out);
    DDIF_HIPCHK(hipGetLastError());
    return DDIF_OK;
}

int ddif_metrics(const float* gt, const float* pred, int B, int C, int H, int W, float ergas_ratio, float* out, void* stream) {
    if (!gt || !pred || !out) return ddif::fail(DDIF_ERR_INVALID, "ddif_metrics: NULL argument");
    if (B < 1 || C < 1 || H < 2 || W < 2 || !(ergas_ratio > 0.f)) return ddif::fail(DDIF_ERR_INVALID, "ddif_metrics: bad shape / ratio");
    const int n = (H - 1) * (W - 1);
    int nchunk = (n + 256 * 8 - 1) / (256 * 8);
    if (nchunk < 1) nchunk = 1;
    if (nchunk > 64) nchunk = 64;
    double* ws = nullptr;
    if (int e = ddif::scratch((size_t)B * C * 6 + (size_t)B * nchunk * 2, &ws)) return e;
    double* sums = ws;
    double* part = ws + (size_t)B * C * 6;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ddif::metric_channel_sums_kernel, dim3(C, B), dim3(256), 64, s, gt, pred, C, H, W, sums);
    hipLaunchKernelGGL(ddif::metric_sam_kernel, dim3(nchunk, B), dim3(256), 64, s, gt, pred, C, H, W, nchunk, part);
    hipLaunchKernelGGL(ddif::metric_finalize_kernel, dim3((B + 63) / 64), dim3(64), 0, s, (const double*)sums, (const double*)part, B, C, H, W, nchunk, ergas_ratio, out);
    DDIF_HIPCHK(hipGetLastError());
    return DDIF_OK;
}

int ddif_ssim(const float* gt, const float* pred, int B, int C, int H, int W, float data_range, float* out, void* stream) {
    if (!gt || !pred || !out) return ddif::fail(DDIF_ERR_INVALID, "ddif_ssim: NULL argument");
    if (B < 1 || C < 1 || H < 7 || W < 7 || !(data_range > 0.f)) return ddif::fail(DDIF_ERR_INVALID, "ddif_ssim: images must be at least 7x7 (the window) and data_range > 0");
    const int n = (H - 6) * (W - 6);
    int nchunk = (n + 255) / 256;
    if (nchunk > 64) nchunk = 64;
    double* part = nullptr;
    if (int e = ddif::scratch((size_t)B * C * nchunk, &part)) return e;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ddif::ssim_window_kernel, dim3(nchunk, C, B), dim3(256), 64, s, gt, pred, C, H, W, nchunk, data_range, part);
    hipLaunchKernelGGL(ddif::ssim_finalize_kernel, dim3((B + 63) / 64), dim3(64), 0, s, (const double*)part, B, C, H, W, nchunk, out);
    DDIF_HIPCHK(hipGetLastError());
    return DDIF_OK;
}

int ddif_optim_create(ddif_optim_t* out, int n_tensors, const int64_t* sizes, float* const* params, const float* const* grads,
                      float* const* ema, int device) {
    return ddif_optim_create_ex(out, n_tensors, sizes, params, grads, ema, nullptr, nullptr, device);
}

int ddif_optim_create_ex(ddif_optim_t* out, int n_tensors, const int64_t* sizes, float* const* params, const float* const* grads,
                         float* const* ema, float* const* exp_avg, float* const* exp_avg_sq, int device) {
    if ((exp_avg == nullptr) != (exp_avg_sq == nullptr)) return ddif::fail(DDIF_ERR_INVALID, "ddif_optim_create_ex: give both moment arrays or neither");
    if (!out || n_tensors < 1 || !sizes || !params || !grads) return ddif::fail(DDIF_ERR_INVALID, "ddif_optim_create: bad arguments");
    *out = nullptr;
    int prev = -1;
    (void)hipGetDevice(&prev);
    DDIF_HIPCHK(hipSetDevice(device));
    std::unique_ptr<ddif_optim> h(new ddif_optim());
    ddif::Optim& o = h->o;
    o.device = device;
    constexpr int CH = 4096;
    std::vector<ddif::OptimChunk> chunks;
    int rc = DDIF_OK;
    for (int t = 0; t < n_tensors && rc == DDIF_OK; ++t) {
        const int64_t n = sizes[t];
        if (n < 0 || !params[t] || !grads[t]) {
            rc = ddif::fail(DDIF_ERR_INVALID, "ddif_optim_create: tensor %d has a NULL pointer or a negative size", t);
            break;
        }
        if (n == 0) continue;
        float *m = nullptr, *v = nullptr;
        if (exp_avg) {  // caller-owned moments (checkpointable optimizer state): borrowed like params / grads, NOT zeroed here
            m = exp_avg[t];
            v = exp_avg_sq[t];
            if (!m || !v) {
                rc = ddif::fail(DDIF_ERR_INVALID, "ddif_optim_create_ex: tensor %d has a NULL moment pointer", t);
                break;
            }
        } else {
            if (hipMalloc((void**)&m, (size_t)n * 4) != hipSuccess || hipMalloc((void**)&v, (size_t)n * 4) != hipSuccess) {
                rc = ddif::fail(DDIF_ERR_HIP, "ddif_optim_create: hipMalloc of the moment buffers failed");
                if (m) (void)hipFree(m);
                break;
            }
            o.allocs.push_back(m);
            o.allocs.push_back(v);
            (void)hipMemset(m, 0, (size_t)n * 4);
            (void)hipMemset(v, 0, (size_t)n * 4);
        }
        for (int64_t off = 0; off < n; off += CH) {
            ddif::OptimChunk c{};
            c.p = params[t] + off;
            c.g = grads[t] + off;
            c.m = m + off;
            c.v = v + off;
            c.ema = (ema && ema[t]) ? ema[t] + off : nullptr;
            c.n = (int)((n - off) < CH ? (n - off) : CH);
            chunks.push_back(c);
        }
        o.nparams += n;
    }
    if (rc == DDIF_OK && chunks.empty()) rc = ddif::fail(DDIF_ERR_INVALID, "ddif_optim_create: no parameters");
    if (rc == DDIF_OK) {
        o.nchunks = (int)chunks.size();
        void *dc = nullptr, *dp = nullptr, *dn = nullptr;
        if (hipMalloc(&dc, chunks.size() * sizeof(ddif::OptimChunk)) != hipSuccess || hipMalloc(&dp, chunks.size() * sizeof(double)) != hipSuccess ||
            hipMalloc(&dn, 64) != hipSuccess)
            rc = ddif::fail(DDIF_ERR_HIP, "ddif_optim_create: hipMalloc of the chunk table failed");
        if (dc) o.allocs.push_back(dc);
        if (dp) o.allocs.push_back(dp);
        if (dn) o.allocs.push_back(dn);
        if (rc == DDIF_OK) {
            o.d_chunks = (ddif::OptimChunk*)dc;
            o.d_partial = (double*)dp;
            o.d_normcoef = (float*)dn;
            if (hipMemcpy(dc, chunks.data(), chunks.size() * sizeof(ddif::OptimChunk), hipMemcpyHostToDevice) != hipSuccess)
                rc = ddif::fail(DDIF_ERR_HIP, "ddif_optim_create: upload of the chunk table failed");
        }
    }
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (rc != DDIF_OK) return rc;
    *out = h.release();
    return DDIF_OK;
}

void ddif_optim_destroy(ddif_optim_t h) { delete h; }

int ddif_optim_step(ddif_optim_t h, float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step, float max_grad_norm,
                    int ema_mode, float ema_decay, float* grad_norm_host, void* stream) {
    if (!h) return ddif::fail(DDIF_ERR_INVALID, "ddif_optim_step: NULL handle");
    if (step < 1 || ema_mode < 0 || ema_mode > 2) return ddif::fail(DDIF_ERR_INVALID, "ddif_optim_step: step is 1-based, ema_mode in {0,1,2}");
    ddif::Optim& o = h->o;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (prev != o.device) DDIF_HIPCHK(hipSetDevice(o.device));
    hipStream_t s = (hipStream_t)stream;
    ddif::OptimHyper hy{};
    hy.lr = lr;
    hy.beta1 = beta1;
    hy.beta2 = beta2;
    hy.eps = eps;
    hy.weight_decay = weight_decay;
    hy.bc1 = (float)(1.0 - std::pow((double)beta1, (double)step));
    hy.bc2_sqrt = (float)std::sqrt(1.0 - std::pow((double)beta2, (double)step));
    hy.max_norm = max_grad_norm;
    hy.ema_decay = ema_decay;
    hy.ema_mode = ema_mode;
    hipLaunchKernelGGL(ddif::optim_gradnorm_kernel, dim3(o.nchunks), dim3(256), 64, s, (const ddif::OptimChunk*)o.d_chunks, o.d_partial);
    hipLaunchKernelGGL(ddif::optim_clipcoef_kernel, dim3(1), dim3(256), 64, s, (const double*)o.d_partial, o.nchunks, max_grad_norm, o.d_normcoef);
    hipLaunchKernelGGL(ddif::optim_update_kernel, dim3(o.nchunks), dim3(256), 0, s, (const ddif::OptimChunk*)o.d_chunks, hy, (const float*)o.d_normcoef);
    int rc = DDIF_OK;
    if (hipGetLastError() != hipSuccess) rc = ddif::fail(DDIF_ERR_HIP, "ddif_optim_step: kernel launch failed");
    if (rc == DDIF_OK && grad_norm_host) {  // synchronous read-back (logging): the only host sync of the step, and optional
        if (hipMemcpyAsync(grad_norm_host, o.d_normcoef, sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
            rc = ddif::fail(DDIF_ERR_HIP, "ddif_optim_step: reading the gradient norm failed");
    }
    if (prev >= 0 && prev != o.device) (void)hipSetDevice(prev);
    return rc;
}

}  // extern "C"
