// Host side of the stateless backward ops (kernels_bwd_ops.h): thin launch wrappers behind the C ABI (include/ddif.h).  They run on
// the CURRENT device (the one the pointers live on) and on the given stream; no workspace, no handles.
#include "ddif_plan.h"
#include "kernels_bwd_ops.h"

namespace ddif {
static inline dim3 ops_grid(size_t n) {
    size_t g = (n + 255) / 256;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return dim3((unsigned)g);
}
// per-device scratch for the two-pass reductions (partials between the passes).  One host thread and one stream per device at a time,
// like every handle of this library; grown on demand, never shrunk, freed at process exit by the driver.
static double* ops_scratch(size_t doubles) {
    static double* buf[64] = {nullptr};
    static size_t cap[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (cap[dev] < doubles) {
        if (buf[dev]) {
            (void)hipDeviceSynchronize();  // nothing in flight may still read the old block
            (void)hipFree(buf[dev]);
            buf[dev] = nullptr;
            cap[dev] = 0;
        }
        const size_t want = doubles < (1u << 16) ? (1u << 16) : doubles;
        if (hipMalloc(reinterpret_cast<void**>(&buf[dev]), want * sizeof(double)) != hipSuccess) return nullptr;
        cap[dev] = want;
    }
    return buf[dev];
}
static inline int gn_chunks(size_t per_sample) {
    size_t n = per_sample / 4096;
    return (int)(n < 1 ? 1 : (n > 32 ? 32 : n));
}
static int ops_done(const char* what) {
    if (hipGetLastError() != hipSuccess) return fail(DDIF_ERR_HIP, "%s: kernel launch failed", what);
    return DDIF_OK;
}
}  // namespace ddif

extern "C" {

int ddif_dwconv3x3_bwd(const float* x, const float* w, const float* dy, int B, int C, int H, int W, float* dx, float* dw, void* stream) {
    if (!w || !dy || B < 1 || C < 1 || H < 1 || W < 1 || (dw && !x)) return ddif::fail(DDIF_ERR_INVALID, "ddif_dwconv3x3_bwd: bad argument");
    [[maybe_unused]] hipStream_t s = (hipStream_t)stream;
    if (dx) hipLaunchKernelGGL(ddif::dwconv3x3_bwd_dx_kernel, ddif::ops_grid((size_t)B * C * H * W), dim3(256), 0, s, dy, w, B, C, H, W, dx);
    if (dw) {
        double* part = ddif::ops_scratch((size_t)B * C * 9);
        if (!part) return ddif::fail(DDIF_ERR_HIP, "ddif_dwconv3x3_bwd: scratch allocation failed");
        hipLaunchKernelGGL(ddif::dwconv3x3_bwd_dw_partial_kernel, dim3(C, B), dim3(256), 9 * 256 * sizeof(double), s, x, dy, C, H, W, part);
        hipLaunchKernelGGL(ddif::dwconv3x3_bwd_dw_reduce_kernel, dim3((C * 9 + 255) / 256), dim3(256), 0, s, (const double*)part, B, C, dw);
    }
    return ddif::ops_done("ddif_dwconv3x3_bwd");
}

int ddif_film_bwd(const float* xc, const float* scale_shift, const float* dout, int B, int C, int H, int W, float* dxc, float* dscale_shift, void* stream) {
    if (!xc || !scale_shift || !dout || B < 1 || C < 1 || H < 1 || W < 1) return ddif::fail(DDIF_ERR_INVALID, "ddif_film_bwd: bad argument");
    hipLaunchKernelGGL(ddif::film_bwd_kernel, ddif::ops_grid((size_t)B * C * H * W), dim3(256), 0, (hipStream_t)stream, xc, scale_shift, dout, B, C, H * W, dxc,
                       dscale_shift);
    return ddif::ops_done("ddif_film_bwd");
}

int ddif_selfattn_core_bwd(const float* qkv, const float* dout, int B, int C, int H, int W, int heads, float* dqkv, void* stream) {
    if (!qkv || !dout || !dqkv || B < 1 || heads < 1 || C < heads || C % heads) return ddif::fail(DDIF_ERR_INVALID, "ddif_selfattn_core_bwd: bad argument");
    const int d = C / heads, n = H * W;
    if (n > 64 || d > 32) return ddif::fail(DDIF_ERR_INVALID, "ddif_selfattn_core_bwd: n = H*W <= 64 and head dim <= 32 (the engine's bottleneck attention) only");
    const size_t smem = ((size_t)4 * d * n + (size_t)2 * n * n + n) * sizeof(float);
    if (smem > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(ddif::selfattn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return ddif::fail(DDIF_ERR_HIP, "ddif_selfattn_core_bwd: hipFuncSetAttribute failed");
    // scale 1/sqrt(C), not 1/sqrt(d): models/sr3_dwt.py:352
    hipLaunchKernelGGL(ddif::selfattn_bwd_kernel, dim3(B * heads), dim3(256), smem, (hipStream_t)stream, qkv, dout, heads, d, n, 1.0f / sqrtf((float)C), dqkv);
    return ddif::ops_done("ddif_selfattn_core_bwd");
}

int ddif_linattn_core_bwd(const float* q_pre, const float* kv_pre, const float* dout, int B, int qd, int H, int W, int heads, float* dq_pre, float* dkv_pre,
                          void* stream) {
    if (!q_pre || !kv_pre || !dout || !dq_pre || !dkv_pre || B < 1 || heads < 1 || qd < heads || qd % heads)
        return ddif::fail(DDIF_ERR_INVALID, "ddif_linattn_core_bwd: bad argument");
    const int d = qd / heads;
    if (d > 32 || W > 64 || H < 1) return ddif::fail(DDIF_ERR_INVALID, "ddif_linattn_core_bwd: head dim <= 32 and W <= 64 only");
    if (H > 64) return ddif::fail(DDIF_ERR_INVALID, "ddif_linattn_core_bwd: H <= 64 only");
    const size_t smem = ((size_t)8 * d * W + (size_t)2 * d * d + d + (size_t)2 * d * H) * sizeof(float);
    if (smem > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(ddif::linattn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return ddif::fail(DDIF_ERR_HIP, "ddif_linattn_core_bwd: hipFuncSetAttribute failed");
    hipLaunchKernelGGL(ddif::linattn_bwd_kernel, dim3(B * heads), dim3(256), smem, (hipStream_t)stream, q_pre, kv_pre, dout, heads, d, H, W, 1.0f / sqrtf((float)d),
                       dq_pre, dkv_pre);
    return ddif::ops_done("ddif_linattn_core_bwd");
}

int ddif_linear_bwd(const float* x, const float* w, const float* dy, int B, int nin, int nout, float* dx, float* dw, float* db, void* stream) {
    if (!w || !dy || B < 1 || nin < 1 || nout < 1 || (dw && !x)) return ddif::fail(DDIF_ERR_INVALID, "ddif_linear_bwd: bad argument");
    const size_t total = (dx ? (size_t)B * nin : 0) + (dw ? (size_t)nout * nin : 0) + (db ? (size_t)nout : 0);
    if (total) hipLaunchKernelGGL(ddif::linear_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, w, dy, B, nin, nout, dx, dw, db);
    return ddif::ops_done("ddif_linear_bwd");
}

int ddif_swish_bwd(const float* x, const float* dy, int64_t n, float* dx, void* stream) {
    if (!x || !dy || !dx || n < 1) return ddif::fail(DDIF_ERR_INVALID, "ddif_swish_bwd: bad argument");
    hipLaunchKernelGGL(ddif::swish_bwd_kernel, ddif::ops_grid((size_t)n), dim3(256), 0, (hipStream_t)stream, x, dy, (size_t)n, dx);
    return ddif::ops_done("ddif_swish_bwd");
}

int ddif_l1_loss_bwd(const float* pred, const float* target, int64_t n, float upstream, float* dpred, void* stream) {
    if (!pred || !target || !dpred || n < 1) return ddif::fail(DDIF_ERR_INVALID, "ddif_l1_loss_bwd: bad argument");
    hipLaunchKernelGGL(ddif::l1_bwd_kernel, ddif::ops_grid((size_t)n), dim3(256), 0, (hipStream_t)stream, pred, target, (size_t)n, upstream, dpred);
    return ddif::ops_done("ddif_l1_loss_bwd");
}

int ddif_groupnorm_bwd(const float* x, const float* gamma, const float* dy, int B, int C, int H, int W, float* dx, float* dgamma, float* dbeta, double* workspace,
                       void* stream) {
    if (!x || !gamma || !dy || !workspace || B < 1 || C < 1 || H < 1 || W < 1) return ddif::fail(DDIF_ERR_INVALID, "ddif_groupnorm_bwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const size_t per = (size_t)C * H * W;
    const int nchunk = ddif::gn_chunks(per);
    double* part = ddif::ops_scratch((size_t)B * nchunk * 2);
    if (!part) return ddif::fail(DDIF_ERR_HIP, "ddif_groupnorm_bwd: scratch allocation failed");
    hipLaunchKernelGGL(ddif::gn_stats_partial_kernel, dim3(nchunk, B), dim3(256), 2 * 256 * sizeof(double), s, x, per, nchunk, part);
    hipLaunchKernelGGL(ddif::gn_bwd_plane_kernel, dim3(C, B), dim3(256), 2 * 256 * sizeof(double), s, x, dy, (const double*)part, nchunk, C, H * W, workspace);
    hipLaunchKernelGGL(ddif::gn_bwd_sample_finalize_kernel, dim3((B + 63) / 64), dim3(64), 0, s, (const double*)part, nchunk, gamma, B, C, H * W, workspace);
    if (dx) hipLaunchKernelGGL(ddif::gn_bwd_dx_kernel, ddif::ops_grid((size_t)B * C * H * W), dim3(256), 0, s, x, dy, gamma, (const double*)workspace, B, C, H * W, dx);
    if (dgamma || dbeta) hipLaunchKernelGGL(ddif::gn_bwd_affine_kernel, dim3((C + 255) / 256), dim3(256), 0, s, (const double*)workspace, B, C, dgamma, dbeta);
    return ddif::ops_done("ddif_groupnorm_bwd");
}

/* ---- forward counterparts for the training graph ---- */
int ddif_dwconv3x3_fwd(const float* x, const float* w, int B, int C, int H, int W, float* y, void* stream) {
    if (!x || !w || !y || B < 1 || C < 1 || H < 1 || W < 1) return ddif::fail(DDIF_ERR_INVALID, "ddif_dwconv3x3_fwd: bad argument");
    hipLaunchKernelGGL(ddif::dwconv3x3_fwd_kernel, ddif::ops_grid((size_t)B * C * H * W), dim3(256), 0, (hipStream_t)stream, x, w, B, C, H, W, y);
    return ddif::ops_done("ddif_dwconv3x3_fwd");
}
int ddif_groupnorm_fwd(const float* x, const float* gamma, const float* beta, const float* mask, int B, int C, int H, int W, int silu, float* y, void* stream) {
    if (!x || !gamma || !beta || !y || B < 1 || C < 1 || H < 1 || W < 1) return ddif::fail(DDIF_ERR_INVALID, "ddif_groupnorm_fwd: bad argument");
    const size_t per = (size_t)C * H * W;
    const int nchunk = ddif::gn_chunks(per);
    double* part = ddif::ops_scratch((size_t)B * nchunk * 2);
    if (!part) return ddif::fail(DDIF_ERR_HIP, "ddif_groupnorm_fwd: scratch allocation failed");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ddif::gn_stats_partial_kernel, dim3(nchunk, B), dim3(256), 2 * 256 * sizeof(double), s, x, per, nchunk, part);
    const size_t blocks = (per + 255) / 256;
    hipLaunchKernelGGL(ddif::gn_apply_nchw_kernel, dim3((unsigned)(blocks > 64 ? 64 : blocks), B), dim3(256), 0, s, x, (const double*)part, nchunk, gamma, beta, mask, C,
                       H * W, silu, y);
    return ddif::ops_done("ddif_groupnorm_fwd");
}
int ddif_swish_fwd(const float* x, int64_t n, float* y, void* stream) {
    if (!x || !y || n < 1) return ddif::fail(DDIF_ERR_INVALID, "ddif_swish_fwd: bad argument");
    hipLaunchKernelGGL(ddif::swish_fwd_kernel, ddif::ops_grid((size_t)n), dim3(256), 0, (hipStream_t)stream, x, (size_t)n, y);
    return ddif::ops_done("ddif_swish_fwd");
}
int ddif_film_fwd(const float* xc, const float* scale_shift, int B, int C, int H, int W, float* out, void* stream) {
    if (!xc || !scale_shift || !out || B < 1 || C < 1 || H < 1 || W < 1) return ddif::fail(DDIF_ERR_INVALID, "ddif_film_fwd: bad argument");
    hipLaunchKernelGGL(ddif::film_fwd_kernel, ddif::ops_grid((size_t)B * C * H * W), dim3(256), 0, (hipStream_t)stream, xc, scale_shift, B, C, H * W, out);
    return ddif::ops_done("ddif_film_fwd");
}
int ddif_add_scaled(const float* a, const float* f, const float* alpha, int B, int64_t per_sample, float* out, void* stream) {
    if (!a || !f || !out || B < 1 || per_sample < 1) return ddif::fail(DDIF_ERR_INVALID, "ddif_add_scaled: bad argument");
    hipLaunchKernelGGL(ddif::add_scaled_kernel, ddif::ops_grid((size_t)B * per_sample), dim3(256), 0, (hipStream_t)stream, a, f, alpha, (size_t)per_sample,
                       (size_t)B * per_sample, out);
    return ddif::ops_done("ddif_add_scaled");
}
int ddif_linear_fwd(const float* x, const float* w, const float* bias, int B, int nin, int nout, float* y, void* stream) {
    if (!x || !w || !y || B < 1 || nin < 1 || nout < 1) return ddif::fail(DDIF_ERR_INVALID, "ddif_linear_fwd: bad argument");
    hipLaunchKernelGGL(ddif::linear_fwd_kernel, dim3((B * nout + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, w, bias, B, nin, nout, y);
    return ddif::ops_done("ddif_linear_fwd");
}
int ddif_selfattn_core_fwd(const float* qkv, int B, int C, int H, int W, int heads, float* out, void* stream) {
    if (!qkv || !out || B < 1 || heads < 1 || C < heads || C % heads) return ddif::fail(DDIF_ERR_INVALID, "ddif_selfattn_core_fwd: bad argument");
    const int d = C / heads, n = H * W;
    if (n > 64 || d > 32) return ddif::fail(DDIF_ERR_INVALID, "ddif_selfattn_core_fwd: n = H*W <= 64 and head dim <= 32 only");
    const size_t smem = ((size_t)3 * d * n + (size_t)n * n) * sizeof(float);
    hipLaunchKernelGGL(ddif::selfattn_fwd_kernel, dim3(B * heads), dim3(256), smem, (hipStream_t)stream, qkv, heads, d, n, 1.0f / sqrtf((float)C), out);
    return ddif::ops_done("ddif_selfattn_core_fwd");
}
int ddif_linattn_core_fwd(const float* q_pre, const float* kv_pre, int B, int qd, int H, int W, int heads, float* out, void* stream) {
    if (!q_pre || !kv_pre || !out || B < 1 || heads < 1 || qd < heads || qd % heads) return ddif::fail(DDIF_ERR_INVALID, "ddif_linattn_core_fwd: bad argument");
    const int d = qd / heads;
    if (d > 32 || W > 64 || H < 1) return ddif::fail(DDIF_ERR_INVALID, "ddif_linattn_core_fwd: head dim <= 32 and W <= 64 only");
    if (H > 64) return ddif::fail(DDIF_ERR_INVALID, "ddif_linattn_core_fwd: H <= 64 only");
    const size_t smem = ((size_t)4 * d * W + (size_t)d * d + (size_t)2 * d * H) * sizeof(float);
    if (smem > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(ddif::linattn_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return ddif::fail(DDIF_ERR_HIP, "ddif_linattn_core_fwd: hipFuncSetAttribute failed");
    hipLaunchKernelGGL(ddif::linattn_fwd_kernel, dim3(B * heads), dim3(256), smem, (hipStream_t)stream, q_pre, kv_pre, heads, d, H, W, 1.0f / sqrtf((float)d), out);
    return ddif::ops_done("ddif_linattn_core_fwd");
}

int ddif_q_sample(const float* x0, const float* noise, const float* a, const float* s, int B, int64_t per_sample, float* out, void* stream) {
    if (!x0 || !noise || !a || !s || !out || B < 1 || per_sample < 1) return ddif::fail(DDIF_ERR_INVALID, "ddif_q_sample: bad argument");
    hipLaunchKernelGGL(ddif::q_sample_ew_kernel, ddif::ops_grid((size_t)B * per_sample), dim3(256), 0, (hipStream_t)stream, x0, noise, a, s, B, (size_t)per_sample, out);
    return ddif::ops_done("ddif_q_sample");
}
int ddif_l1_loss_fwd(const float* pred, const float* target, int64_t n, float* out, void* stream) {
    if (!pred || !target || !out || n < 1) return ddif::fail(DDIF_ERR_INVALID, "ddif_l1_loss_fwd: bad argument");
    hipLaunchKernelGGL(ddif::l1_fwd_kernel, dim3(1), dim3(256), 256 * sizeof(double), (hipStream_t)stream, pred, target, (size_t)n, out);
    return ddif::ops_done("ddif_l1_loss_fwd");
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------- launchers for the native training step (ddif_train.cpp)
namespace ddif {
namespace tk {
void linear_bwd(hipStream_t s, const float* x, const float* w, const float* dy, int B, int nin, int nout, float* dx, float* dw, float* db) {
    const int total = (dx ? B * nin : 0) + (dw ? nout * nin : 0) + (db ? nout : 0);
    hipLaunchKernelGGL(linear_bwd_kernel, dim3((total + 255) / 256), dim3(256), 0, s, x, w, dy, B, nin, nout, dx, dw, db);
}
void l1_fwd(hipStream_t s, const float* pred, const float* target, size_t n, float* out) {
    hipLaunchKernelGGL(l1_fwd_kernel, dim3(1), dim3(256), 256 * sizeof(double), s, pred, target, n, out);
}
void l1_bwd(hipStream_t s, const float* pred, const float* target, size_t n, float upstream, float* dpred) {
    hipLaunchKernelGGL(l1_bwd_kernel, ops_grid(n), dim3(256), 0, s, pred, target, n, upstream, dpred);
}
}  // namespace tk
}  // namespace ddif
