// Host side of the 3x3 convolution backward op (kernels_bwd.h): the first building block of the training step
// (SURVEY.md 8(a) a15).  extern "C" entry points are declared in include/ddif.h.
#include <algorithm>

#include "ddif_plan.h"
#include "kernels_bwd.h"

namespace ddif {
// DDIF_TRAIN_X3=0: the training convs (forward and dgrad) on the exact-fp32 MFMA instead of the bf16x3 split products the inference path uses
// (same fp32-class accuracy, ~2.5x the matrix rate; kernels_conv.h MATH = 1).  The weight gradient always runs on the exact instruction.
static bool train_x3() {
    static const bool v = [] { const char* e = getenv("DDIF_TRAIN_X3"); return !e || atoi(e) != 0; }();
    return v;
}
static inline dim3 grid_for(size_t n) {
    size_t g = (n + 255) / 256;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return dim3((unsigned)g);
}

struct ConvBwd {
    int device = 0, B = 0, Cin = 0, Cout = 0, H = 0, W = 0;
    Net net;          // only its cfg is read (add_conv's launch closure); no weights
    Plan plan;        // owns every device buffer of this op
    PackedConv pc;    // the dgrad conv: Cout "input" channels -> Cin "output" channels, weights = wpack
    float *x_nhwc = nullptr, *dy_nhwc = nullptr, *wpack = nullptr, *wpack_x3 = nullptr, *partial = nullptr, *bpart = nullptr;
    Tensor dx;        // NHWC output of the dgrad conv
    std::vector<Op> prog;
    int n_chunks = 0, nb_pad = 0, nbchunk = 0;
    tk::WgradGeom g3, g1;      // weight-gradient launch geometry of the 3x3 / the 1x1 (centre-tap) form: the training step's (tk::wgrad_geom)
    float* wbpart = nullptr;   // per-split bias partials of the fused bias gradient
    bool centre_only = false;  // a 1x1 conv riding the 3x3 kernels: dW is (Cout, Cin) from the centre tap
};
int convbwd_init(ConvBwd& c, int B, int Cin, int Cout, int H, int W, int device);
void convbwd_core(ConvBwd& c, hipStream_t s, const float* w, bool want_dx, float* dw, float* db);
}  // namespace ddif

struct ddif_convbwd {
    ddif::ConvBwd c;
};

extern "C" {

int ddif_convbwd_create(ddif_convbwd_t* out, int B, int Cin, int Cout, int H, int W, int device) {
    if (!out || B < 1 || Cin < 4 || Cout < 4 || (Cin & 3) || (Cout & 3) || H < 1 || W < 1)
        return ddif::fail(DDIF_ERR_INVALID, "ddif_convbwd_create: B >= 1, 4 | Cin, 4 | Cout, H, W >= 1 required");
    *out = nullptr;
    int prev = -1;
    (void)hipGetDevice(&prev);
    DDIF_HIPCHK(hipSetDevice(device));
    std::unique_ptr<ddif_convbwd> h(new ddif_convbwd());
    const int rc = ddif::convbwd_init(h->c, B, Cin, Cout, H, W, device);
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (rc) return rc;
    *out = h.release();
    return DDIF_OK;
}
}  // extern "C"

namespace ddif {
// the caller has made `device` current
int convbwd_init(ConvBwd& c, int B, int Cin, int Cout, int H, int W, int device) {
    c.device = device;
    c.B = B; c.Cin = Cin; c.Cout = Cout; c.H = H; c.W = W;
    c.plan.net = &c.net;
    c.plan.B = B;
    c.plan.H = H;
    c.plan.W = W;
    int rc = 0;
    auto TRY = [&](int e) { if (!rc) rc = e; };
    TRY(c.plan.dalloc(&c.plan.zeros, (size_t)1024));
    if (!rc && hipMemset(c.plan.zeros, 0, 1024 * sizeof(float)) != hipSuccess) rc = ddif::fail(DDIF_ERR_HIP, "ddif_convbwd_create: hipMemset failed");
    TRY(c.plan.dalloc(&c.x_nhwc, (size_t)B * H * W * Cin));
    TRY(c.plan.dalloc(&c.dy_nhwc, (size_t)B * H * W * Cout));
    // dgrad conv: "cin" = Cout (16-channel chunks), "cout" = Cin
    c.n_chunks = (Cout + 15) / 16;
    c.nb_pad = (((Cin + 31) / 32) + 3) & ~3;
    TRY(c.plan.dalloc(&c.wpack, (size_t)c.nb_pad * c.n_chunks * 9 * 2 * 256));
    TRY(c.plan.dalloc(&c.wpack_x3, (size_t)c.nb_pad * c.n_chunks * 9 * 3 * 256));
    c.pc.w = c.wpack;
    c.pc.w_x3 = c.wpack_x3;
    c.pc.cin = Cout;
    c.pc.cout = Cin;
    c.pc.ks = 3;
    c.pc.ck = 16;
    c.pc.n_chunks = c.n_chunks;
    if (!rc) {
        ddif::ConvSpec s;
        s.pc = &c.pc;
        s.in0.p = c.dy_nhwc;
        s.in0.C = Cout;
        s.in0.H = H;
        s.in0.W = W;
        s.use_bias = false;
        s.exact = !train_x3();  // exact-fp32 MFMA (bitwise an fmaf chain) on request; bf16x3 split products by default
        s.name = "conv3x3.dgrad";
        TRY(c.plan.add_conv(c.prog, s, &c.dx));
    }
    // wgrad: the launches of the training step (tk::wgrad_geom / tk::wgrad below): the bf16x3 kernel where the width allows it, the fp32 kernel otherwise
    c.g3 = tk::wgrad_geom(B, Cin, Cout, H, W, false);
    c.g1 = tk::wgrad_geom(B, Cin, Cout, H, W, true);
    if (!rc && (c.g3.rb < 1 || c.g1.rb < 1)) rc = ddif::fail(DDIF_ERR_INVALID, "ddif_convbwd_create: W=%d is too wide for the wgrad kernel's LDS tiles", W);
    if (!rc) {
        TRY(c.plan.dalloc(&c.partial, std::max(c.g3.partial_floats, c.g1.partial_floats) + 64));
        TRY(c.plan.dalloc(&c.wbpart, (size_t)std::max(c.g3.nsplit, c.g1.nsplit) * c.g3.n_co * 32 + 64));
        c.nbchunk = (int)std::min<size_t>(512, std::max<size_t>(1, (size_t)B * H * W / 16));  // >= 16 pixels per chunk
        TRY(c.plan.dalloc(&c.bpart, (size_t)c.nbchunk * Cout));
        TRY(tk::wgrad_prepare());
    }
    return rc;
}

// x_nhwc / dy_nhwc hold the op's inputs; dX is left in c.dx (NHWC) when want_dx
void convbwd_core(ConvBwd& c, hipStream_t s, const float* w, bool want_dx, float* dw, float* db) {
    const int HW = c.H * c.W;
    if (want_dx) {
        const size_t nw = (size_t)c.nb_pad * c.n_chunks * 9 * 2 * 256;
        hipLaunchKernelGGL(pack_dgrad_weights_kernel, grid_for(nw), dim3(256), 0, s, w, c.Cout, c.Cin, c.n_chunks, c.nb_pad, c.wpack);
        hipLaunchKernelGGL(pack_weights_x3_kernel, grid_for(nw), dim3(256), 0, s, w, c.Cout, c.Cin, 3, 1, c.n_chunks, c.nb_pad, c.wpack_x3);
        StepCtx ctx;
        for (auto& op : c.prog) op.run(s, ctx);
    }
    if (dw) {  // (+ the bias gradient from the same launch: column sums of the dY bands already staged)
        tk::wgrad(s, c.x_nhwc, c.dy_nhwc, c.B, c.H, c.W, c.Cin, c.Cout, c.centre_only ? c.g1 : c.g3, c.centre_only, c.partial, dw, c.wbpart, db);
        return;
    }
    if (db) {
        hipLaunchKernelGGL(bias_grad_partial_kernel, dim3(c.nbchunk), dim3(256), 256 * sizeof(float), s, (const float*)c.dy_nhwc, (size_t)c.B * HW, c.Cout, c.nbchunk, c.bpart);
        hipLaunchKernelGGL(bias_grad_reduce_kernel, dim3(c.Cout), dim3(64), 64 * sizeof(float), s, (const float*)c.bpart, c.nbchunk, c.Cout, db);
    }
}
}  // namespace ddif

extern "C" {

void ddif_convbwd_destroy(ddif_convbwd_t h) { delete h; }

int ddif_convbwd_run(ddif_convbwd_t h, const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, void* stream) {
    if (!h || !dy || !w) return ddif::fail(DDIF_ERR_INVALID, "ddif_convbwd_run: NULL argument");
    if ((dw && !x)) return ddif::fail(DDIF_ERR_INVALID, "ddif_convbwd_run: dw needs x");
    ddif::ConvBwd& c = h->c;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (prev != c.device) DDIF_HIPCHK(hipSetDevice(c.device));
    hipStream_t s = (hipStream_t)stream;
    const int HW = c.H * c.W;
    hipLaunchKernelGGL(ddif::bwd_nchw_to_nhwc_kernel, ddif::grid_for((size_t)c.B * HW * c.Cout), dim3(256), ddif::TR_SMEM, s, dy, c.B, c.Cout, HW, c.dy_nhwc);
    if (dw) hipLaunchKernelGGL(ddif::bwd_nchw_to_nhwc_kernel, ddif::grid_for((size_t)c.B * HW * c.Cin), dim3(256), ddif::TR_SMEM, s, x, c.B, c.Cin, HW, c.x_nhwc);
    ddif::convbwd_core(c, s, w, dx != nullptr, dw, db);
    if (dx) hipLaunchKernelGGL(ddif::bwd_nhwc_to_nchw_kernel, ddif::grid_for((size_t)c.B * HW * c.Cin), dim3(256), ddif::TR_SMEM, s, (const float*)c.dx.p, c.B, c.Cin, HW, dx);
    int rc = DDIF_OK;
    if (hipGetLastError() != hipSuccess) rc = ddif::fail(DDIF_ERR_HIP, "ddif_convbwd_run: kernel launch failed");
    if (prev >= 0 && prev != c.device) (void)hipSetDevice(prev);
    return rc;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------- Block backward
namespace ddif {
struct BlockBwd {
    ConvBwd conv;      // owns the NHWC staging buffers: conv.x_nhwc = a (the conv's input), conv.dy_nhwc = dY, conv.dx = dA
    float *x_nhwc = nullptr, *mask_nhwc = nullptr, *dx_nhwc = nullptr, *S = nullptr;
    double *spart = nullptr, *cpart = nullptr;
    float* w3 = nullptr;  // ks == 1: the 1x1 weights embedded in a 3x3 tensor for the dgrad conv
    int nchunk = 32, ks = 3, pro = DDIF_BWD_PRO_GN_SILU, resample = DDIF_BWD_PLAIN;
    int H = 0, W = 0;  // of the op's INPUT x (the conv runs at 2H x 2W under DDIF_BWD_UP2)
};
}  // namespace ddif
struct ddif_blockbwd {
    ddif::BlockBwd b;
};

extern "C" {

int ddif_blockbwd_create(ddif_blockbwd_t* out, int B, int Cin, int Cout, int H, int W, int device) {
    return ddif_blockbwd_create_ex(out, B, Cin, Cout, H, W, 3, DDIF_BWD_PRO_GN_SILU, DDIF_BWD_PLAIN, device);
}

int ddif_blockbwd_create_ex(ddif_blockbwd_t* out, int B, int Cin, int Cout, int H, int W, int ks, int pro, int resample, int device) {
    if (!out || B < 1 || Cin < 4 || Cout < 4 || (Cin & 3) || (Cout & 3) || H < 1 || W < 1)
        return ddif::fail(DDIF_ERR_INVALID, "ddif_blockbwd_create: B >= 1, 4 | Cin, 4 | Cout, H, W >= 1 required");
    if ((ks != 1 && ks != 3) || pro < DDIF_BWD_PRO_NONE || pro > DDIF_BWD_PRO_SILU || resample < DDIF_BWD_PLAIN || resample > DDIF_BWD_UP2)
        return ddif::fail(DDIF_ERR_INVALID, "ddif_blockbwd_create_ex: ks must be 1 or 3, pro one of DDIF_BWD_PRO_*, resample one of DDIF_BWD_PLAIN / _DOWN2 / _UP2");
    if (resample != DDIF_BWD_PLAIN && (ks != 3 || pro != DDIF_BWD_PRO_NONE))
        return ddif::fail(DDIF_ERR_INVALID, "ddif_blockbwd_create_ex: Downsample / Upsample are plain 3x3 convs in the reference");
    *out = nullptr;
    int prev = -1;
    (void)hipGetDevice(&prev);
    DDIF_HIPCHK(hipSetDevice(device));
    std::unique_ptr<ddif_blockbwd> h(new ddif_blockbwd());
    ddif::BlockBwd& k = h->b;
    const int up = resample == DDIF_BWD_UP2 ? 2 : 1;
    int rc = ddif::convbwd_init(k.conv, B, Cin, Cout, H * up, W * up, device);
    auto TRY = [&](int e) { if (!rc) rc = e; };
    const size_t n = (size_t)B * H * W * Cin;
    k.H = H;
    k.W = W;
    k.resample = resample;
    k.nchunk = H * W < 32 ? H * W : 32;
    k.ks = ks;
    k.pro = pro;
    ddif::Plan& pl = k.conv.plan;
    if (ks == 1) {
        TRY(pl.dalloc(&k.w3, (size_t)Cout * Cin * 9));
        k.conv.centre_only = true;
    }
    TRY(pl.dalloc(&k.x_nhwc, n));
    TRY(pl.dalloc(&k.mask_nhwc, n));
    TRY(pl.dalloc(&k.dx_nhwc, n));
    TRY(pl.dalloc(&k.S, (size_t)B * 2));
    TRY(pl.dalloc(&k.spart, (size_t)B * k.nchunk * 2));
    TRY(pl.dalloc(&k.cpart, (size_t)B * k.nchunk * Cin * 2));
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (rc) return rc;
    *out = h.release();
    return DDIF_OK;
}

void ddif_blockbwd_destroy(ddif_blockbwd_t h) { delete h; }

int ddif_blockbwd_run(ddif_blockbwd_t h, const float* x, const float* gamma, const float* beta, const float* mask, const float* w, const float* dy, float* dx,
                      float* dgamma, float* dbeta, float* dw, float* db, float* dy_plane_sums, void* stream) {
    if (!h || !x || !w || !dy) return ddif::fail(DDIF_ERR_INVALID, "ddif_blockbwd_run: NULL argument");
    ddif::BlockBwd& k = h->b;
    const bool gn = k.pro == DDIF_BWD_PRO_GN || k.pro == DDIF_BWD_PRO_GN_SILU;
    const bool act = k.pro == DDIF_BWD_PRO_SILU;
    const int silu = k.pro == DDIF_BWD_PRO_GN_SILU;
    if (gn && (!gamma || !beta)) return ddif::fail(DDIF_ERR_INVALID, "ddif_blockbwd_run: GroupNorm prologue without gamma / beta");
    if (!gn && mask) return ddif::fail(DDIF_ERR_INVALID, "ddif_blockbwd_run: a dropout mask needs the GroupNorm prologue (Dropout only follows Swish in the reference)");
    ddif::ConvBwd& c = k.conv;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (prev != c.device) DDIF_HIPCHK(hipSetDevice(c.device));
    hipStream_t s = (hipStream_t)stream;
    const int HW = k.H * k.W, B = c.B, Ci = c.Cin;  // of x
    const size_t n = (size_t)B * HW * Ci;
    const size_t nq = ((size_t)HW * Ci / 4 + 255) / 256;
    const dim3 ew((unsigned)(nq > 64 ? 64 : nq), (unsigned)B);
    const float* m = mask ? k.mask_nhwc : nullptr;
    const bool pre = gn || act;  // something sits between x and the conv's input
    // boundary layout -> NHWC
    const int Ho = k.resample == DDIF_BWD_DOWN2 ? (k.H - 1) / 2 + 1 : c.H, Wo = k.resample == DDIF_BWD_DOWN2 ? (k.W - 1) / 2 + 1 : c.W;  // of dy
    if (k.resample == DDIF_BWD_UP2)
        hipLaunchKernelGGL(ddif::upsample2_nchw_to_nhwc_kernel, ddif::grid_for(n * 4), dim3(256), 0, s, x, B, Ci, k.H, k.W, c.x_nhwc);
    else
        hipLaunchKernelGGL(ddif::bwd_nchw_to_nhwc_kernel, ddif::grid_for(n), dim3(256), ddif::TR_SMEM, s, x, B, Ci, HW, pre ? k.x_nhwc : c.x_nhwc);
    if (mask) hipLaunchKernelGGL(ddif::bwd_nchw_to_nhwc_kernel, ddif::grid_for(n), dim3(256), ddif::TR_SMEM, s, mask, B, Ci, HW, k.mask_nhwc);
    if (k.resample == DDIF_BWD_DOWN2)
        hipLaunchKernelGGL(ddif::zero_stuff_nchw_to_nhwc_kernel, ddif::grid_for((size_t)B * c.H * c.W * c.Cout), dim3(256), 0, s, dy, B, c.Cout, Ho, Wo, c.H, c.W, c.dy_nhwc);
    else
        hipLaunchKernelGGL(ddif::bwd_nchw_to_nhwc_kernel, ddif::grid_for((size_t)B * c.H * c.W * c.Cout), dim3(256), ddif::TR_SMEM, s, dy, B, c.Cout, c.H * c.W, c.dy_nhwc);
    if (gn) {
        // forward recompute: GroupNorm statistics of x, then a = Dropout(SiLU(GroupNorm(x))) -- the conv's input, needed by wgrad
        hipLaunchKernelGGL(ddif::gnb_stats_kernel, dim3(k.nchunk, B), dim3(256), 2 * 256 * sizeof(double), s, (const float*)k.x_nhwc, (size_t)HW * Ci, k.nchunk, k.spart);
        hipLaunchKernelGGL(ddif::gnb_act_kernel, ew, dim3(256), 0, s, (const float*)k.x_nhwc, (const double*)k.spart, k.nchunk, gamma, beta, m, HW, Ci, silu, c.x_nhwc);
    } else if (act) {
        hipLaunchKernelGGL(ddif::silu_fwd_kernel, ddif::grid_for(n), dim3(256), 0, s, (const float*)k.x_nhwc, n, c.x_nhwc);
    }
    // conv backward: dA (needed whenever anything in front of the conv wants a gradient), dW, db
    const bool want_da = gn || dx;
    const size_t nw1 = (size_t)c.Cout * Ci;
    if (k.ks == 1) {
        hipLaunchKernelGGL(ddif::embed_1x1_kernel, ddif::grid_for(nw1 * 9), dim3(256), 0, s, w, nw1, k.w3);
        ddif::convbwd_core(c, s, k.w3, want_da, dw, db);  // dW comes out as (Cout, Cin): only the centre tap is contracted
    } else {
        ddif::convbwd_core(c, s, w, want_da, dw, db);
    }
    if (dy_plane_sums) hipLaunchKernelGGL(ddif::plane_sum_nchw_kernel, dim3(B * c.Cout), dim3(256), 256 * sizeof(double), s, dy, Ho * Wo, dy_plane_sums);
    const float* da = c.dx.p;
    if (gn) {
        // GroupNorm (+ SiLU + dropout) backward
        hipLaunchKernelGGL(ddif::gnb_bwd_partial_kernel<0>, dim3(k.nchunk, B), dim3(256), 256 * 8 * sizeof(double), s, (const float*)k.x_nhwc, (const float*)nullptr, 0, da, m,
                           (const double*)k.spart, k.nchunk, (const double*)nullptr, 0, gamma, beta, HW, Ci, k.nchunk, silu, k.cpart, (double*)nullptr);
        hipLaunchKernelGGL(ddif::gnb_bwd_reduce_kernel, dim3((Ci + 31) / 32 + B), dim3(ddif::GNB_RED_NT), 2 * ddif::GNB_RED_NT * sizeof(double), s, (const double*)k.cpart, gamma, B, k.nchunk, Ci, dgamma, dbeta,
                           k.S);
        if (dx) {
            hipLaunchKernelGGL(ddif::gnb_bwd_dx_kernel<0>, ew, dim3(256), 0, s, (const float*)k.x_nhwc, (const float*)nullptr, 0, da, m, (const double*)k.spart, k.nchunk,
                               (const double*)nullptr, 0, gamma, beta, (const float*)k.S, (const float*)nullptr, HW, Ci, silu, k.dx_nhwc, (float*)nullptr, (const double*)nullptr, 0);
            hipLaunchKernelGGL(ddif::bwd_nhwc_to_nchw_kernel, ddif::grid_for(n), dim3(256), ddif::TR_SMEM, s, (const float*)k.dx_nhwc, B, Ci, HW, dx);
        }
    } else if (dx) {
        if (act) {
            hipLaunchKernelGGL(ddif::silu_bwd_kernel, ddif::grid_for(n), dim3(256), 0, s, (const float*)k.x_nhwc, da, n, k.dx_nhwc);
            hipLaunchKernelGGL(ddif::bwd_nhwc_to_nchw_kernel, ddif::grid_for(n), dim3(256), ddif::TR_SMEM, s, (const float*)k.dx_nhwc, B, Ci, HW, dx);
        } else if (k.resample == DDIF_BWD_UP2) {
            hipLaunchKernelGGL(ddif::sumpool2_nhwc_to_nchw_kernel, ddif::grid_for(n), dim3(256), 0, s, da, B, Ci, k.H, k.W, dx);
        } else {
            hipLaunchKernelGGL(ddif::bwd_nhwc_to_nchw_kernel, ddif::grid_for(n), dim3(256), ddif::TR_SMEM, s, da, B, Ci, HW, dx);
        }
    }
    int rc = DDIF_OK;
    if (hipGetLastError() != hipSuccess) rc = ddif::fail(DDIF_ERR_HIP, "ddif_blockbwd_run: kernel launch failed");
    if (prev >= 0 && prev != c.device) (void)hipSetDevice(prev);
    return rc;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------- conv forward (training graph)
namespace ddif {
struct ConvFwd {
    int device = 0, B = 0, Cin = 0, Cout = 0, H = 0, W = 0, ks = 3, stride = 1, up2 = 0;
    Net net;
    Plan plan;
    PackedConv pc;
    float *x_nhwc = nullptr, *wpack = nullptr, *wpack_x3 = nullptr, *bias = nullptr;
    Tensor y;
    std::vector<Op> prog;
    int n_chunks = 0, nb_pad = 0;
};
}  // namespace ddif
struct ddif_convfwd {
    ddif::ConvFwd c;
};

extern "C" {

int ddif_convfwd_create(ddif_convfwd_t* out, int B, int Cin, int Cout, int H, int W, int ks, int stride, int up2, int device) {
    if (!out || B < 1 || Cin < 4 || Cout < 4 || (Cin & 3) || (Cout & 3) || H < 1 || W < 1 || (ks != 1 && ks != 3) || (stride != 1 && stride != 2) || (up2 && stride != 1))
        return ddif::fail(DDIF_ERR_INVALID, "ddif_convfwd_create: B >= 1, 4 | Cin, 4 | Cout, ks in {1,3}, stride in {1,2}, up2 only with stride 1");
    if (ks == 1 && (stride != 1 || up2)) return ddif::fail(DDIF_ERR_INVALID, "ddif_convfwd_create: 1x1 convs are plain in the reference");
    *out = nullptr;
    int prev = -1;
    (void)hipGetDevice(&prev);
    DDIF_HIPCHK(hipSetDevice(device));
    std::unique_ptr<ddif_convfwd> h(new ddif_convfwd());
    ddif::ConvFwd& c = h->c;
    c.device = device;
    c.B = B; c.Cin = Cin; c.Cout = Cout; c.H = H; c.W = W; c.ks = ks; c.stride = stride; c.up2 = up2;
    c.plan.net = &c.net;
    c.plan.B = B;
    c.plan.H = H;
    c.plan.W = W;
    int rc = 0;
    auto TRY = [&](int e) { if (!rc) rc = e; };
    TRY(c.plan.dalloc(&c.plan.zeros, (size_t)1024));
    if (!rc && hipMemset(c.plan.zeros, 0, 1024 * sizeof(float)) != hipSuccess) rc = ddif::fail(DDIF_ERR_HIP, "ddif_convfwd_create: hipMemset failed");
    TRY(c.plan.dalloc(&c.x_nhwc, (size_t)B * H * W * Cin));
    TRY(c.plan.dalloc(&c.bias, (size_t)((Cout + 31) & ~31)));
    c.n_chunks = (Cin + 15) / 16;
    c.nb_pad = (((Cout + 31) / 32) + 3) & ~3;
    TRY(c.plan.dalloc(&c.wpack, (size_t)c.nb_pad * c.n_chunks * 9 * 2 * 256));
    TRY(c.plan.dalloc(&c.wpack_x3, (size_t)c.nb_pad * c.n_chunks * 9 * 3 * 256));
    c.pc.w = c.wpack;
    c.pc.w_x3 = c.wpack_x3;
    c.pc.bias = c.bias;
    c.pc.cin = Cin;
    c.pc.cout = Cout;
    c.pc.ks = 3;  // 1x1 weights ride in the centre tap
    c.pc.ck = 16;
    c.pc.n_chunks = c.n_chunks;
    if (!rc) {
        ddif::ConvSpec s;
        s.pc = &c.pc;
        s.in0.p = c.x_nhwc;
        s.in0.C = Cin;
        s.in0.H = H;
        s.in0.W = W;
        s.stride = stride;
        s.ups = up2 ? 1 : 0;
        s.use_bias = true;
        s.exact = !ddif::train_x3();  // bf16x3 split products like the inference path; DDIF_TRAIN_X3=0: exact-fp32 MFMA
        s.name = "train.conv";
        TRY(c.plan.add_conv(c.prog, s, &c.y));
    }
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (rc) return rc;
    *out = h.release();
    return DDIF_OK;
}

void ddif_convfwd_destroy(ddif_convfwd_t h) { delete h; }

int ddif_convfwd_run(ddif_convfwd_t h, const float* x, const float* w, const float* bias, float* y, void* stream) {
    if (!h || !x || !w || !y) return ddif::fail(DDIF_ERR_INVALID, "ddif_convfwd_run: NULL argument");
    ddif::ConvFwd& c = h->c;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (prev != c.device) DDIF_HIPCHK(hipSetDevice(c.device));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ddif::bwd_nchw_to_nhwc_kernel, ddif::grid_for((size_t)c.B * c.H * c.W * c.Cin), dim3(256), ddif::TR_SMEM, s, x, c.B, c.Cin, c.H * c.W, c.x_nhwc);
    const size_t nw = (size_t)c.nb_pad * c.n_chunks * 9 * 2 * 256;
    hipLaunchKernelGGL(ddif::pack_fwd_weights_kernel, ddif::grid_for(nw), dim3(256), 0, s, w, c.Cout, c.Cin, c.ks, c.n_chunks, c.nb_pad, c.wpack);
    hipLaunchKernelGGL(ddif::pack_weights_x3_kernel, ddif::grid_for(nw), dim3(256), 0, s, w, c.Cout, c.Cin, c.ks, 0, c.n_chunks, c.nb_pad, c.wpack_x3);
    int rc = DDIF_OK;
    if (bias) {
        if (hipMemcpyAsync(c.bias, bias, (size_t)c.Cout * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) rc = ddif::fail(DDIF_ERR_HIP, "ddif_convfwd_run: bias copy failed");
    } else if (hipMemsetAsync(c.bias, 0, (size_t)c.Cout * sizeof(float), s) != hipSuccess) {
        rc = ddif::fail(DDIF_ERR_HIP, "ddif_convfwd_run: bias clear failed");
    }
    ddif::StepCtx ctx;
    for (auto& op : c.prog) op.run(s, ctx);
    hipLaunchKernelGGL(ddif::bwd_nhwc_to_nchw_kernel, ddif::grid_for((size_t)c.B * c.y.H * c.y.W * c.Cout), dim3(256), ddif::TR_SMEM, s, (const float*)c.y.p, c.B, c.Cout, c.y.H * c.y.W, y);
    if (!rc && hipGetLastError() != hipSuccess) rc = ddif::fail(DDIF_ERR_HIP, "ddif_convfwd_run: kernel launch failed");
    if (prev >= 0 && prev != c.device) (void)hipSetDevice(prev);
    return rc;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------- launchers for the native training step (ddif_train.cpp)
namespace ddif {
namespace tk {
void silu_fwd(hipStream_t s, const float* x, size_t n, float* y) { hipLaunchKernelGGL(silu_fwd_kernel, grid_for(n), dim3(256), 0, s, x, n, y); }
void silu_bwd(hipStream_t s, const float* x, const float* da, size_t n, float* dx) { hipLaunchKernelGGL(silu_bwd_kernel, grid_for(n), dim3(256), 0, s, x, da, n, dx); }
WgradGeom wgrad_geom(int B, int Cin, int Cout, int H, int W, bool centre) {  // as convbwd_init
    WgradGeom g;
    g.n_co = (Cout + 31) / 32;
    g.n_ci = (Cin + 31) / 32;
    // rows per band: the largest of {16, 8, 4, 2, 1} (<= H) whose two tiles fit DDIF_WGRAD_SMEM_KB (default 72 KB: two workgroups per CU, so one's
    // band loads overlap the other's MFMAs; the low-resolution levels then stage a whole 8x8 sample or half a 16x16 one per band)
    static const size_t lim = [] { const char* e = getenv("DDIF_WGRAD_SMEM_KB"); return (size_t)(e ? atoi(e) : 72) * 1024; }();
    g.pf = 0;
    g.centre = centre ? 1 : 0;
    const int halo = centre ? 0 : 1;
    for (g.rb = H < 16 ? H : 16; g.rb >= 1; --g.rb) {  // first choice: the largest band whose float4 items fit the kernel's prefetch registers (WG_PF per thread)
        if (((size_t)g.rb * W + (size_t)(g.rb + 2 * halo) * (W + 2 * halo)) * 8 <= (size_t)(centre ? WG_PF_CENTRE : WG_PF) * 256) {
            g.pf = 1;
            break;
        }
    }
    if (g.pf) {  // among the bands that fit: fewest rows processed (padding rows of the last band count) + half a row of per-band overhead
        int best = g.rb;
        double best_cost = 1e30;
        for (int r = g.rb; r >= 1; --r) {
            const int nb = (H + r - 1) / r;
            const double cost = (double)nb * r + 0.5 * nb;
            if (cost < best_cost - 1e-9) {
                best_cost = cost;
                best = r;
            }
        }
        g.rb = best;
    }
    if (!g.pf) g.centre = 0;  // the halo-free form exists for the prefetching kernel only
    if (!g.pf)
        for (g.rb = 16; g.rb >= 1; g.rb >>= 1) {
            if (g.rb > H && g.rb > 1) continue;
            g.smem = ((size_t)g.rb * W * 32 + (size_t)(g.rb + 2) * (W + 2) * 32 + 4096) * sizeof(float);
            if (g.smem <= lim || (g.rb == 1 && g.smem <= 150 * 1024)) break;
        }
    if (g.rb >= 1) {
        const int hl = g.centre ? 0 : 1;
        g.smem = ((size_t)g.rb * W * 32 + (size_t)(g.rb + 2 * hl) * (W + 2 * hl) * 32 + 4096) * sizeof(float);
    }
    if (g.rb < 1) g.rb = 0;  // W too wide (caller checks)
    // the bf16x3 kernel (kernels_bwd.h conv3x3_wgrad_x3_kernel; DDIF_WGRAD_X3=0: the fp32 kernel everywhere): 8 | W, W <= 128; the largest band of <= 256 staging
    // items (RB rows of dY + RB new rows of X) whose transposed bf16 tiles fit two workgroups per CU (failing that, one)
    static const bool x3_env = [] { const char* e = getenv("DDIF_WGRAD_X3"); return !e || atoi(e) != 0; }();
    if (x3_env && W % 8 == 0 && W <= 128) {
        auto stride = [](int n) { return n + (((n / 8) % 2 == 0) ? 8 : 0); };  // an odd number of 16-byte slots
        const int segs = W / 8, hl = centre ? 0 : 1, xw = centre ? W : W + 16;
        // the largest band of <= 256 staging items (RB x W / 8 <= 16) that fits the LDS: measured (tools/mbench_wgrad.cpp, gpurun_out/r05_t) full 256-item bands with ONE
        // workgroup per CU beat smaller bands with two at every level (64 x 64: 25.9 vs 26.7 us, 32 x 32: 25.9 vs 30.1, 16 x 16: 13.2 vs 16.2)
        for (int rb = H < 16 ? H : 16; rb >= 1 && !g.x3; --rb) {
            if (rb * segs > 16) continue;
            const int ys = stride(rb * W), xs = stride((rb + 2 * hl) * xw);
            size_t smem = (size_t)2 * 96 * (ys + xs);
            if (smem < 16384) smem = 16384;  // the epilogue's reduction scratch aliases the tiles
            if (smem > 150 * 1024) continue;
            g.x3 = 1;
            g.rb = rb;
            g.xw = xw;
            g.ys = ys;
            g.xs = xs;
            g.smem = smem;
            g.centre = centre ? 1 : 0;
        }
    }
    const int bands = B * ((H + (g.rb ? g.rb : 1) - 1) / (g.rb ? g.rb : 1));
    // workgroups per launch: two per CU -- one where the bf16x3 kernel's tiles leave room for one only (a second wave of workgroups costs 30 %: same measurement)
    int want = ((g.x3 && g.smem > 76 * 1024) ? 256 : 512) / (g.n_co * g.n_ci);
    if (want < 1) want = 1;
    if (want > bands) want = bands;
    if (want > 512) want = 512;
    g.nsplit = want;
    g.partial_floats = (size_t)g.nsplit * g.n_co * g.n_ci * 9 * 1024;
    g.nbchunk = (int)std::min<size_t>(512, std::max<size_t>(1, (size_t)B * H * W / 16));
    return g;
}
int wgrad_prepare() {
    DDIF_HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    DDIF_HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    DDIF_HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_kernel<1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    DDIF_HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_x3_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    DDIF_HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_x3_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    return 0;
}
void wgrad(hipStream_t s, const float* x, const float* dy, int B, int H, int W, int Cin, int Cout, const WgradGeom& g, bool centre, float* partial, float* dw, float* bpart,
           float* db) {
    WgradArgs a{};
    a.x = x;
    a.dy = dy;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.n_ci = g.n_ci;
    a.rb = g.rb;
    a.bands_y = (H + g.rb - 1) / g.rb;
    a.partial = partial;
    a.centre_only = centre ? 1 : 0;
    a.wshift = -1;
    for (int k = 0; k < 16; ++k)
        if ((1 << k) == W) a.wshift = k;
    a.bpartial = (db && bpart) ? bpart : nullptr;  // bias gradient fused: per-split column sums from the kernel, finished by the reduce kernel
    static const bool dump = getenv("DDIF_WGRAD_DUMP") != nullptr;  // development aid: geometry of every launch, in order (match against a kernel trace)
    if (dump) fprintf(stderr, "[wgrad] B=%d H=%d W=%d Cin=%d Cout=%d centre=%d rb=%d nsplit=%d blocks=%d pf=%d x3=%d smem=%zu\n", B, H, W, Cin, Cout, (int)centre, g.rb, g.nsplit, g.n_co * g.n_ci, g.pf, g.x3, g.smem);
    if (g.x3) {
        WgradX3Geom gm{g.rb, g.xw, g.ys, g.xs};
        const dim3 grid(g.n_co * g.n_ci, g.nsplit);
        if (centre) hipLaunchKernelGGL(conv3x3_wgrad_x3_kernel<1>, grid, dim3(256), g.smem, s, a, gm);
        else hipLaunchKernelGGL(conv3x3_wgrad_x3_kernel<0>, grid, dim3(256), g.smem, s, a, gm);
    } else if (g.pf && g.centre && centre) hipLaunchKernelGGL((conv3x3_wgrad_kernel<1, 1>), dim3(g.n_co * g.n_ci, g.nsplit), dim3(256), g.smem, s, a);
    else if (g.pf) hipLaunchKernelGGL(conv3x3_wgrad_kernel<1>, dim3(g.n_co * g.n_ci, g.nsplit), dim3(256), g.smem, s, a);
    else hipLaunchKernelGGL(conv3x3_wgrad_kernel<0>, dim3(g.n_co * g.n_ci, g.nsplit), dim3(256), g.smem, s, a);
    // one workgroup per row of 32 input channels of the partial layout (capped), + one for the bias
    const int nblk = g.n_co * g.n_ci;
    const int rows = centre ? nblk * 32 : nblk * 9 * 32;
    const unsigned rgrid = (unsigned)(rows < 2048 ? rows : 2048) + (a.bpartial ? 1u : 0u);
    if (centre)
        hipLaunchKernelGGL(wgrad_reduce_centre_kernel, dim3(rgrid), dim3(256), 256 * sizeof(float), s, (const float*)partial, g.nsplit, nblk, g.n_ci, Cout, Cin, dw,
                           (const float*)a.bpartial, db);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rgrid), dim3(256), 256 * sizeof(float), s, (const float*)partial, g.nsplit, nblk, g.n_ci, Cout, Cin, dw,
                           (const float*)a.bpartial, db);
}
void bias_grad(hipStream_t s, const float* dy, size_t npix, int Cout, int nbchunk, float* bpart, float* db) {
    hipLaunchKernelGGL(bias_grad_partial_kernel, dim3(nbchunk), dim3(256), 256 * sizeof(float), s, dy, npix, Cout, nbchunk, bpart);
    hipLaunchKernelGGL(bias_grad_reduce_kernel, dim3(Cout), dim3(64), 64 * sizeof(float), s, (const float*)bpart, nbchunk, Cout, db);
}
void gn_stats(hipStream_t s, const float* x, int B, size_t per_sample, int nchunk, double* spart) {
    hipLaunchKernelGGL(gnb_stats_kernel, dim3(nchunk, B), dim3(256), 2 * 256 * sizeof(double), s, x, per_sample, nchunk, spart);
}
static inline dim3 ew_grid2(int B, int HW, int C) {
    const size_t nq = ((size_t)HW * C / 4 + 255) / 256;
    return dim3((unsigned)(nq > 64 ? 64 : (nq < 1 ? 1 : nq)), (unsigned)B);
}
// `st` [B][np][2]: the partials of x -- gn_stats' output or, in the training step, what the forward producer of x left behind (no second read of x)
void gn_act(hipStream_t s, const float* x, const double* st, int np, const float* gamma, const float* beta, const float* mask, int B, int HW, int C, int silu, float* out) {
    hipLaunchKernelGGL(gnb_act_kernel, ew_grid2(B, HW, C), dim3(256), 0, s, x, st, np, gamma, beta, mask, HW, C, silu, out);
}
// gpart != nullptr (the training step): two launches -- the partials (with their gamma-weighted chunk sums) and dx, which forms the per-sample sums from those; the
// dgamma / dbeta reduction of `cpart` is the caller's (one launch for all GroupNorms of the iteration, gn_bwd_reduce_all); dgamma / dbeta / S are not touched here
void gn_bwd(hipStream_t s, const float* x, const float* da, const float* mask, const double* st, int np, const float* gamma, const float* beta, int B, int HW, int C,
            int nchunk, int silu, double* cpart, float* S, float* dgamma, float* dbeta, const float* res, float* dx, double* gpart) {
    hipLaunchKernelGGL(gnb_bwd_partial_kernel<0>, dim3(nchunk, B), dim3(256), 256 * 8 * sizeof(double), s, x, (const float*)nullptr, 0, da, mask, st, np, (const double*)nullptr, 0,
                       gamma, beta, HW, C, nchunk, silu, cpart, gpart);
    // (plane sums + finalize as ONE single-workgroup launch measured 46 us against 10 + 15 us for the pair; as (C / 32 + B) workgroups of 1024 threads it is one launch)
    if (!gpart)
        hipLaunchKernelGGL(gnb_bwd_reduce_kernel, dim3((C + 31) / 32 + B), dim3(GNB_RED_NT), 2 * GNB_RED_NT * sizeof(double), s, (const double*)cpart, gamma, B, nchunk, C, dgamma, dbeta, S);
    if (dx)
        hipLaunchKernelGGL(gnb_bwd_dx_kernel<0>, ew_grid2(B, HW, C), dim3(256), 0, s, x, (const float*)nullptr, 0, da, mask, st, np, (const double*)nullptr, 0, gamma, beta,
                           (const float*)S, res, HW, C, silu, dx, (float*)nullptr, (const double*)gpart, nchunk);
}
void gn_bwd_reduce_all(hipStream_t s, const GnRedRec* recs_dev, int nrec, int nblocks, int B) {
    hipLaunchKernelGGL(gnb_bwd_reduce_all_kernel, dim3(nblocks), dim3(1024), 2 * 1024 * sizeof(double), s, recs_dev, nrec, B);
}
// GroupNorm (no SiLU, no mask) over the never-materialised concatenation cat[x0 (c0 channels), x1 (c1)]: statistics from both producers' partials, the gradient
// written to the two tensors' own gradients
void gn_bwd_cat(hipStream_t s, const float* x0, int c0, const float* x1, int c1, const double* st0, int np0, const double* st1, int np1, const float* da, const float* gamma,
                const float* beta, int B, int HW, int nchunk, double* cpart, float* S, float* dgamma, float* dbeta, float* dx0, float* dx1, double* gpart) {
    const int C = c0 + c1;
    hipLaunchKernelGGL(gnb_bwd_partial_kernel<1>, dim3(nchunk, B), dim3(256), 256 * 8 * sizeof(double), s, x0, x1, c0, da, (const float*)nullptr, st0, np0, st1, np1, gamma, beta, HW,
                       C, nchunk, 0, cpart, gpart);
    if (!gpart)
        hipLaunchKernelGGL(gnb_bwd_reduce_kernel, dim3((C + 31) / 32 + B), dim3(GNB_RED_NT), 2 * GNB_RED_NT * sizeof(double), s, (const double*)cpart, gamma, B, nchunk, C, dgamma,
                           dbeta, S);
    hipLaunchKernelGGL(gnb_bwd_dx_kernel<1>, ew_grid2(B, HW, C), dim3(256), 0, s, x0, x1, c0, da, (const float*)nullptr, st0, np0, st1, np1, gamma, beta, (const float*)S,
                       (const float*)nullptr, HW, C, 0, dx0, dx1, (const double*)gpart, nchunk);
}
}  // namespace tk
}  // namespace ddif
