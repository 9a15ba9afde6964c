// Plan: per-(B,H,W) workspaces, the cond-only caches and the launch program of one denoising step.
#pragma once
#include <string>
#include <memory>
#include "ddif_net.h"

namespace ddif {

struct ConvArgs;

struct Tensor {
    float* p = nullptr;
    int C = 0, H = 0, W = 0;
    double* st = nullptr;  // GroupNorm partials [B][np][2] written by the producer
    int np = 0;
};

struct StepCtx {
    const float* x = nullptr;   // network input x      (NHWC)
    const float* sc = nullptr;  // self-conditioning    (NHWC)
    const float* tb = nullptr;  // time-bias row(s)
    int tb_stride = 0;          // floats between the rows of consecutive samples (0: one row for the batch)
    const int* step_ptr = nullptr;  // samplers: device step counter selecting the time-bias row (row stride tb_rowstride)
    int tb_rowstride = 0;
    // DDPM / DDIM loops: the sampler update runs in the final conv's epilogue (kernels_conv.h EPI_SAMP) when the plan carries that variant
    float* samp_out = nullptr;      // x_{t-1} (null: plain network output)
    const void* samp_run = nullptr; // device SamplerRun
    int* step_next = nullptr;       // the counter the NEXT step reads
    int samp_kind = 0;
};

struct Op {
    std::function<void(hipStream_t, const StepCtx&)> run;
    double flop = 0, bytes = 0;
    double mfma_w = 16;  // matrix-pipe issue weight of the op's flops in units of the dense 16-bit MFMA rate: 3 = f16x2, 6 = bf16x3, 16 = exact fp32 MFMA / VALU
    bool timed = false;  // member of the dominant kernel class (3x3 implicit-GEMM convs at the high-resolution levels)
    bool side = false;   // train-mode cond-only program: a decoder-only op -- issued on the plan's side stream, joined in front of the first decoder block
    int cls = 5;         // profiling class: 0 conv3x3 (> 256 px / sample), 1 conv1x1 (> 256 px), 2 low-resolution levels, 3 attention, 4 softmax statistics, 5 other
    const char* name = "";
    std::string label;   // layer + shape, for the DDIF_OP_TIMING dump
};

typedef void (*ConvKernelFn)(ConvArgs);
struct ConvVariant {
    ConvKernelFn fn = nullptr;
    size_t smem = 0;
    int th = 0, tw = 0, nt = 0, nthr = 256;
    bool x3 = false;  // bf16x3 instantiation: wants PackedConv::w_x3
    bool f16 = false; // f16x2 instantiation: wants PackedConv::w_f16 (x3 is set as well: split-operand path)
    bool b1 = false;  // bf16x1 instantiation (the throughput variant): wants PackedConv::w_b1 (x3 is set as well)
    int wg_cap = 2;   // persistent workgroups per CU (upper bound; LDS may allow fewer)
    bool lr = false;  // low-resolution kernel (kernels_lr.h): smem is the whole requirement, nothing is added per launch
    bool wr = false;  // resident-weights instantiation (kernels_conv.h MATH = 5): ONE 32-channel stage per work item -- the launch passes n_chunks = 1, CK = 32; reads w_f16
    const char* name = "";
};
// operand format of a split-operand conv: what get_conv_variant / get_lr_variant instantiate and which weight pack the launch reads
enum { MATH_BF16X3 = 0, MATH_F16X2 = 1, MATH_BF16X1 = 2 };
ConvVariant get_conv_variant(int ks, int stride, int ups, int ck, int pro, int cfg, int vec, int epi, int math = MATH_BF16X3);
ConvVariant get_lr_variant(int ks, int mb, int pro, int epi, int math = MATH_BF16X3, bool rows = false);  // ddif_lr.cpp (rows: the whole-width staging, kernels_lr.h ROWS)
ConvVariant get_conv_variant_k1(int stride, int ups, int ck, int pro, int cfg, int vec, int epi);   // ddif_conv_k1.cpp
ConvVariant get_conv_variant_k3(int stride, int ups, int ck, int pro, int cfg, int vec);            // ddif_conv_k3.cpp
ConvVariant get_conv_variant_k3e(int stride, int ups, int ck, int pro, int cfg, int vec, int epi);  // ddif_conv_k3e.cpp
ConvVariant get_xf_variant(int stride, int pro, int cfg, int vec, int epi, int nbx);      // EPI_XF instantiations (f16x2 3x3 convs with 32 couts; ddif_xf.cpp)
// ddif_set_math_mode (include/ddif.h): 0 = fp32-class split products (default), 1 = the bf16 throughput variant, for plans built afterwards
extern int g_math_mode;
// ddif_set_f16_raw (include/ddif.h): 1 (default) = convs without a GroupNorm prologue may run f16x2 on their raw input, watched by the plan's range flag;
// 0 = they stay on bf16x3 (full fp32 range).  Snapshotted, like the math mode, when a plan is created.
extern int g_f16_raw;
// fused linear-attention block (kernels_lafuse.h, ddif_la.cpp)
struct LaFuseArgs;
bool lafuse_supported(int H, int fea, int dout);
int lafuse_strip(int H, int nw = 8);
int lafuse_launch(const LaFuseArgs& a, int grid, hipStream_t s, bool prepare_only, int nw = 8);
bool lafuse8_supported(int H, int W, int c0, int c1, int dout);  // the 8 x 8 level (kernels_lafuse8.h)
int lafuse8_launch(const LaFuseArgs& a, int grid, hipStream_t s, bool prepare_only);
// launchers of kernels that live in other translation units (every non-template kernel header is compiled into exactly one object)
namespace tk {
// kernels_train.h  (ddif_train.cpp)
void film_apply(hipStream_t s, const float* xc, const float* film, int B, int HW, int C, float* out, double* st_out, int chunks);
int film_chunks(int HW, int C);
size_t linattn_part_floats(int B, int H, int W, int C, int d);
void linattn_fwd(hipStream_t s, const float* q, const float* kv, int B, int heads, int d, int H, int W, float* out, int ld_o, float* ctx, float* part);
void linattn_ctx(hipStream_t s, const float* kv, int B, int heads, int d, int H, int W, float* ctx, float* part);
void linattn_apply(hipStream_t s, const float* q, const float* ctx, int B, int heads, int d, int H, int W, float* out, int ld_o);
void linattn_bwd(hipStream_t s, const float* q, const float* kv, const float* dout, int ld_g, const float* ctx, int B, int heads, int d, int H, int W, float* dq, float* dkv,
                 float* dctx, float* part);
void linattn_bwd_q(hipStream_t s, const float* q, const float* dout, int ld_g, const float* ctx, int B, int heads, int d, int H, int W, float* dq, float* dctx, float* part);
void linattn_bwd_k(hipStream_t s, const float* kv, const float* dctx, int B, int heads, int d, int H, int W, float* dkv);
int linattn_prepare();
// kernels_bwd.h  (ddif_bwd.cpp)
void silu_fwd(hipStream_t s, const float* x, size_t n, float* y);
void silu_bwd(hipStream_t s, const float* x, const float* da, size_t n, float* dx);
struct WgradGeom { int n_co = 0, n_ci = 0, nsplit = 0, rb = 0, nbchunk = 0, pf = 0, centre = 0; size_t smem = 0, partial_floats = 0;
                   int x3 = 0, xw = 0, ys = 0, xs = 0; };  // x3: the bf16x3 kernel (conv3x3_wgrad_x3_kernel) with its tile geometry
WgradGeom wgrad_geom(int B, int Cin, int Cout, int H, int W, bool centre = false);
int wgrad_prepare();
void wgrad(hipStream_t s, const float* x, const float* dy, int B, int H, int W, int Cin, int Cout, const WgradGeom& g, bool centre, float* partial, float* dw,
           float* bpart = nullptr, float* db = nullptr);  // db: the bias gradient too (bpart: >= nsplit * n_co * 32 floats of scratch)
void bias_grad(hipStream_t s, const float* dy, size_t npix, int Cout, int nbchunk, float* bpart, float* db);
void gn_stats(hipStream_t s, const float* x, int B, size_t per_sample, int nchunk, double* spart);
void gn_act(hipStream_t s, const float* x, const double* st, int np, const float* gamma, const float* beta, const float* mask, int B, int HW, int C, int silu, float* out);
void gn_bwd(hipStream_t s, const float* x, const float* da, const float* mask, const double* st, int np, const float* gamma, const float* beta, int B, int HW, int C,
            int nchunk, int silu, double* cpart, float* S, float* dgamma, float* dbeta, const float* res, float* dx, double* gpart = nullptr);
void gn_bwd_reduce_all(hipStream_t s, const GnRedRec* recs_dev, int nrec, int nblocks, int B);
void gn_bwd_cat(hipStream_t s, const float* x0, int c0, const float* x1, int c1, const double* st0, int np0, const double* st1, int np1, const float* da, const float* gamma,
                const float* beta, int B, int HW, int nchunk, double* cpart, float* S, float* dgamma, float* dbeta, float* dx0, float* dx1, double* gpart = nullptr);
// kernels_bwd_ops.h  (ddif_bwd_ops.cpp)
void linear_bwd(hipStream_t s, const float* x, const float* w, const float* dy, int B, int nin, int nout, float* dx, float* dw, float* db);
void l1_fwd(hipStream_t s, const float* pred, const float* target, size_t n, float* out);
void l1_bwd(hipStream_t s, const float* pred, const float* target, size_t n, float upstream, float* dpred);
// kernels_misc.h  (ddif_plan.cpp)
void dw3x3_plain(hipStream_t s, const float* in, int C, int B, int H, int W, const float* w9c, float* out, bool flip = false);  // depthwise conv, weights [9][C]
}  // namespace tk
struct AttnBlockArgs;
int attn_block_prepare();                                                        // ddif_lr.cpp (kernels_attn.h)
void attn_block_launch(const AttnBlockArgs& a, int grid, hipStream_t s);
int attn_block_split();  // workgroups per sample (2 since round 6; DDIF_ATTN_SPLIT=1: one)

// round 6: ask add_conv() to fold the NEXT block's CondInjection (x_conv + FiLM) into this conv's epilogue (kernels_conv.h EPI_XF); `done` and `out` come back
struct XfReq {
    const PackedConv* pc = nullptr;  // the x_conv (bias, shapes)
    const float* w = nullptr;        // its register-GEMM pack (Net::vec "<block>.cond_inj.x_conv.xf")
    const float* film = nullptr;     // scale | shift of that block, [B, H, W, 2 cout]
    Tensor out;                      // y, with statistics partials
    bool done = false;
};

struct ConvSpec {
    const PackedConv* pc = nullptr;
    Tensor in0, in1;           // in1.p == nullptr: single source
    bool dyn_input = false;    // stem: sources come from StepCtx (sc, x)
    int stride = 1, ups = 0, pro = PRO_NONE;
    const float* gamma = nullptr;
    const float* beta = nullptr;
    bool use_bias = true;
    int tb_off = -1;
    const float* res = nullptr;
    const float* film = nullptr;
    const float* w_override = nullptr;  // per-sample weights (packed) instead of pc->w
    long long w_bstride = 0;
    const float* cs_mx = nullptr;       // PRO_COLSM statistics of in0
    const float* cs_sm = nullptr;
    const float* dw_w = nullptr;        // PRO_GN_DW: depthwise weights [9][C]
    float* out_xn = nullptr;            // PRO_GN_DW: normalised input written out
    float* cso_mx = nullptr;            // low-resolution kernel: column-softmax statistics of the OUTPUT ([B][W][Cout]); the image
    float* cso_sm = nullptr;            //   must fit one tile vertically (caller checks with lr_colstats_ok)
    bool silu = false;
    bool stats = false;
    bool exact = false;                 // force the exact-fp32 MFMA instantiation of kernels_conv.h (gradient convs; pc->w only)
    bool samp = false;                  // the network's final conv: also look up the sampler-epilogue variant (EPI_SAMP)
    XfReq* xf = nullptr;                // fold request (honoured only where an EPI_XF instantiation of the chosen tiling exists)
    const char* name = "conv";
};

struct Plan {
    Net* net = nullptr;
    unsigned long long generation = 0;  // Net::generation at build time: entry points refuse a plan that outlived a re-commit
    int B = 0, H = 0, W = 0, C = 0, P = 0, CC = 0;
    std::vector<void*> allocs;
    std::vector<Op> pre, step;
    bool cond_set = false;
    // train mode (ddif_plan_create_train): Dropout after every ResnetBlock's SiLU and DropPath on every decoder FFN
    bool train_mode = false;
    struct DropSite { float* mask; int C, H, W; };
    std::vector<DropSite> drop_sites;      // NHWC masks (0 or 1/(1-p)), one per Block with dropout, in execution order
    std::vector<float*> path_sites;        // per-sample DropPath scales [B], one per FastAttnCondInjection, in execution order
    float* mask_recs = nullptr;            // device table of every site for the one-launch mask generator (kernels_misc.h MaskRec), built on first use
    int n_mask_recs = 0;
    unsigned long long mask_blocks = 0;
    size_t bytes_allocated = 0;

    // fixed buffers
    float* cond_nchw = nullptr;       // not owned (borrowed during set_cond)
    Tensor lms;                       // cond[:, :C] NHWC
    std::vector<Tensor> cenc, cdec;   // resized cond per level
    std::vector<int> LH, LW;          // level sizes
    Tensor x_in, sc_in, net_out;      // NHWC staging of the boundary tensors
    float* zeros = nullptr;     // 4 KiB of 0.0f: stands in for absent conv biases / time-bias rows
    float* img[2] = {nullptr, nullptr};
    float* mbuf[3] = {nullptr, nullptr, nullptr};
    float* io_nchw = nullptr;         // scratch (B,C,H,W)
    float* tvals = nullptr;           // device time values
    float* tb = nullptr;              // time-bias table
    int tb_rows = 0;
    float* small = nullptr;           // device scratch for per-sample coefficient arrays (2*B floats)
    // sampler run state (device) + hipGraph replay of a pair of denoising steps
    int* d_step = nullptr;              // two counters: a step's kernels read d_step[parity], the sampler update writes d_step[parity ^ 1]
    bool final_fused = false;           // the final conv carries the sampler epilogue: no separate update / counter launches in the DDPM / DDIM loops
    int math_mode = 0;                  // g_math_mode / g_f16_raw at creation: one plan is built under ONE arithmetic even when the process-wide switches change while it builds
    int f16_raw = 1;
    int* d_range = nullptr;             // sticky device flag: an f16x2 conv on a raw input staged a value beyond the scaled half range (ConvArgs::range_flag)
    int n_range_convs = 0;              // launches of the step / cond-only programs that watch it
    int range_status(hipStream_t s, int* overflow);  // synchronises `s`, reads and clears the flag
    void* d_run = nullptr;            // SamplerRun
    float* d_tabs = nullptr;
    int tabs_cap = 0;
    hipStream_t cap_stream = nullptr;
    void* graph_exec[2] = {nullptr, nullptr};  // [0] DDPM pair, [1] DDIM pair
    bool use_graph = true;

    // profiling
    int prof_every = 0, prof_max = 0;
    bool prof_all = false;  // time every op of a profiled step (per-class breakdown), not only the dominant class
    std::vector<hipEvent_t> ev0, ev1;
    std::vector<double> ev_flop, ev_bytes, ev_mflop;
    std::vector<int> ev_cls;
    int ev_used = 0;
    long long prof_steps = 0;  // whole steps recorded since ddif_prof_begin (a step that does not fit the remaining events is not profiled)
    std::string prof_name;
    ddif_prof_class cls_res[6] = {};  // per-class sums of the last ddif_prof_collect

    // ---- liveness-based arena for the activations of the step program (built in two passes: a dry pass records, for every
    //      step tensor, the first / last launch that touches it; the real pass places them with first-fit interval colouring)
    struct Live { size_t bytes; int first, last; size_t off; };
    std::vector<Live> lives;
    std::map<const void*, int> fake2id;   // dry pass: fake address -> allocation index
    bool dry = false;                     // dry pass: no device allocation, no device work
    size_t dry_next = 0;                  // dry pass: next fake address offset
    int arena_next = 0;                   // real pass: index of the next step tensor
    char* arena = nullptr;
    size_t arena_bytes = 0, unaliased_bytes = 0;
    void use(const void* p);              // dry pass: the launch being created reads or writes p

    ~Plan();
    int build();
    int build_impl();
    template <typename T>
    int dalloc(T** p, size_t n);
    int alloc_tensor(Tensor* t, int C, int H, int W, bool step_act = false);
    int add_conv(std::vector<Op>& prog, const ConvSpec& s, Tensor* out);
    int ensure_tb(int rows);
    int time_rows(const float* t_host, int rows, hipStream_t s);
    void run_prog(std::vector<Op>& prog, hipStream_t s, const StepCtx& ctx, bool prof);
    int n_conv3_f16 = 0;
    int n_conv3_b1 = 0;  // ... on the bf16x1 throughput variant
    int n_conv3 = 0, n_conv3_x3 = 0;  // 3x3 conv ops of the step program / of them on the bf16x3 path (reported by prof_collect)
    bool op_timing_done = false;  // DDIF_OP_TIMING=<csv path>: one profiled step is timed op by op (development aid)

    // ---- native training step (ddif_train.cpp): the forward builder (ddif_plan.cpp, train_mode) records one TrainMod per module of
    //      UNetSR3.forward with every tensor the reverse pass needs; build_backward() turns the list into the reverse launch program
    struct TrainMod {
        enum Kind { STEM, FILM, RES, ATTN, DOWN, UP, DEC, FINAL } kind = STEM;
        std::string key;             // state-dict prefix of the module (e.g. "downs.3.cond_inj", "ups.2.res_block", "mid.0.attn")
        Tensor in, out;              // chain input / output
        Tensor t[12];                // module-specific saved tensors (see ddif_train.cpp)
        float* mask = nullptr;       // RES: dropout mask of block2 (NHWC)
        float* scale = nullptr;      // DEC: per-sample DropPath scales
        float* ctx = nullptr;        // DEC: linear-attention context [B][fea * d] of the forward (read by the reverse pass)
        float* la_part = nullptr;    // DEC: per-workgroup partial contexts (scratch of this block's launches)
        int slot = -1;               // RES: offset of the block's FeatureWiseAffine row in a time-bias row
        int lev = 0;
        bool has_res = false;        // DEC: attn_res is a conv (fea != dim_out)
        bool pushes_feat = false;    // the output is also a skip connection (encoder feature)
        int skip_from = -1;          // DEC: index (in tmods) of the module whose output is this block's skip input
    };
    std::vector<TrainMod> tmods;
    std::vector<Tensor> cenc_pad, kdw_pad_unused;  // train: cond images zero-padded to 4 | channels (weight gradients)
    std::map<std::string, float*> grad_slots;      // state-dict key -> bound gradient tensor (reference layout), ddif_plan_train_bind
    std::vector<std::function<void(hipStream_t)>> bwd;  // reverse program, in FORWARD order (run back to front)
    struct TrainScratch;
    std::shared_ptr<TrainScratch> ts;
    // reverse pass on two streams: the weight-gradient launches (large, compute-bound, leaves of the graph) go to `wg_stream` while the gradient
    // chain (many small latency-bound launches) continues on the caller's stream; events order them (ddif_train.cpp).  Off in the emulator.
    hipStream_t wg_stream = nullptr;
    hipEvent_t wg_fork = nullptr, wg_join = nullptr, side_read = nullptr, a_free[3] = {nullptr, nullptr, nullptr};
    bool wg_async = false;
    bool side_pending = false;                 // set_cond issued side-stream work that no launch of the caller's stream has waited for yet
    hipStream_t train_fork(hipStream_t main);  // the stream a weight-gradient launch goes to, ordered after everything issued on `main` so far
    void train_join(hipStream_t main);         // `main` waits for everything issued on the side stream
    float* d_loss = nullptr;          // device scalar
    float* dtb = nullptr;             // [B][nslots] gradient of the time-bias rows
    float* taux = nullptr;            // [B][32 + 128 + 128 + 32] pe | pre-activation | hidden | temb of the time MLP (train forward)
    float* d_net_out = nullptr;       // d(loss)/d(net_out), NHWC
    int build_backward();
    int train_step(const float* x0, const float* noise, const float* a_h, const float* s_h, const float* t_h, const float* sc, float* loss_dev, float* pred,
                   hipStream_t s);
    int train_bind(int n, const char* const* keys, float* const* grads);
    int train_forward_backward(const float* x, const float* t_h, const float* sc, const float* target, float* loss_dev, float* pred, hipStream_t s);
    int train_core(const float* t_h, bool has_sc, const float* target_nhwc, float* loss_dev, float* pred, hipStream_t s);
    int train_backward(const float* target_nhwc, float upstream, float* loss_dev, hipStream_t s);
    void train_set_stem_source(const float* sc_nhwc);
    int time_rows_aux(const float* t_host, int rows, float* aux, hipStream_t s);

    int train_set_dropout(int site, const float* mask_nchw, hipStream_t s);
    int train_set_droppath(const float* scales_host, hipStream_t s);
    int train_get_dropout(int site, float* mask_nchw, hipStream_t s);
    int train_get_droppath(float* scales_dev, hipStream_t s);
    int train_random_masks(uint64_t seed, uint64_t tile0, float p_drop, float p_path, hipStream_t s);
    int set_cond(const float* cond, hipStream_t s);
    int forward(const float* x, const float* t_host, const float* sc, float* out, hipStream_t s);
    int run_sampler(int kind, int n_steps, const float* const* tabs_host, int n_tabs, const float* t_model, const float* xT,
                    const float* noise, uint64_t seed, uint64_t tile0, float lo, float hi, int do_clamp, float* out, hipStream_t s);
    void drop_graphs();
    int sample_ddpm(const ddif_ddpm_tables* t, const float* xT, const float* noise, uint64_t seed, uint64_t tile0,
                    float lo, float hi, int do_clamp, float* out, hipStream_t s);
    int sample_ddim(const ddif_ddim_tables* t, const float* xT, const float* noise, uint64_t seed, uint64_t tile0,
                    float lo, float hi, int do_clamp, float* out, hipStream_t s);
    int sample_dpmpp(const ddif_dpm_tables* t, const float* xT, float lo, float hi, int do_clamp, float* out,
                     hipStream_t s);
    int q_sample_forward(const float* x0, const float* noise, const float* a_h, const float* s_h, const float* t_h,
                         const float* sc, float* pred, hipStream_t s);
};

}  // namespace ddif
