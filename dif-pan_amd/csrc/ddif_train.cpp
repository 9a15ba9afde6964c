// Native training step (BASELINE configs[4]): the reverse launch program of UNetSR3 under .train().
//
// Reference: `diff_loss, recon = diffusion(res, cond=cond); diff_loss.backward()` (diffusion_engine.py:230-233) -- autograd through
// models/sr3_dwt.py:169-219 (+ F.l1_loss, diffusion/diffusion_ddpm_pan.py:742-749).  Here the forward is the train-mode launch program of
// ddif_plan.cpp (every activation NHWC, kept until the reverse pass; Plan::tmods records one entry per module with the tensors it saved)
// and this file builds the reverse program from that list: per module a closure that enqueues its backward kernels --
//   * conv dgrad   = the forward implicit-GEMM kernels (bf16x3 split products / low-resolution split-K kernel, kernels_conv.h / kernels_lr.h)
//                    on transposed + tap-flipped weights packed on the device by the same launch that refreshes the forward packs
//                    (Net::build_dgrad_packs, kernels_refresh.h);  real 1x1 kernels for 1x1 convs;
//   * conv wgrad   = conv3x3_wgrad_kernel (exact-fp32 MFMA, fixed-order split-K reduction), centre tap only for 1x1 convs;
//   * GroupNorm (+ SiLU, + Dropout mask) backward, FiLM, SiLU, DropPath, linear attention, self-attention, depthwise convs: kernels_bwd.h /
//     kernels_train.h, all NHWC -- no layout conversion anywhere between the boundary tensors.
// Gradients are written (not accumulated) into caller-bound tensors in the reference's parameter layouts (ddif_plan_train_bind).
// Everything is deterministic: fixed-order reductions, no atomics.
#include <algorithm>
#include <cstring>

#include "ddif_plan.h"
#include "kernels_train.h"
#include "kernels_linattn.h"

namespace ddif {

static bool train_x3_() {  // DDIF_TRAIN_X3=0: exact-fp32 MFMA for the dgrad convs (as ddif_bwd.cpp)
    static const bool v = [] { const char* e = getenv("DDIF_TRAIN_X3"); return !e || atoi(e) != 0; }();
    return v;
}
static inline void dd_wait_event(hipStream_t s, hipEvent_t e) {  // (the host emulator runs everything in order on one stream: never reached there)
#ifndef DDIF_EMU
    (void)hipStreamWaitEvent(s, e, 0);
#else
    (void)s;
    (void)e;
#endif
}
static inline dim3 tgrid(size_t n) {
    size_t g = (n + 255) / 256;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return dim3((unsigned)g);
}

namespace tk {
int film_chunks(int HW, int C) {
    int chunks = (HW * C / 4 + 256 * 8 - 1) / (256 * 8);
    return chunks < 1 ? 1 : chunks;
}
void film_apply(hipStream_t s, const float* xc, const float* film, int B, int HW, int C, float* out, double* st_out, int chunks) {
    hipLaunchKernelGGL(film_apply_kernel, dim3(chunks, B), dim3(256), 64, s, xc, film, HW, C, out, st_out);
}
// linear attention of the train-mode forward: k side (partial contexts per row group) -> reduce -> q side.  ctx [B][C * d] is kept for the reverse pass
size_t linattn_part_floats(int B, int H, int W, int C, int d) {
    const int gk = la_groups(W, C, H), gq = la_groups(H, C, W);
    return (size_t)B * (gk > gq ? gk : gq) * C * d;
}
// the context depends on kv = kv.1(kv.0(cond)) only: the training plan computes it with the cond-only program (ddif_plan_set_cond), once per iteration,
// and both forward passes of the iteration (self-conditioning, main) only run the q side
void linattn_ctx(hipStream_t s, const float* kv, int B, int heads, int d, int H, int W, float* ctx, float* part) {
    const int C = heads * d, gk = la_groups(W, C, H);
    hipLaunchKernelGGL(la_kside_fwd_kernel, dim3(gk, B), dim3(LA_THREADS), la_kside_fwd_smem(H, W, C), s, kv, H, W, C, d, part);
    hipLaunchKernelGGL(la_reduce_kernel, tgrid((size_t)B * C * d), dim3(256), 0, s, (const float*)part, B, gk, C * d, ctx);
}
void linattn_apply(hipStream_t s, const float* q, const float* ctx, int B, int heads, int d, int H, int W, float* out, int ld_o) {
    const int C = heads * d, gq = la_groups(H, C, W);
    hipLaunchKernelGGL(la_qside_fwd_kernel, dim3(gq, B), dim3(LA_THREADS), la_qside_fwd_smem(H, W, C, d), s, q, ctx, H, W, C, d, 1.0f / std::sqrt((float)d), out, ld_o);
}
void linattn_fwd(hipStream_t s, const float* q, const float* kv, int B, int heads, int d, int H, int W, float* out, int ld_o, float* ctx, float* part) {
    linattn_ctx(s, kv, B, heads, d, H, W, ctx, part);
    linattn_apply(s, q, ctx, B, heads, d, H, W, out, ld_o);
}
// reverse: q side (dq_pre + partial dctx per column group) -> reduce -> k side (dk_pre, dv)
void linattn_bwd_q(hipStream_t s, const float* q, const float* dout, int ld_g, const float* ctx, int B, int heads, int d, int H, int W, float* dq, float* dctx, float* part) {
    const int C = heads * d, gq = la_groups(H, C, W);
    hipLaunchKernelGGL(la_qside_bwd_kernel, dim3(gq, B), dim3(LA_THREADS), la_qside_bwd_smem(H, W, C, d), s, q, dout, ld_g, ctx, H, W, C, d, 1.0f / std::sqrt((float)d), dq,
                       part);
    hipLaunchKernelGGL(la_reduce_kernel, tgrid((size_t)B * C * d), dim3(256), 0, s, (const float*)part, B, gq, C * d, dctx);
}
// the k side only feeds kv's own backward (cond needs no gradient): a leaf subtree of the reverse graph
void linattn_bwd_k(hipStream_t s, const float* kv, const float* dctx, int B, int heads, int d, int H, int W, float* dkv) {
    const int C = heads * d, gk = la_groups(W, C, H);
    hipLaunchKernelGGL(la_kside_bwd_kernel, dim3(gk, B), dim3(LA_THREADS), la_kside_bwd_smem(H, W, C, d), s, kv, dctx, H, W, C, d, dkv);
}
void linattn_bwd(hipStream_t s, const float* q, const float* kv, const float* dout, int ld_g, const float* ctx, int B, int heads, int d, int H, int W, float* dq, float* dkv,
                 float* dctx, float* part) {
    linattn_bwd_q(s, q, dout, ld_g, ctx, B, heads, d, H, W, dq, dctx, part);
    linattn_bwd_k(s, kv, dctx, B, heads, d, H, W, dkv);
}
int linattn_prepare() {
    DDIF_HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(la_kside_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    DDIF_HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(la_qside_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    DDIF_HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(la_qside_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    DDIF_HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(la_kside_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    return 0;
}
}  // namespace tk

struct Plan::TrainScratch {
    float *a = nullptr, *a_ring[3] = {nullptr, nullptr, nullptr}, *tmp = nullptr, *tmp2 = nullptr, *partial = nullptr, *bpart = nullptr, *wbpart = nullptr, *S = nullptr, *wpad = nullptr;
    double *spart = nullptr, *cpart = nullptr, *dwpart = nullptr;
    size_t n_a = 0, n_tmp = 0, n_partial = 0, n_bpart = 0, n_cpart = 0, n_wpad = 0, n_dwpart = 0;
    const float* stem_sc = nullptr;  // set by train_step: the self-conditioning source of THIS iteration (sc_in or x_in)
    // time MLP backward
    float *dte = nullptr, *dh1 = nullptr, *ds = nullptr, *dwall = nullptr, *dball = nullptr;
    SlotScatter* slot_tab = nullptr;
    std::vector<SlotScatter> slot_host;
    // GroupNorm backward: every pass keeps its per-chunk partials in a buffer of its own; ONE launch at the end of the reverse program turns all of them into dgamma / dbeta
    struct GnSite { double* cpart; double* gpart; float** dg; float** dbt; int C, nchunk; };
    std::vector<GnSite> gn_sites;
    std::vector<GnRedRec> gnred_host;
    GnRedRec* gnred_dev = nullptr;
    hipError_t upload_err = hipSuccess;  // a table upload inside the reverse program failed: the step reports it (train_backward) and the host mirror stays stale, so the next step retries
};

int Plan::train_bind(int n, const char* const* keys, float* const* grads) {
    if (!train_mode) return fail(DDIF_ERR_STATE, "ddif_plan_train_bind: not a train-mode plan");
    if (n < 1 || !keys || !grads) return fail(DDIF_ERR_INVALID, "ddif_plan_train_bind: bad arguments");
    std::map<std::string, float*> given;
    for (int i = 0; i < n; ++i) {
        if (!keys[i] || !grads[i]) return fail(DDIF_ERR_INVALID, "ddif_plan_train_bind: NULL key / pointer at %d", i);
        given[keys[i]] = grads[i];
    }
    for (auto& kv : grad_slots) {
        auto it = given.find(kv.first);
        if (it == given.end()) return fail(DDIF_ERR_MISSING, "ddif_plan_train_bind: no gradient tensor for '%s'", kv.first.c_str());
        kv.second = it->second;
    }
    return 0;
}

int Plan::build_backward() {
#define DDIF_TRY(x) do { if (int e__ = (x)) return e__; } while (0)
    ts = std::make_shared<TrainScratch>();
    TrainScratch* T = ts.get();
    bwd.clear();
    grad_slots.clear();
    const int BB = B;
    auto V = [&](const std::string& k) -> const float* {
        auto it = net->vec.find(k);
        return it == net->vec.end() ? nullptr : it->second;
    };
    auto G = [&](const std::string& k) -> float** { return &grad_slots[k]; };  // std::map nodes are stable
    auto need = [](size_t& cur, size_t n) { if (n > cur) cur = n; };
    auto numel = [&](const Tensor& t) { return (size_t)BB * t.H * t.W * t.C; };
    auto fbuf = [&](float** p, size_t n) -> int { return dalloc(p, n); };
    int ring_ctr = 0;  // slots of T->a_ring handed to the GroupNorm-recompute / weight-gradient pairs in turn
    // a GroupNorm backward site: its own partial buffers (the chip has 288 GB; all of them together are a few hundred MB at B = 32), reduced at the end of the program
    auto gn_site = [&](int C, int nchunk, float** dg, float** dbt, double** cpart, double** gpart) -> int {
        DDIF_TRY(dalloc(cpart, (size_t)BB * nchunk * C * 2 + 64));
        DDIF_TRY(dalloc(gpart, (size_t)BB * nchunk * 2 + 64));
        T->gn_sites.push_back({*cpart, *gpart, dg, dbt, C, nchunk});
        return 0;
    };
    DDIF_TRY(tk::wgrad_prepare());
    DDIF_TRY(tk::linattn_prepare());
#ifndef DDIF_EMU
    {   // the side stream of the weight gradients (DDIF_TRAIN_STREAMS=0: everything on the caller's stream)
        const char* env = getenv("DDIF_TRAIN_STREAMS");  // read per plan (a test builds one plan of each kind in one process)
        const bool two = !env || atoi(env) != 0;
        if (two && !wg_stream) {
            // lowest priority: the gradient chain on the caller's stream is a sequence of small dependent launches -- it should never queue behind
            // the weight gradients' workgroups, which only fill the compute units it leaves idle
            int least = 0, greatest = 0;
            DDIF_HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
            DDIF_HIPCHK(hipStreamCreateWithPriority(&wg_stream, hipStreamNonBlocking, least));
            DDIF_HIPCHK(hipEventCreateWithFlags(&wg_fork, hipEventDisableTiming));
            DDIF_HIPCHK(hipEventCreateWithFlags(&wg_join, hipEventDisableTiming));
            for (auto& e : a_free) DDIF_HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            DDIF_HIPCHK(hipEventCreateWithFlags(&side_read, hipEventDisableTiming));
            net->reader_events.push_back(side_read);
        }
        wg_async = two;
    }
#endif

    // ---- building blocks (each appends launches to a closure list `L` executed in order)
    using Launch = std::function<void(hipStream_t)>;
    struct Seq { std::vector<Launch> v; };

    // dgrad conv of forward conv `key`: dy [B,H,W,Cout_f] -> dx [B,H,W,(Cin_f padded to 4)]
    //   side: the conv belongs to a leaf subtree whose input was produced on the side stream -- it is issued there too (in order behind it)
    auto add_dgrad = [&](Seq& L, const std::string& key, Tensor dy, Tensor* dx, bool side = false) -> int {
        auto it = net->dconv.find(key);
        if (it == net->dconv.end()) return fail(DDIF_ERR_MISSING, "training: no dgrad weights for %s", key.c_str());
        auto prog = std::make_shared<std::vector<Op>>();
        ConvSpec s;
        s.pc = &it->second;
        s.in0 = dy;
        s.in0.st = nullptr;
        s.use_bias = false;
        s.exact = !train_x3_();
        s.name = "dgrad";
        if (int e = add_conv(*prog, s, dx)) return e;
        L.v.push_back([this, prog, side](hipStream_t st) {
            StepCtx ctx;
            hipStream_t rs = (side && this->wg_async) ? this->wg_stream : st;
            for (auto& op : *prog) op.run(rs, ctx);
        });
        return 0;
    };
    // weight (+ bias) gradient of a conv with input x [B,H,W,Cin] (Cin % 4 == 0) and output gradient dy [B,H,W,Cout]
    auto add_wgrad = [&](Seq& L, const float* x, int Cin, const float* dy, int Cout, int H_, int W_, int ks, float** dw, float** db) -> int {
        if (Cin % 4 || Cout % 4) return fail(DDIF_ERR_INVALID, "training: weight gradient needs 4 | channels (got %d -> %d)", Cin, Cout);
        const tk::WgradGeom g = tk::wgrad_geom(BB, Cin, Cout, H_, W_, ks == 1);
        if (g.rb < 1) return fail(DDIF_ERR_INVALID, "training: W=%d is too wide for the weight-gradient kernel", W_);
        need(T->n_partial, g.partial_floats);
        need(T->n_bpart, (size_t)g.nsplit * g.n_co * 32);
        L.v.push_back([=](hipStream_t st) { tk::wgrad(this->train_fork(st), x, dy, BB, H_, W_, Cin, Cout, g, ks == 1, T->partial, *dw, T->wbpart, db ? *db : nullptr); });
        return 0;
    };
    // backward of  [GroupNorm (+SiLU) (+mask)] -> conv  (one `Block` of the reference, :288-300, and the other normalised convs)
    //   pro: 0 none, 1 GroupNorm, 2 GroupNorm + SiLU.  a_mat: the conv's input if the forward materialised it (block2's dropped activation).
    //   res: a gradient to add to dx (the residual path of a ResnetBlock / SelfAttention), folded into GroupNorm's dx launch.
    auto add_block_bwd = [&](Seq& L, Tensor x, int pro, const std::string& nkey, const float* mask, const float* a_mat, const std::string& ckey, int ks, bool has_bias,
                             Tensor dy, bool need_dx, float** dx_out, const float* res = nullptr) -> int {
        const int HW = x.H * x.W, C = x.C;
        const size_t n = numel(x);
        const int nchunk = HW < 32 ? HW : 32;
        const float *gamma = nullptr, *beta = nullptr;
        const double* fst = nullptr;
        int fnp = 0, ring = 0;
        if (pro) {
            gamma = V(nkey + ".weight");
            beta = V(nkey + ".bias");
            if (!gamma || !beta) return fail(DDIF_ERR_MISSING, "training: %s weights missing", nkey.c_str());
            need(T->n_cpart, (size_t)BB * nchunk * C * 2);
            if (!a_mat) need(T->n_a, n);
            // statistics of x: the fp64 partials its forward producer left behind (no second read of x); recomputed if there are none
            if (x.st && x.np > 0 && !getenv("DDIF_TRAIN_GN_RESTAT")) {
                fst = x.st;
                fnp = x.np;
            } else {
                fnp = nchunk;
                L.v.push_back([=](hipStream_t st) { tk::gn_stats(st, x.p, BB, (size_t)HW * C, nchunk, T->spart); });
            }
            if (!a_mat) {
                const double* fs = fst;
                const int fn = fnp;
                ring = ring_ctr++ % 3;
                const int rk = ring;
                L.v.push_back([=](hipStream_t st) {
                    if (this->wg_async) dd_wait_event(st, this->a_free[rk]);  // the weight gradient that read this slot three blocks ago is done
                    tk::gn_act(st, x.p, fs ? fs : T->spart, fn, gamma, beta, mask, BB, HW, C, pro == 2, T->a_ring[rk]);
                });
            }
        }
        float** dw = G(ckey + ".weight");
        float** db = has_bias ? G(ckey + ".bias") : nullptr;
        // (the scratch pointer T->a is read when the launch runs, not now)
        if (pro && !a_mat) {
            const tk::WgradGeom g = tk::wgrad_geom(BB, C, dy.C, x.H, x.W, ks == 1);
            if (g.rb < 1) return fail(DDIF_ERR_INVALID, "training: W=%d is too wide for the weight-gradient kernel", x.W);
            need(T->n_partial, g.partial_floats);
            need(T->n_bpart, (size_t)g.nsplit * g.n_co * 32);
            const int Co = dy.C, H_ = x.H, W_ = x.W;
            const int rk = ring;
            L.v.push_back([=](hipStream_t st) {
                hipStream_t ws = this->train_fork(st);
                tk::wgrad(ws, T->a_ring[rk], dy.p, BB, H_, W_, C, Co, g, ks == 1, T->partial, *dw, T->wbpart, db ? *db : nullptr);
                if (this->wg_async) (void)hipEventRecord(this->a_free[rk], ws);
            });
        } else {
            DDIF_TRY(add_wgrad(L, a_mat ? a_mat : x.p, C, dy.p, dy.C, x.H, x.W, ks, dw, db));
        }
        if (!need_dx && !pro) return 0;
        Tensor da;
        DDIF_TRY(add_dgrad(L, ckey, dy, &da));
        if (da.C != C) return fail(DDIF_ERR_STATE, "training: dgrad of %s yields %d channels, expected %d", ckey.c_str(), da.C, C);
        if (!pro) {
            if (res) return fail(DDIF_ERR_STATE, "training: residual fusion needs the GroupNorm form of the block");
            *dx_out = da.p;
            return 0;
        }
        float* dx = nullptr;
        if (need_dx) DDIF_TRY(fbuf(&dx, n));
        float** dg = G(nkey + ".weight");
        float** dbt = G(nkey + ".bias");
        double *cp = nullptr, *gp = nullptr;
        DDIF_TRY(gn_site(C, nchunk, dg, dbt, &cp, &gp));
        L.v.push_back([=](hipStream_t st) {
            tk::gn_bwd(st, x.p, da.p, mask, fst ? fst : T->spart, fnp, gamma, beta, BB, HW, C, nchunk, pro == 2, cp, nullptr, nullptr, nullptr, res, dx, gp);
        });
        *dx_out = dx;
        return 0;
    };
    auto add_add2 = [&](Seq& L, const float* a, const float* b, size_t n, float* out) {
        L.v.push_back([=](hipStream_t st) { hipLaunchKernelGGL(add2_kernel, tgrid(n), dim3(256), 0, st, a, b, n, out); });
    };

    // ---- cond-only padding ops of the weight gradients over 9 / 11-channel cond images (part of set_cond)
    {
        std::vector<const float*> done;
        for (auto& m : tmods) {
            Tensor src, dst;
            if (m.kind == TrainMod::FILM) {
                src = m.t[3];
                dst = m.t[4];
            } else if (m.kind == TrainMod::DEC) {
                src = m.t[5];
                dst = m.t[6];
            } else {
                continue;
            }
            if (std::find(done.begin(), done.end(), (const float*)dst.p) != done.end()) continue;
            done.push_back(dst.p);
            const size_t npix = (size_t)BB * src.H * src.W;
            Op op;
            op.name = "pad_channels";
            op.run = [src, dst, npix](hipStream_t st, const StepCtx&) {
                hipLaunchKernelGGL(pad_channels_kernel, tgrid(npix * dst.C), dim3(256), 0, st, (const float*)src.p, src.C, dst.C, npix, dst.p);
            };
            op.side = true;  // read by the reverse pass only; behind the decoder-only ops that produce their inputs
            pre.push_back(std::move(op));
        }
    }

    // ---- loss gradient buffer, time-bias gradient rows
    DDIF_TRY(fbuf(&d_net_out, numel(net_out)));
    DDIF_TRY(fbuf(&d_loss, 64));
    DDIF_TRY(fbuf(&dtb, (size_t)BB * net->nslots));
    const int inner = net->cfg.inner_channel;
    DDIF_TRY(fbuf(&taux, (size_t)BB * 10 * inner));

    // ---- walk the modules back to front
    float* g = d_net_out;
    std::map<int, float*> skipg;
    std::vector<Seq> seqs;  // in EXECUTION order (reverse of the forward)
    for (int i = (int)tmods.size() - 1; i >= 0; --i) {
        const TrainMod& m = tmods[i];
        Seq L;
        float* dy = g;
        if (m.pushes_feat) {
            auto it = skipg.find(i);
            if (it == skipg.end()) return fail(DDIF_ERR_STATE, "training: encoder feature %d has no decoder consumer", i);
            float* sum = nullptr;
            DDIF_TRY(fbuf(&sum, numel(m.out)));
            add_add2(L, g, it->second, numel(m.out), sum);
            dy = sum;
        }
        Tensor dyT = m.out;
        dyT.p = dy;
        dyT.st = nullptr;
        float* dx = nullptr;
        switch (m.kind) {
            case TrainMod::FINAL: {
                DDIF_TRY(add_block_bwd(L, m.in, 2, "final_conv.block.0", nullptr, nullptr, "final_conv.block.3", 3, true, dyT, true, &dx));
                break;
            }
            case TrainMod::RES: {
                const Tensor h1 = m.t[0], y2 = m.t[1];
                const std::string rb = m.key;
                // block2: GroupNorm + SiLU + Dropout -> conv (the dropped activation y2 was materialised by the forward)
                float* dh1 = nullptr;
                DDIF_TRY(add_block_bwd(L, h1, 2, rb + ".block2.block.0", m.mask, y2.p, rb + ".block2.block.3", 3, true, dyT, true, &dh1));
                // FeatureWiseAffine: h1 = conv1(...) + row[b][c]  ->  d(row) = plane sums of dh1
                {
                    const int HW = h1.H * h1.W, C = h1.C, ld = net->nslots, slot = m.slot;
                    float* dst = dtb + slot;
                    const int nck = HW >= 1024 ? 16 : (HW >= 256 ? 4 : 1);
                    need(T->n_bpart, (size_t)BB * nck * C);
                    L.v.push_back([=](hipStream_t st) {  // a leaf (the time MLP's backward runs after the join): on the side stream; T->bpart is used there only
                        hipStream_t ws = this->train_fork(st);
                        hipLaunchKernelGGL(plane_sum_partial_nhwc_kernel, dim3(nck, BB), dim3(256), 256 * sizeof(float), ws, (const float*)dh1, HW, C, nck, T->bpart);
                        hipLaunchKernelGGL(plane_sum_final_kernel, dim3((BB * C + 255) / 256), dim3(256), 0, ws, (const float*)T->bpart, BB, C, nck, ld, dst);
                    });
                }
                Tensor dh1T = h1;
                dh1T.p = dh1;
                dh1T.st = nullptr;
                DDIF_TRY(add_block_bwd(L, m.in, 2, rb + ".block1.block.0", nullptr, nullptr, rb + ".block1.block.3", 3, true, dh1T, true, &dx, dy));  // + the residual path
                break;
            }
            case TrainMod::ATTN: {
                const Tensor qkv = m.t[0], o = m.t[1];
                const int n = m.in.H * m.in.W, C = m.in.C, d = C / 8;
                if (n > 64 || d > 32) return fail(DDIF_ERR_INVALID, "training: self-attention over %d tokens (head dim %d): the reverse pass handles <= 64 tokens", n, d);
                float* do_ = nullptr;
                DDIF_TRY(add_block_bwd(L, o, 0, "", nullptr, nullptr, m.key + ".out", 1, true, dyT, true, &do_));
                float* dqkv = nullptr;
                DDIF_TRY(fbuf(&dqkv, numel(qkv)));
                {
                    const size_t sm = (size_t)(4 * d * (n + 1) + 2 * n * (n + 1) + n) * sizeof(float);
                    const float sc = 1.0f / std::sqrt((float)C);
                    L.v.push_back([=](hipStream_t st) {
                        hipLaunchKernelGGL(selfattn_bwd_nhwc_kernel, dim3(BB * 8), dim3(256), sm, st, (const float*)qkv.p, (const float*)do_, 8, d, n, sc, dqkv);
                    });
                }
                Tensor dq = qkv;
                dq.p = dqkv;
                dq.st = nullptr;
                DDIF_TRY(add_block_bwd(L, m.in, 1, m.key + ".norm", nullptr, nullptr, m.key + ".qkv", 1, false, dq, true, &dx, dy));  // + the residual path
                break;
            }
            case TrainMod::FILM: {
                const Tensor xc = m.t[0], film = m.t[1], hid = m.t[2], cpad = m.t[4];
                const std::string ci = m.key;
                float *dxc = nullptr, *dfilm = nullptr;
                DDIF_TRY(fbuf(&dxc, numel(xc)));
                DDIF_TRY(fbuf(&dfilm, numel(film)));
                {
                    const size_t npix = (size_t)BB * xc.H * xc.W;
                    const int C = xc.C;
                    L.v.push_back([=](hipStream_t st) {
                        hipLaunchKernelGGL(film_bwd_nhwc_kernel, tgrid(npix * C), dim3(256), 0, st, (const float*)xc.p, (const float*)film.p, (const float*)dy, npix, C, dxc, dfilm);
                    });
                }
                Tensor dxcT = xc;
                dxcT.p = dxc;
                DDIF_TRY(add_block_bwd(L, m.in, 0, "", nullptr, nullptr, ci + ".x_conv", 1, true, dxcT, true, &dx));
                Tensor dfT = film;
                dfT.p = dfilm;
                float* dhid = nullptr;
                DDIF_TRY(add_block_bwd(L, hid, 2, ci + ".body.1", nullptr, nullptr, ci + ".body.3", 1, true, dfT, true, &dhid));
                // body.0: conv3x3 over the (zero-padded) cond image, no bias, no input gradient
                {
                    const int Cp = cpad.C, Cr = m.t[3].C, Co = hid.C;
                    need(T->n_wpad, (size_t)Co * Cp * 9);
                    float** dwp = &T->wpad;
                    DDIF_TRY(add_wgrad(L, cpad.p, Cp, dhid, Co, hid.H, hid.W, 3, dwp, nullptr));
                    float** dw = G(ci + ".body.0.weight");
                    L.v.push_back([=](hipStream_t st) {  // (in order behind its weight gradient, on that stream)
                        hipLaunchKernelGGL(unpad_weight_kernel, tgrid((size_t)Co * Cr * 9), dim3(256), 0, this->wg_async ? this->wg_stream : st, (const float*)T->wpad, Co, Cp, Cr, 9, *dw);
                    });
                }
                break;
            }
            case TrainMod::DOWN: {
                const int H_ = m.in.H, W_ = m.in.W, C = m.in.C, Ho = m.out.H, Wo = m.out.W, Co = m.out.C;
                float* dyz = nullptr;
                DDIF_TRY(fbuf(&dyz, (size_t)BB * H_ * W_ * Co));
                L.v.push_back([=](hipStream_t st) {
                    hipLaunchKernelGGL(zero_stuff_nhwc_kernel, tgrid((size_t)BB * H_ * W_ * Co), dim3(256), 0, st, (const float*)dy, BB, Co, Ho, Wo, H_, W_, dyz);
                });
                Tensor dz;
                dz.p = dyz;
                dz.C = Co;
                dz.H = H_;
                dz.W = W_;
                DDIF_TRY(add_wgrad(L, m.in.p, C, dyz, Co, H_, W_, 3, G(m.key + ".weight"), G(m.key + ".bias")));  // (the inserted zeros add nothing to the bias sums)
                Tensor dxT;
                DDIF_TRY(add_dgrad(L, m.key, dz, &dxT));
                dx = dxT.p;
                break;
            }
            case TrainMod::UP: {
                const int H_ = m.in.H, W_ = m.in.W, C = m.in.C, Co = m.out.C;
                need(T->n_tmp, (size_t)BB * 4 * H_ * W_ * C);
                L.v.push_back([=](hipStream_t st) {
                    hipLaunchKernelGGL(upsample2_nhwc_kernel, tgrid((size_t)BB * 4 * H_ * W_ * C), dim3(256), 0, st, (const float*)m.in.p, BB, C, H_, W_, T->tmp);
                });
                // (T->tmp is read when the launch runs)
                {
                    const tk::WgradGeom gg = tk::wgrad_geom(BB, C, Co, 2 * H_, 2 * W_);
                    if (gg.rb < 1) return fail(DDIF_ERR_INVALID, "training: W=%d is too wide for the weight-gradient kernel", 2 * W_);
                    need(T->n_partial, gg.partial_floats);
                    need(T->n_bpart, (size_t)gg.nsplit * gg.n_co * 32);
                    float** dw = G(m.key + ".weight");
                    float** db = G(m.key + ".bias");
                    L.v.push_back([=](hipStream_t st) {
                        tk::wgrad(this->train_fork(st), T->tmp, dy, BB, 2 * H_, 2 * W_, C, Co, gg, false, T->partial, *dw, T->wbpart, *db);
                        this->train_join(st);  // T->tmp is scratch of the gradient chain: nothing may overwrite it before this launch has read it
                    });
                }
                Tensor dxu;
                DDIF_TRY(add_dgrad(L, m.key, dyT, &dxu));
                DDIF_TRY(fbuf(&dx, numel(m.in)));
                L.v.push_back([=](hipStream_t st) {
                    hipLaunchKernelGGL(sumpool2_nhwc_kernel, tgrid((size_t)BB * H_ * W_ * C), dim3(256), 0, st, (const float*)dxu.p, BB, C, H_, W_, dx);
                });
                break;
            }
            case TrainMod::STEM: {
                // input = cat[self_cond, x] (:174); padded to 4 | channels for the weight-gradient kernel.  No input gradient.
                const int Cx = x_in.C, Cs = net->cfg.self_condition ? sc_in.C : 0;
                const int Cin = Cs + Cx, Cp = (Cin + 3) & ~3, Co = m.out.C, H_ = m.out.H, W_ = m.out.W;
                const size_t npix = (size_t)BB * H_ * W_;
                need(T->n_tmp, npix * Cin);
                need(T->n_a, npix * Cp);
                const float* xin = x_in.p;
                L.v.push_back([=](hipStream_t st) {
                    if (Cs) hipLaunchKernelGGL(concat2_kernel, tgrid(npix * Cin), dim3(256), 0, st, T->stem_sc, Cs, xin, Cx, npix, T->tmp);
                    hipLaunchKernelGGL(pad_channels_kernel, tgrid(npix * Cp), dim3(256), 0, st, Cs ? (const float*)T->tmp : xin, Cin, Cp, npix, T->a);
                });
                {
                    const tk::WgradGeom gg = tk::wgrad_geom(BB, Cp, Co, H_, W_);
                    if (gg.rb < 1) return fail(DDIF_ERR_INVALID, "training: W=%d is too wide for the weight-gradient kernel", W_);
                    need(T->n_partial, gg.partial_floats);
                    need(T->n_bpart, (size_t)gg.nsplit * gg.n_co * 32);
                    need(T->n_wpad, (size_t)Co * Cp * 9);
                    float** dw = G(m.key + ".weight");
                    float** db = G(m.key + ".bias");
                    L.v.push_back([=](hipStream_t st) {
                        hipStream_t ws = this->train_fork(st);
                        tk::wgrad(ws, T->a, dy, BB, H_, W_, Cp, Co, gg, false, T->partial, T->wpad, T->wbpart, *db);
                        hipLaunchKernelGGL(unpad_weight_kernel, tgrid((size_t)Co * Cin * 9), dim3(256), 0, ws, (const float*)T->wpad, Co, Cp, Cin, 9, *dw);
                        this->train_join(st);
                    });
                }
                dx = nullptr;
                break;
            }
            case TrainMod::DEC: {
                const Tensor skip = m.t[0], xn = m.t[1], dwq = m.t[2], q = m.t[3], kv = m.t[4], kdw = m.t[5], kdwp = m.t[6], o = m.t[7], a = m.t[8], f0 = m.t[9], f1 = m.t[10],
                             f2 = m.t[11];
                const std::string ci = m.key;
                const int Hl = xn.H, Wl = xn.W, fea = xn.C, d = fea / 8, Co = a.C;
                const size_t npix = (size_t)BB * Hl * Wl;
                if (Hl > 64 || Wl > 64 || d > 32) return fail(DDIF_ERR_INVALID, "training: linear attention at %dx%d (head dim %d): the reverse pass handles <= 64x64, d <= 32", Hl, Wl, d);
                // DropPath: out = a + scale[b] * f3c
                float* df3c = nullptr;
                DDIF_TRY(fbuf(&df3c, numel(a)));
                {
                    const size_t per = (size_t)Hl * Wl * Co, tot = npix * Co;
                    const float* sc = m.scale;
                    L.v.push_back([=](hipStream_t st) { hipLaunchKernelGGL(scale_rows_kernel, tgrid(tot), dim3(256), 0, st, (const float*)dy, sc, per, tot, df3c); });
                }
                Tensor d3 = a;
                d3.p = df3c;
                d3.st = nullptr;
                float* df2 = nullptr;
                DDIF_TRY(add_block_bwd(L, f2, 0, "", nullptr, nullptr, ci + ".ffn.3", 1, true, d3, true, &df2));
                Tensor d2 = f2;
                d2.p = df2;
                d2.st = nullptr;
                float* df1 = nullptr;
                DDIF_TRY(add_block_bwd(L, f1, 0, "", nullptr, nullptr, ci + ".ffn.2", 3, false, d2, true, &df1));
                float* df0 = nullptr;
                DDIF_TRY(fbuf(&df0, numel(f0)));
                {
                    const size_t n = numel(f0);
                    L.v.push_back([=](hipStream_t st) { tk::silu_bwd(st, f0.p, df1, n, df0); });
                }
                Tensor d0 = f0;
                d0.p = df0;
                d0.st = nullptr;
                float* da_f = nullptr;
                DDIF_TRY(add_block_bwd(L, a, 0, "", nullptr, nullptr, ci + ".ffn.0", 3, false, d0, true, &da_f));
                float* da = nullptr;
                DDIF_TRY(fbuf(&da, numel(a)));
                add_add2(L, dy, da_f, numel(a), da);
                // a = attn_out(o) + b_o + [attn_res(xn) + b_r | xn]
                DDIF_TRY(add_wgrad(L, o.p, fea, da, Co, Hl, Wl, 1, G(ci + ".attn_out.weight"), G(ci + ".attn_out.bias")));
                if (m.has_res) {
                    DDIF_TRY(add_wgrad(L, xn.p, fea, da, Co, Hl, Wl, 1, G(ci + ".attn_res.weight"), G(ci + ".attn_res.bias")));
                }
                Tensor daT = a;
                daT.p = da;
                daT.st = nullptr;
                Tensor dcat;  // [do | dxn_a] (2 fea channels) or do (fea channels)
                DDIF_TRY(add_dgrad(L, ci + ".attn_mix", daT, &dcat));
                const int ldc = dcat.C;
                if (ldc != (m.has_res ? 2 * fea : fea)) return fail(DDIF_ERR_STATE, "training: %s.attn_mix dgrad yields %d channels", ci.c_str(), ldc);
                // linear attention core
                float *dq = nullptr, *dkv = nullptr;
                DDIF_TRY(fbuf(&dq, numel(q)));
                DDIF_TRY(fbuf(&dkv, numel(kv)));
                {
                    if (!m.ctx || !m.la_part) return fail(DDIF_ERR_STATE, "training: %s has no saved attention context", ci.c_str());
                    const float* ctx = m.ctx;
                    float* part = m.la_part;
                    float* dctx = nullptr;  // per block: the k side reads it from the side stream while the chain moves on
                    DDIF_TRY(fbuf(&dctx, (size_t)BB * fea * d));
                    L.v.push_back([=](hipStream_t st) {
                        tk::linattn_bwd_q(st, q.p, dcat.p, ldc, ctx, BB, 8, d, Hl, Wl, dq, dctx, part);
                        tk::linattn_bwd_k(this->train_fork(st), kv.p, dctx, BB, 8, d, Hl, Wl, dkv);  // kv's backward is a leaf subtree: side stream from here on
                    });
                }
                // q = q.1(dwq) + b;  dwq = depthwise3x3(xn; q.0)
                Tensor dqT = q;
                dqT.p = dq;
                dqT.st = nullptr;
                float* ddwq = nullptr;
                DDIF_TRY(add_block_bwd(L, dwq, 0, "", nullptr, nullptr, ci + ".q.1", 1, true, dqT, true, &ddwq));
                float* dxn = nullptr;  // total gradient of xn
                DDIF_TRY(fbuf(&dxn, numel(xn)));
                {
                    const float* w9 = V(ci + ".q.0.weight");  // [9][C], refreshed with the weights
                    need(T->n_tmp, numel(xn));
                    const int nsplit = std::min(512, BB * Hl);
                    need(T->n_dwpart, (size_t)nsplit * fea * 9);
                    float** dw0 = G(ci + ".q.0.weight");
                    const float* other = m.has_res ? dcat.p + fea : da;  // gradient of xn through attn_res (or the identity)
                    const int ld_other = m.has_res ? ldc : Co;
                    L.v.push_back([=](hipStream_t st) {
                        tk::dw3x3_plain(st, ddwq, fea, BB, Hl, Wl, w9, T->tmp, true);  // d(xn) through the depthwise conv: the same kernel on mirrored taps
                        hipLaunchKernelGGL(add2_ld_kernel, tgrid(npix * fea), dim3(256), 0, st, (const float*)T->tmp, fea, other, ld_other, fea, npix, dxn);
                        hipStream_t ws = this->train_fork(st);  // a leaf: beside the gradient chain, like the conv weight gradients
                        hipLaunchKernelGGL(dw_wgrad_partial_nhwc_kernel, dim3((fea + 31) / 32, nsplit), dim3(256), 8 * 32 * 9 * sizeof(double), ws, (const float*)xn.p, fea,
                                           (const float*)ddwq, fea, BB, fea, Hl, Wl, nsplit, T->dwpart);
                        hipLaunchKernelGGL(dw_wgrad_reduce_kernel, dim3((fea * 9 + 7) / 8), dim3(256), 256 * sizeof(double), ws, (const double*)T->dwpart, nsplit, fea, *dw0);
                    });
                }
                // kv = kv.1(kdw) + b;  kdw = depthwise3x3(cond; kv.0): weight gradients only (cond needs no gradient)
                {
                    const int cd = kdw.C, Cp = kdwp.C;
                    need(T->n_wpad, (size_t)2 * fea * Cp);
                    float** dwp = &T->wpad;
                    DDIF_TRY(add_wgrad(L, kdwp.p, Cp, dkv, 2 * fea, Hl, Wl, 1, dwp, G(ci + ".kv.1.bias")));
                    float** dw1 = G(ci + ".kv.1.weight");
                    L.v.push_back([=](hipStream_t st) {
                        hipLaunchKernelGGL(unpad_weight_kernel, tgrid((size_t)2 * fea * cd), dim3(256), 0, this->wg_async ? this->wg_stream : st, (const float*)T->wpad, 2 * fea, Cp, cd, 1, *dw1);
                    });
                    Tensor dkvT = kv;
                    dkvT.p = dkv;
                    dkvT.st = nullptr;
                    Tensor dkdw;  // padded to Cp channels
                    DDIF_TRY(add_dgrad(L, ci + ".kv.1", dkvT, &dkdw, true));
                    const int nsplit = std::min(512, BB * Hl);
                    need(T->n_dwpart, (size_t)nsplit * cd * 9);
                    float** dwk0 = G(ci + ".kv.0.weight");
                    const Tensor cimg = cdec[m.lev];
                    const int ldk = dkdw.C;
                    L.v.push_back([=](hipStream_t st) {
                        hipStream_t ws = this->train_fork(st);
                        hipLaunchKernelGGL(dw_wgrad_partial_nhwc_kernel, dim3((cd + 31) / 32, nsplit), dim3(256), 8 * 32 * 9 * sizeof(double), ws, (const float*)cimg.p, cd,
                                           (const float*)dkdw.p, ldk, BB, cd, Hl, Wl, nsplit, T->dwpart);
                        hipLaunchKernelGGL(dw_wgrad_reduce_kernel, dim3((cd * 9 + 7) / 8), dim3(256), 256 * sizeof(double), ws, (const double*)T->dwpart, nsplit, cd, *dwk0);
                    });
                }
                // prenorm_x: xn = GroupNorm(cat[h, skip])  (no SiLU); then split the gradient of the cat
                {
                    const int HW = Hl * Wl, nchunk = HW < 32 ? HW : 32, Ca = m.in.C, Cb = skip.C;
                    const float *gamma = V(ci + ".prenorm_x.weight"), *beta = V(ci + ".prenorm_x.bias");
                    if (!gamma || !beta) return fail(DDIF_ERR_MISSING, "training: %s.prenorm_x missing", ci.c_str());
                    need(T->n_a, npix * fea);     // cat[h, skip]
                    need(T->n_tmp, npix * fea);   // its gradient
                    need(T->n_cpart, (size_t)BB * nchunk * fea * 2);
                    float** dg = G(ci + ".prenorm_x.weight");
                    float** dbt = G(ci + ".prenorm_x.bias");
                    double *cp = nullptr, *gp = nullptr;
                    DDIF_TRY(gn_site(fea, nchunk, dg, dbt, &cp, &gp));
                    float* dskip = nullptr;
                    DDIF_TRY(fbuf(&dx, numel(m.in)));
                    DDIF_TRY(fbuf(&dskip, numel(skip)));
                    const Tensor hin = m.in;
                    if (hin.st && skip.st && Ca % 4 == 0 && Cb % 4 == 0 && !getenv("DDIF_TRAIN_GN_RESTAT")) {
                        // the cat is never formed: both sources are read in place, with the statistics their producers left, and each gets its gradient directly
                        L.v.push_back([=](hipStream_t st) {
                            tk::gn_bwd_cat(st, hin.p, Ca, skip.p, Cb, hin.st, hin.np, skip.st, skip.np, dxn, gamma, beta, BB, HW, nchunk, cp, nullptr, nullptr, nullptr, dx, dskip, gp);
                        });
                    } else
                    L.v.push_back([=](hipStream_t st) {
                        hipLaunchKernelGGL(concat2_kernel, tgrid(npix * fea), dim3(256), 0, st, (const float*)hin.p, Ca, (const float*)skip.p, Cb, npix, T->a);
                        tk::gn_stats(st, T->a, BB, (size_t)HW * fea, nchunk, T->spart);
                        tk::gn_bwd(st, T->a, dxn, nullptr, T->spart, nchunk, gamma, beta, BB, HW, fea, nchunk, 0, cp, nullptr, nullptr, nullptr, nullptr, T->tmp, gp);
                        hipLaunchKernelGGL(split2_kernel, tgrid(npix * fea), dim3(256), 0, st, (const float*)T->tmp, Ca, Cb, npix, (const float*)nullptr, (const float*)nullptr, dx, dskip);
                    });
                    if (m.skip_from < 0) return fail(DDIF_ERR_STATE, "training: decoder block without a skip source");
                    skipg[m.skip_from] = dskip;
                }
                break;
            }
        }
        g = dx;
        seqs.push_back(std::move(L));
    }

    // ---- time embedding: d(rows) -> every FeatureWiseAffine Linear, noise_level_mlp (:59-64, 241-258)
    {
        const int ns = net->nslots, in4 = 4 * inner;
        DDIF_TRY(fbuf(&T->dte, (size_t)BB * inner));
        DDIF_TRY(fbuf(&T->dh1, (size_t)BB * in4));
        DDIF_TRY(fbuf(&T->ds, (size_t)BB * in4));
        DDIF_TRY(fbuf(&T->dwall, (size_t)ns * inner));
        DDIF_TRY(fbuf(&T->dball, (size_t)ns));
        float* pe = taux;                                   // [B][inner]
        float* spre = taux + (size_t)BB * inner;            // [B][4 inner] pre-activation
        float* hid = spre + (size_t)BB * in4;               // [B][4 inner] swish
        float* te = hid + (size_t)BB * in4;                 // [B][inner]
        float **gw1 = G("noise_level_mlp.1.weight"), **gb1 = G("noise_level_mlp.1.bias"), **gw3 = G("noise_level_mlp.3.weight"), **gb3 = G("noise_level_mlp.3.bias");
        struct SlotG { float** w; float** b; int off, n; };
        auto slots = std::make_shared<std::vector<SlotG>>();
        for (auto& m : tmods)
            if (m.kind == TrainMod::RES) slots->push_back(SlotG{G(m.key + ".noise_func.noise_func.0.weight"), G(m.key + ".noise_func.noise_func.0.bias"), m.slot, m.t[0].C});
        {   // one scatter record per ResnetBlock: sized from the module list
            float* raw = nullptr;
            DDIF_TRY(fbuf(&raw, (slots->size() + 1) * sizeof(SlotScatter) / sizeof(float) + 16));
            T->slot_tab = reinterpret_cast<SlotScatter*>(raw);
        }
        Seq L;
        const float *wall = net->wall, *w3 = net->w3, *w1 = net->w1;
        float* dtb_ = dtb;
        L.v.push_back([=](hipStream_t st) {
            this->train_join(st);  // the time-bias row gradients (plane sums of every ResnetBlock) were written on the side stream
            hipLaunchKernelGGL(time_dte_kernel, dim3(BB), dim3(256), 256 * sizeof(float), st, (const float*)dtb_, wall, ns, inner, T->dte);
            tk::linear_bwd(st, te, wall, dtb_, BB, inner, ns, nullptr, T->dwall, T->dball);
            {   // the table is rebuilt only when the bound gradient tensors moved
                std::vector<SlotScatter> tab;
                for (auto& sg : *slots) tab.push_back(SlotScatter{*sg.w, *sg.b, sg.off, sg.n});
                if (T->slot_host.size() != tab.size() || memcmp(T->slot_host.data(), tab.data(), tab.size() * sizeof(SlotScatter)) != 0) {
                    hipError_t e = hipMemcpyAsync(T->slot_tab, tab.data(), tab.size() * sizeof(SlotScatter), hipMemcpyHostToDevice, st);
                    if (e == hipSuccess) e = hipStreamSynchronize(st);  // `tab` is a local: the copy must have read it (rare: first step / re-bind)
                    if (e == hipSuccess) T->slot_host = tab;          // the mirror follows the device table only when the upload succeeded
                    else T->upload_err = e;
                }
                hipLaunchKernelGGL(slot_scatter_kernel, dim3((unsigned)tab.size()), dim3(256), 0, st, (const SlotScatter*)T->slot_tab, (int)tab.size(), (const float*)T->dwall,
                                   (const float*)T->dball, inner);
            }
            tk::linear_bwd(st, hid, w3, T->dte, BB, in4, inner, T->dh1, *gw3, *gb3);
            tk::silu_bwd(st, spre, T->dh1, (size_t)BB * in4, T->ds);
            tk::linear_bwd(st, pe, w1, T->ds, BB, inner, in4, nullptr, *gw1, *gb1);
            if (!T->gn_sites.empty()) {  // dgamma / dbeta of every GroupNorm of the iteration: one launch (the table follows the bound gradient tensors, like the slot table)
                std::vector<GnRedRec> tab;
                int nblk = 0;
                for (auto& g : T->gn_sites) {
                    tab.push_back(GnRedRec{g.cpart, *g.dg, *g.dbt, g.C, g.nchunk, nblk, 0});
                    nblk += (g.C + 31) / 32;
                }
                if (T->gnred_host.size() != tab.size() || memcmp(T->gnred_host.data(), tab.data(), tab.size() * sizeof(GnRedRec)) != 0) {
                    hipError_t e = hipMemcpyAsync(T->gnred_dev, tab.data(), tab.size() * sizeof(GnRedRec), hipMemcpyHostToDevice, st);
                    if (e == hipSuccess) e = hipStreamSynchronize(st);  // `tab` is a local (rare: first step / re-bind)
                    if (e == hipSuccess) T->gnred_host = tab;
                    else T->upload_err = e;
                }
                tk::gn_bwd_reduce_all(st, T->gnred_dev, (int)tab.size(), nblk, BB);
            }
        });
        seqs.push_back(std::move(L));
    }

    // ---- scratch
    DDIF_TRY(fbuf(&T->a, T->n_a + 64));
    for (int k = 0; k < 3; ++k) DDIF_TRY(fbuf(&T->a_ring[k], T->n_a + 64));
    DDIF_TRY(fbuf(&T->tmp, T->n_tmp + 64));
    DDIF_TRY(fbuf(&T->partial, T->n_partial + 64));
    DDIF_TRY(fbuf(&T->bpart, T->n_bpart + 64));
    DDIF_TRY(fbuf(&T->wbpart, T->n_bpart + 64));
    DDIF_TRY(fbuf(&T->S, (size_t)BB * 2 + 64));
    DDIF_TRY(fbuf(&T->wpad, T->n_wpad + 64));
    DDIF_TRY(dalloc(&T->spart, (size_t)BB * 32 * 2 + 512));
    DDIF_TRY(dalloc(&T->cpart, T->n_cpart + 64));
    DDIF_TRY(dalloc(&T->dwpart, T->n_dwpart + 64));
    {
        double* raw = nullptr;
        DDIF_TRY(dalloc(&raw, (T->gn_sites.size() + 1) * sizeof(GnRedRec) / sizeof(double) + 8));
        T->gnred_dev = reinterpret_cast<GnRedRec*>(raw);
    }

    for (auto& L : seqs) {
        auto sp = std::make_shared<Seq>(std::move(L));
        bwd.push_back([sp](hipStream_t st) {
            for (auto& f : sp->v) f(st);
        });
    }
    return 0;
#undef DDIF_TRY
}

// run after the forward (Plan::train_step, ddif_plan.cpp): loss, its gradient, the reverse program
int Plan::train_backward(const float* target_nhwc, float upstream, float* loss_dev, hipStream_t s) {
    const size_t n = (size_t)B * H * W * C;
    {
        const int nblk = 256;
        hipLaunchKernelGGL(l1_partial_kernel, dim3(nblk), dim3(256), 256 * sizeof(double), s, (const float*)net_out.p, target_nhwc, n, ts->spart);
        hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(64), 0, s, (const double*)ts->spart, nblk, n, d_loss);
    }
    tk::l1_bwd(s, net_out.p, target_nhwc, n, upstream, d_net_out);
    for (auto& f : bwd) f(s);
    train_join(s);  // every weight gradient is in place
    if (loss_dev) DDIF_HIPCHK(hipMemcpyAsync(loss_dev, d_loss, sizeof(float), hipMemcpyDeviceToDevice, s));
    if (ts->upload_err != hipSuccess) {  // a gradient-routing table did not reach the device: this step's dgamma / dbeta / time-MLP gradients are not valid
        const hipError_t e = ts->upload_err;
        ts->upload_err = hipSuccess;
        return fail(DDIF_ERR_HIP, "training step: table upload failed (%s)", hipGetErrorString(e));
    }
    DDIF_HIPCHK(hipGetLastError());
    return 0;
}

void Plan::train_set_stem_source(const float* sc_nhwc) { ts->stem_sc = sc_nhwc; }

hipStream_t Plan::train_fork(hipStream_t main) {
    if (!wg_async) return main;
    (void)hipEventRecord(wg_fork, main);  // (one event serves every fork: a wait captures the record in front of it)
    dd_wait_event(wg_stream, wg_fork);
    return wg_stream;
}
void Plan::train_join(hipStream_t main) {
    if (!wg_async) return;
    (void)hipEventRecord(wg_join, wg_stream);
    dd_wait_event(main, wg_join);
}

}  // namespace ddif

// ---------------------------------------------------------------------------------------------------------------- C ABI: the NHWC linear attention core alone
static int la_check(const char* what, int B, int qd, int H, int W, int heads) {
    if (B < 1 || heads < 1 || qd < heads || qd % heads || H < 1 || W < 1) return ddif::fail(DDIF_ERR_INVALID, "%s: bad argument", what);
    const int d = qd / heads;
    if (d > 32 || d % 4) return ddif::fail(DDIF_ERR_INVALID, "%s: head dim <= 32 and 4 | head dim only", what);
    if ((H > W ? H : W) * qd > ddif::LA_TILE_MAX) return ddif::fail(DDIF_ERR_INVALID, "%s: max(H, W) * channels must not exceed %d", what, ddif::LA_TILE_MAX);
    return 0;
}
int64_t ddif_linattn_nhwc_workspace(int B, int qd, int H, int W, int heads) {
    if (la_check("ddif_linattn_nhwc_workspace", B, qd, H, W, heads)) return -1;
    const int d = qd / heads;
    return (int64_t)(2 * (size_t)B * qd * d + ddif::tk::linattn_part_floats(B, H, W, qd, d));
}
int ddif_linattn_nhwc_fwd(const float* q_pre, const float* kv_pre, int B, int qd, int H, int W, int heads, float* out, float* workspace, void* stream) {
    if (int e = la_check("ddif_linattn_nhwc_fwd", B, qd, H, W, heads)) return e;
    if (!q_pre || !kv_pre || !out || !workspace) return ddif::fail(DDIF_ERR_INVALID, "ddif_linattn_nhwc_fwd: NULL argument");
    if (int e = ddif::tk::linattn_prepare()) return e;
    const int d = qd / heads;
    float* ctx = workspace;
    float* part = workspace + 2 * (size_t)B * qd * d;
    ddif::tk::linattn_fwd((hipStream_t)stream, q_pre, kv_pre, B, heads, d, H, W, out, qd, ctx, part);
    if (hipGetLastError() != hipSuccess) return ddif::fail(DDIF_ERR_HIP, "ddif_linattn_nhwc_fwd: launch failed");
    return 0;
}
int ddif_linattn_nhwc_bwd(const float* q_pre, const float* kv_pre, const float* dout, int B, int qd, int H, int W, int heads, float* dq_pre, float* dkv_pre, float* workspace,
                          void* stream) {
    if (int e = la_check("ddif_linattn_nhwc_bwd", B, qd, H, W, heads)) return e;
    if (!q_pre || !kv_pre || !dout || !dq_pre || !dkv_pre || !workspace) return ddif::fail(DDIF_ERR_INVALID, "ddif_linattn_nhwc_bwd: NULL argument");
    if (int e = ddif::tk::linattn_prepare()) return e;
    const int d = qd / heads;
    float* ctx = workspace;
    float* dctx = workspace + (size_t)B * qd * d;
    float* part = workspace + 2 * (size_t)B * qd * d;
    ddif::tk::linattn_bwd((hipStream_t)stream, q_pre, kv_pre, dout, qd, ctx, B, heads, d, H, W, dq_pre, dkv_pre, dctx, part);
    if (hipGetLastError() != hipSuccess) return ddif::fail(DDIF_ERR_HIP, "ddif_linattn_nhwc_bwd: launch failed");
    return 0;
}
