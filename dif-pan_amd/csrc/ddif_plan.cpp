// Plan construction (launch program of UNetSR3.forward, reference models/sr3_dwt.py:169-219 with the cond-only
// branches hoisted into set_cond) and the sampling loops (reference diffusion/diffusion_ddpm_pan.py:445-507,
// 624-666; solver/dpm_solver.py:1179-1221).
#include "ddif_plan.h"
#include <algorithm>
#include "kernels_conv.h"
#include "kernels_misc.h"
#include "attn_args.h"
#include "kernels_lafuse.h"

namespace ddif {

// ------------------------------------------------------------------------------------------------ conv variants
// The instantiations live in three translation units of their own (conv_variants.h: ddif_conv_k1.cpp = 1x1 convs, ddif_conv_k3.cpp = 3x3 convs with the plain
// epilogue, ddif_conv_k3e.cpp = 3x3 convs with an epilogue variant; ddif_xf.cpp = EPI_XF; ddif_lr.cpp = the low-resolution kernel): they build in parallel, and
// the plan builder below no longer recompiles 150 kernels when its logic changes.
// vec (kernels_conv.h VEC): 0 = scalar staging (stem with C = 31, cond convs with 9 / 11 / 34 / 40 input channels; only the
// prologue-free 3x3 / 1x1 kernels), 1 = float4 staging with one source per channel chunk, 2 = float4 staging with a
// per-thread source select (stem: cat[x, x] of 8 + 8 or 4 + 4 channels inside one 16-channel chunk).
// epi: EPI_* bits (kernels_conv.h) -- FiLM (CondInjection.x_conv), scalar output path (Cout % 4 != 0), residual add.
// Only the combinations the network uses are instantiated.
ConvVariant get_conv_variant(int ks, int stride, int ups, int ck, int pro, int cfg, int vec, int epi, int math) {
    if (cfg == 20 || cfg == 21)
        return (stride == 1 && !ups && vec == 1) ? get_lr_variant(ks, cfg == 20 ? 2 : 4, pro, epi, (math == MATH_F16X2 && pro == PRO_COLSM) ? MATH_BF16X3 : math) : ConvVariant();
    if (ks == 1) return get_conv_variant_k1(stride, ups, ck, pro, cfg, vec, epi);
    if (ks == 3) return epi ? get_conv_variant_k3e(stride, ups, ck, pro, cfg, vec, epi) : get_conv_variant_k3(stride, ups, ck, pro, cfg, vec);
    return ConvVariant();
}

static int num_cus();
static int x3_enabled() {  // DDIF_X3=0: the exact-fp32 MFMA instantiation everywhere (bitwise an fmaf chain); covered by tests/test_env_switches.py
    static const int x3 = [] { const char* e = getenv("DDIF_X3"); return e ? atoi(e) : 1; }();
    return x3;
}
static int f16_enabled() {  // DDIF_F16=0: the split-operand convs stay on bf16x3 (six products) instead of f16x2 (three); tests/test_env_switches.py
    static const int f16 = [] { const char* e = getenv("DDIF_F16"); return e ? atoi(e) : 1; }();
    return f16;
}
int g_math_mode = [] { const char* e = getenv("DDIF_MATH"); return (e && std::strcmp(e, "bf16") == 0) ? 1 : 0; }();
int g_f16_raw = [] { const char* e = getenv("DDIF_F16_RAW"); return e ? (atoi(e) != 0) : 1; }();
static int lr_rows_enabled() {  // DDIF_LR_ROWS=0: the low-resolution 3x3 convs keep the general (pixel-item) staging where the row staging applies
    static const int v = [] { const char* e = getenv("DDIF_LR_ROWS"); return e ? atoi(e) : 1; }();
    return v;
}
static int lafuse_enabled() {  // DDIF_LAFUSE=0: the decoder's linear-attention half as three launches (q conv, column statistics, attn_out conv)
    static const int v = [] { const char* e = getenv("DDIF_LAFUSE"); return e ? atoi(e) : 1; }();
    return v;
}
// DDIF_XCD (bit mask, default 15 = all): XCD-contiguous work partition (ddif_dev.h wg_work_range) of 1 = the general conv kernel, 2 = the low-resolution kernel,
// 4 = the fused linear-attention block, 8 = the per-sample kernels (bottleneck attention block, gn_dw3x3 of the low levels).  The dispatcher puts workgroup b on
// XCD b % 8; with the map every kernel of the step gives XCD k the same eighth of the samples (tiles 8 k .. 8 k + 7 at B = 64), so halos, the cout tiles of a pixel
// tile and a consumer's input (written by the same XCD one launch earlier) meet in that XCD's private L2: 3.90 -> 3.76 ms per denoising step, same box
// (profiles/r05/k_xcd_ab.txt).  Results do not depend on it -- the partition only decides WHICH workgroup computes an item (tests/test_env_switches.py).
static int xcd_mask() {
    static const int v = [] { const char* e = getenv("DDIF_XCD"); return e ? atoi(e) : 15; }();
    return v;
}
static int la8_enabled() {  // DDIF_LA8=0: the decoder's linear-attention half at the 8 x 8 level as three launches (rounds 3-5) instead of the fused kernel of round 6 (kernels_lafuse8.h)
    static const int v = [] { const char* e = getenv("DDIF_LA8"); return e ? atoi(e) : 1; }();
    return v;
}
static int xf_enabled() {  // DDIF_XF=0: CondInjection.x_conv + FiLM as a launch of its own everywhere (the form of rounds 1-5); tests/test_env_switches.py
    static const int v = [] { const char* e = getenv("DDIF_XF"); return e ? atoi(e) : 1; }();
    return v;
}
static int lr_enabled() {  // DDIF_LR=0: the 8x8 / 16x16 levels on the general conv kernel (kernels_conv.h) as well; covered by tests/test_env_switches.py
    static const int lr = [] { const char* e = getenv("DDIF_LR"); return e ? atoi(e) : 1; }();
    return lr;
}
// cfg 20 / 21: the low-resolution kernel (kernels_lr.h) with 8x8 / 8x16 pixel tiles -- samples of <= 256 pixels whose
// channel counts fit its 16-channel slabs
static int pick_cfg(int ks, int ck, int pro, int vec, int stride, int ups_, int Hout, int Wout, int Cout, int B, int cin, int c0, bool allow_lr = true, bool exact = false, bool f16ok = false, bool b1 = false) {
    const bool wide = (Wout >= 16) && stride == 1;
    (void)pro;
    const bool x3 = x3_enabled() && !exact;
    if (allow_lr && x3 && lr_enabled() && vec == 1 && stride == 1 && !ups_ && Hout * Wout <= 256 && Cout % 4 == 0 && cin % 16 == 0 && c0 % 16 == 0 &&
        ck == (ks == 3 ? 16 : 32))
        // (smaller tiles -- 4 x 8 pixels at the 8 x 8 level, 8 x 8 at 16 x 16: twice the workgroups, each with half the matrix work -- measured
        //  SLOWER, 1.375 -> 1.42 / 1.49 / 1.54 ms for the class, profiles/r04/f_lr_tiles_ab.txt: the items are latency chains that also stream the
        //  weights once per item; more of them only adds weight traffic)
        return (Hout <= 8 && Wout <= 8) ? 20 : 21;
    if (ks == 1 && vec == 1) {
        const int base = wide ? (Cout > 64 ? 3 : (Cout > 32 ? 1 : 0)) : (Cout > 64 ? 4 : 2);
        // 32-cout tiles stay on the exact instruction: with 12 MFMAs per stage the split only adds staging work
        // (measured: softmax_H(q).ctx.attn_out 64+64->32 @64^2 72 vs 58 us)
        if (x3 && ck == 32 && b1) return base + 52;  // 0,1,3,2,4 -> 52,53,55,54,56
        if (x3 && ck == 32 && base != 0) return base + 12;  // 1,3,2,4 -> 13,15,14,16
        return base;
    }
    if (ks == 3 && vec == 1 && stride == 1 && x3) {
        const bool f16 = f16ok && (pro == PRO_NONE || pro == PRO_GN_SILU);
        // 16 x 16-pixel tiles on eight waves where they fill the CUs; below that (8-16 tiles per GPU: what one rank of a strong-scaling run holds) the four-wave
        // 8 x 16 tiling gives twice the workgroups and every wave a SIMD to itself.  Results do not depend on the choice bit for bit: same products per pixel, and
        // the 16 x 16 tilings write their statistics partials per 8 x 16 half tile (ConvArgs::st_halves).  DDIF_TILE16=1 / 0 forces one form.
        static const int tile16_env = [] { const char* e = getenv("DDIF_TILE16"); return e ? atoi(e) : -1; }();
        const long items16 = (long)B * ((Hout + 15) / 16) * ((Wout + 15) / 16) * ((Cout + 31) / 32);
        const bool big = tile16_env >= 0 ? tile16_env != 0 : items16 >= num_cus();
        if (wide && Hout >= 32 && Wout >= 32 && big) return b1 ? 47 : (f16 ? 27 : 7);
        if (wide || Cout <= 32) return b1 ? 48 : (f16 ? 28 : 8);
        return b1 ? 49 : (f16 ? 29 : 9);
    }
    static const bool s2_f16 = [] { const char* e = getenv("DDIF_S2_F16"); return !e || atoi(e) != 0; }();  // DDIF_S2_F16=0: the Downsample convs and the stem stay on the exact-fp32 tilings
    if (ks == 3 && vec == 1 && stride == 2 && !ups_ && x3 && f16ok && !b1 && s2_f16 && pro == PRO_NONE && Wout >= 16) return Cout <= 32 ? 28 : 29;
    if (ks == 3 && vec == 2 && stride == 1 && !ups_ && x3 && f16ok && !b1 && s2_f16 && pro == PRO_NONE && wide && Cout <= 32) {  // the stem
        static const int tile16_env = [] { const char* e = getenv("DDIF_TILE16"); return e ? atoi(e) : -1; }();
        const long items16 = (long)B * ((Hout + 15) / 16) * ((Wout + 15) / 16);
        const bool big = tile16_env >= 0 ? tile16_env != 0 : items16 >= num_cus();
        return (Hout >= 32 && Wout >= 32 && big) ? 27 : 28;
    }
    if (ks == 3 && vec == 1 && wide && !ups_) {
        const long items32 = (long)B * ((Hout + 15) / 16) * ((Wout + 15) / 16) * ((Cout + 31) / 32);
        if (Hout >= 32 && Wout >= 32 && items32 >= 2L * num_cus()) return (Cout % 64 == 0 && items32 >= 4L * num_cus()) ? 6 : 5;
    }
    if (wide) return 0;
    if (Cout <= 32 && stride == 1) return 0;
    return 2;
}

static int num_cus() {  // of the CURRENT device (plan entry points make the net's device current), cached per device
    static int cache[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cache[dev] == 0) {
        int v = 0;
        cache[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    return cache[dev];
}
// Test hook (ddif_debug_set_grid_cap, include/ddif.h): caps the persistent grid of every conv launch of plans built
// afterwards, so that small parity cases walk several work items per workgroup across sample boundaries -- the
// regime the B=64 benchmark runs in.  0 = no cap.
int g_debug_grid_cap = 0;
static int wg_per_cu(size_t smem, int cap = 2) {
    // (3 or 4 persistent workgroups per CU measured slower for the 1x1 class of kernels_conv.h: 1.56 vs 1.46 ms per step)
    int byl = (int)((160 * 1024) / (smem ? smem : 1));
    if (byl < 1) byl = 1;
    return cap < byl ? cap : byl;
}

static inline dim3 ew_grid(size_t n) {
    size_t g = (n + 255) / 256;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return dim3((unsigned)g);
}

// depthwise 3x3 launch: the channel-quad kernel whenever both sources have 4 | channels (every feature tensor), the scalar one for the 11-channel cond images
static inline void launch_dw3x3(hipStream_t s, const DwArgs& a) {
    const dim3 grid(a.B * a.tiles_x * a.tiles_y);
    if (a.c0 % 4 == 0 && a.c1 % 4 == 0) hipLaunchKernelGGL(dw3x3_q4_kernel, grid, dim3(256), 10 * 18 * 32 * sizeof(float), s, a);
    else hipLaunchKernelGGL(dw3x3_kernel, grid, dim3(256), 10 * 18 * 32 * sizeof(float), s, a);
}


// ------------------------------------------------------------------------------------------------ allocation
Plan::~Plan() {
    drop_graphs();
#ifndef DDIF_EMU
    if (cap_stream) (void)hipStreamDestroy(cap_stream);
#endif
#ifndef DDIF_EMU
    if (wg_stream) {
        (void)hipStreamSynchronize(wg_stream);
        (void)hipStreamDestroy(wg_stream);
    }
    if (side_read) {
        auto& ev = net->reader_events;
        ev.erase(std::remove(ev.begin(), ev.end(), side_read), ev.end());
        (void)hipEventDestroy(side_read);
    }
    if (wg_fork) (void)hipEventDestroy(wg_fork);
    if (wg_join) (void)hipEventDestroy(wg_join);
    for (auto e : a_free)
        if (e) (void)hipEventDestroy(e);
#endif
    for (void* p : allocs) (void)hipFree(p);
    for (auto e : ev0) (void)hipEventDestroy(e);
    for (auto e : ev1) (void)hipEventDestroy(e);
}

template <typename T>
int Plan::dalloc(T** p, size_t n) {
    void* q = nullptr;
    const size_t bytes = (n * sizeof(T) + 255) & ~(size_t)255;
    if (dry) {  // fake, never dereferenced
        *p = reinterpret_cast<T*>((uintptr_t)0x100000000000ull + dry_next);
        dry_next += bytes;
        return 0;
    }
    DDIF_HIPCHK(hipMalloc(&q, bytes));
    allocs.push_back(q);
    bytes_allocated += bytes;
    *p = reinterpret_cast<T*>(q);
    return 0;
}

int Plan::alloc_tensor(Tensor* t, int C_, int H_, int W_, bool step_act) {
    t->C = C_;
    t->H = H_;
    t->W = W_;
    t->st = nullptr;
    t->np = 0;
    if (!step_act || train_mode) return dalloc(&t->p, (size_t)B * H_ * W_ * C_);  // train: every activation lives until the reverse pass
    // activation of the step program: lives in the arena between its first and its last launch
    const size_t bytes = ((size_t)B * H_ * W_ * C_ * sizeof(float) + 255) & ~(size_t)255;
    if (dry) {
        if (int e = dalloc(&t->p, (size_t)B * H_ * W_ * C_)) return e;
        fake2id[t->p] = (int)lives.size();
        lives.push_back(Live{bytes, (int)step.size(), (int)step.size(), 0});
        return 0;
    }
    if (arena_next >= (int)lives.size() || lives[arena_next].bytes != bytes) return fail(DDIF_ERR_STATE, "plan arena: the two build passes disagree");
    t->p = reinterpret_cast<float*>(arena + lives[arena_next++].off);
    return 0;
}

void Plan::use(const void* p) {
    if (!dry || !p) return;
    auto it = fake2id.find(p);
    if (it == fake2id.end()) return;
    Live& l = lives[it->second];
    const int idx = (int)step.size();  // index of the launch being created
    if (idx > l.last) l.last = idx;
    if (idx < l.first) l.first = idx;
}

// Two passes over the same builder: the dry pass only records shapes and liveness, the real pass allocates.
int Plan::build() {
    math_mode = g_math_mode;
    f16_raw = g_f16_raw;
    final_fused = false;
    n_range_convs = 0;
    if (!d_range) {
        void* q = nullptr;
        DDIF_HIPCHK(hipMalloc(&q, 64));
        DDIF_HIPCHK(hipMemset(q, 0, 64));
        allocs.push_back(q);
        bytes_allocated += 64;
        d_range = reinterpret_cast<int*>(q);
    }
    if (train_mode) {  // no activation arena: the reverse pass reads the forward's tensors
        dry = false;
        tmods.clear();
        if (int e = net->build_dgrad_packs()) return e;
        if (int e = build_impl()) return e;
        return build_backward();
    }
    dry = true;
    if (int e = build_impl()) return e;
    // the network output is read by the sampler update / layout conversion AFTER the last launch of the program
    {
        auto it = fake2id.find(net_out.p);
        if (it != fake2id.end()) lives[it->second].last = 1 << 30;
    }
    // first-fit interval colouring in order of first use
    std::vector<int> order(lives.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return lives[x].first < lives[y].first; });
    std::vector<int> placed;
    arena_bytes = 0;
    unaliased_bytes = 0;
    // DDIF_ARENA_PAD=<bytes> (development aid): every tensor reserves that much more, which shifts the relative placement of the tensors of a launch -- the probe behind
    // the "one ffn.0 instance of four takes 68-72 us instead of 47" question of rounds 5-6 (profiles/r06/arena_pad_probe.txt)
    static const size_t arena_pad = [] { const char* e = getenv("DDIF_ARENA_PAD"); return e ? (size_t)atoll(e) & ~(size_t)255 : (size_t)0; }();
    for (int id : order) {
        Live& l = lives[id];
        unaliased_bytes += l.bytes;
        const size_t need = l.bytes + arena_pad;
        std::vector<std::pair<size_t, size_t>> busy;  // address ranges of tensors alive at the same time
        for (int o : placed)
            if (!(lives[o].last < l.first || l.last < lives[o].first)) busy.emplace_back(lives[o].off, lives[o].off + lives[o].bytes + arena_pad);
        std::sort(busy.begin(), busy.end());
        size_t off = 0;
        for (auto& r : busy) {
            if (off + need <= r.first) break;
            if (r.second > off) off = r.second;
        }
        l.off = off;
        if (off + l.bytes > arena_bytes) arena_bytes = off + l.bytes;
        placed.push_back(id);
    }
    // reset everything the dry pass produced, then build for real
    dry = false;
    dry_next = 0;
    fake2id.clear();
    pre.clear();
    step.clear();
    cenc.clear();
    cdec.clear();
    drop_sites.clear();
    path_sites.clear();
    mask_recs = nullptr;
    n_mask_recs = 0;
    n_conv3 = n_conv3_x3 = n_conv3_f16 = n_conv3_b1 = 0;
    n_range_convs = 0;
    final_fused = false;
    tb_rows = 0;
    tb = tvals = nullptr;
    arena_next = 0;
    if (arena_bytes) {
        void* q = nullptr;
        DDIF_HIPCHK(hipMalloc(&q, arena_bytes));
        allocs.push_back(q);
        bytes_allocated += arena_bytes;
        arena = reinterpret_cast<char*>(q);
    }
    if (int e = build_impl()) return e;
    if (arena_next != (int)lives.size()) return fail(DDIF_ERR_STATE, "plan arena: the two build passes disagree");
    return 0;
}

int Plan::add_conv(std::vector<Op>& prog, const ConvSpec& s, Tensor* out) {
    const PackedConv& pc = *s.pc;
    const int Hin = s.in0.H, Win = s.in0.W;
    const int Hc = s.ups ? 2 * Hin : Hin, Wc = s.ups ? 2 * Win : Win;
    const int Hout = s.stride == 2 ? (Hc - 1) / 2 + 1 : Hc, Wout = s.stride == 2 ? (Wc - 1) / 2 + 1 : Wc;
    const int c0 = s.in0.C, c1 = s.in1.C;
    if (c0 + c1 != pc.cin) return fail(DDIF_ERR_INVALID, "%s: input channels %d+%d != weight cin %d", s.name, c0, c1, pc.cin);
    if (pc.ks == 1 && (s.stride != 1 || s.ups)) return fail(DDIF_ERR_INVALID, "%s: 1x1 conv with stride/upsample", s.name);
    const int vec = (c0 % 4 != 0 || c1 % 4 != 0) ? 0 : ((c1 == 0 || c0 % pc.ck == 0) ? 1 : 2);
    if ((size_t)B * Hin * Win * (c0 > c1 ? c0 : c1) * 4 >= ((size_t)1 << 32) || (size_t)B * Hout * Wout * pc.cout * 8 >= ((size_t)1 << 32))
        return fail(DDIF_ERR_INVALID, "%s: a tensor of this batch reaches 4 GiB (32-bit offsets); split the batch", s.name);
    // f16x2 (kernels_conv.h MATH = 3): inference plans only (the half packs are not refreshed on the device), shared weights inside the
    // scaled half range (pc.w_f16), and for a GroupNorm prologue an output bound sqrt(N) max|gamma| + max|beta| inside the activation range
    bool f16ok = f16_enabled() && !train_mode && pc.w_f16 && !s.w_override && !s.exact;
    const bool raw_in = !(s.pro == PRO_GN || s.pro == PRO_GN_SILU || s.pro == PRO_GN_DW);  // nothing bounds the staged values: f16x2 only under the plan's range watch
    if (raw_in && !f16_raw) f16ok = false;
    if (f16ok && (s.pro == PRO_GN || s.pro == PRO_GN_SILU)) {
        auto ig = net->vec_absmax.find(s.gamma), ib = net->vec_absmax.find(s.beta);
        const double n = (double)(c0 + c1) * Hin * Win;
        f16ok = ig != net->vec_absmax.end() && ib != net->vec_absmax.end() && std::sqrt(n) * ig->second + ib->second < DDIF_F16_AMAX;
    }
    // bf16x1 (MATH = 4): the throughput variant, only under ddif_set_math_mode(DDIF_MATH_BF16); inference plans, shared weights
    const bool b1ok = math_mode == 1 && !train_mode && pc.w_b1 && !s.w_override && !s.exact;
    const bool f16ok0 = f16ok;
    if (b1ok) f16ok = false;
    int math = b1ok ? MATH_BF16X1 : (f16ok ? MATH_F16X2 : MATH_BF16X3);
    int cfg = pick_cfg(pc.ks, pc.ck, s.pro, vec, s.stride, s.ups, Hout, Wout, pc.cout, B, c0 + c1, c1 ? c0 : c0 + c1, true, s.exact, f16ok, b1ok);
    const int epi = (s.film ? EPI_FILM : 0) | (s.res ? EPI_RES : 0) | (pc.cout % 4 != 0 ? EPI_SOUT : 0) | (s.silu ? EPI_SILU : 0) | (s.cso_mx ? EPI_COLST : 0);
    if (s.cso_mx && ((cfg != 20 && cfg != 21) || Hout > (cfg == 20 ? 8 : 16)))
        return fail(DDIF_ERR_INVALID, "%s: column statistics epilogue needs the low-resolution kernel and H <= 16", s.name);
    static const bool wres_env = [] { const char* e = getenv("DDIF_WRES"); return !e || atoi(e) != 0; }();  // DDIF_WRES=0: tiling 27 everywhere (tests/test_env_switches.py)
    if (cfg == 27 && wres_env && c0 == 32 && c1 == 0 && pc.cout <= 32 && pc.ck == 16 && pc.n_chunks == 2 && !s.w_override && s.w_bstride == 0 &&
        get_conv_variant(pc.ks, s.stride, s.ups, pc.ck, s.pro, 37, vec, epi, math).fn)
        cfg = 37;
    ConvVariant var = get_conv_variant(pc.ks, s.stride, s.ups, pc.ck, s.pro, cfg, vec, epi, math);
    if (!var.fn && (cfg == 20 || cfg == 21)) {  // prologue / epilogue combination the low-resolution kernel does not carry
        cfg = pick_cfg(pc.ks, pc.ck, s.pro, vec, s.stride, s.ups, Hout, Wout, pc.cout, B, c0 + c1, c1 ? c0 : c0 + c1, false, false, f16ok, b1ok);
        var = get_conv_variant(pc.ks, s.stride, s.ups, pc.ck, s.pro, cfg, vec, epi);
    }
    if (!var.fn && b1ok) {  // a combination the throughput variant does not instantiate: the default path of this conv
        f16ok = f16ok0;
        math = f16ok ? MATH_F16X2 : MATH_BF16X3;
        cfg = pick_cfg(pc.ks, pc.ck, s.pro, vec, s.stride, s.ups, Hout, Wout, pc.cout, B, c0 + c1, c1 ? c0 : c0 + c1, true, s.exact, f16ok, false);
        var = get_conv_variant(pc.ks, s.stride, s.ups, pc.ck, s.pro, cfg, vec, epi, math);
        if (!var.fn && (cfg == 20 || cfg == 21)) {
            cfg = pick_cfg(pc.ks, pc.ck, s.pro, vec, s.stride, s.ups, Hout, Wout, pc.cout, B, c0 + c1, c1 ? c0 : c0 + c1, false, false, f16ok, false);
            var = get_conv_variant(pc.ks, s.stride, s.ups, pc.ck, s.pro, cfg, vec, epi);
        }
    }
    if (!var.fn) return fail(DDIF_ERR_INVALID, "%s: no kernel variant (ks=%d stride=%d ups=%d ck=%d pro=%d cfg=%d vec=%d epi=%d)", s.name, pc.ks, s.stride, s.ups, pc.ck, s.pro, cfg, vec, epi);
    // round 6: a low-resolution 3x3 conv whose tile spans the whole image width (the 8x8 / 16x16 levels of a 64x64 tile) stages by rows (kernels_lr.h ROWS): same
    // results, ~1.3 us less instruction issue per launch.  DDIF_LR_ROWS=0: the general staging everywhere (tests/test_env_switches.py)
    bool lr_rows = false;
    if (var.lr && !var.b1 && pc.ks == 3 && Win == var.tw && Wout == Win && Hout == Hin && lr_rows_enabled()) {
        const ConvVariant vr = get_lr_variant(3, cfg == 20 ? 2 : 4, s.pro, epi, var.f16 ? MATH_F16X2 : MATH_BF16X3, true);
        if (vr.fn && vr.smem == var.smem) {
            var = vr;
            lr_rows = true;
        }
    }
    // round 6: the next block's x_conv + FiLM as a second output of this conv (kernels_conv.h EPI_XF) -- where the chosen f16x2 tiling has such an instantiation
    // and one wave holds all 32 couts of its pixels; otherwise the request is left undone and the caller emits the 1x1 launch as before
    bool xf = false;
    if (s.xf && !train_mode && xf_enabled() && var.f16 && !var.lr && pc.ks == 3 && pc.cout == 32 && !s.ups && s.tb_off < 0 && !s.samp && s.xf->pc && s.xf->w && s.xf->film &&
        s.xf->pc->cin == 32 && s.xf->pc->bias && (s.xf->pc->cout == 32 || s.xf->pc->cout == 64) && &prog == &step) {
        const ConvVariant vx = get_xf_variant(s.stride, s.pro, cfg, vec, epi, s.xf->pc->cout / 32);
        if (vx.fn && vx.smem == var.smem) {
            var = vx;
            xf = true;
        }
    }
    if ((s.pro == PRO_GN || s.pro == PRO_GN_SILU || s.pro == PRO_GN_DW) && (!s.in0.st || (s.in1.p && !s.in1.st) || !s.gamma || !s.beta))
        return fail(DDIF_ERR_STATE, "%s: GroupNorm prologue without producer statistics", s.name);
    if (&prog == &step) {  // liveness of everything this launch touches (dry pass)
        use(s.in0.p);
        use(s.in1.p);
        use(s.res);
        use(s.film);
        use(s.cs_mx);
        use(s.cs_sm);
        use(s.out_xn);
        use(s.cso_mx);
        use(s.cso_sm);
    }
    if (int e = alloc_tensor(out, pc.cout, Hout, Wout, &prog == &step)) return e;
    if (xf) {
        if (int e = alloc_tensor(&s.xf->out, s.xf->pc->cout, Hout, Wout, true)) return e;
    }
    ConvArgs a{};
    a.in0 = s.in0.p;
    a.in1 = s.in1.p;
    a.c0 = c0;
    a.c1 = c1;
    a.B = B;
    a.Hin = Hin;
    a.Win = Win;
    a.Hout = Hout;
    a.Wout = Wout;
    a.Cout = pc.cout;
    a.w = s.w_override ? s.w_override : (var.b1 ? pc.w_b1 : (var.f16 ? pc.w_f16 : (var.x3 ? pc.w_x3 : pc.w)));
    if (var.x3 && !s.w_override && !a.w) return fail(DDIF_ERR_STATE, "%s: split-operand variant without split weights", s.name);
    a.w_bstride = s.w_bstride;
    a.cs_mx = s.cs_mx;
    a.cs_sm = s.cs_sm;
    a.dw_w = s.dw_w;
    a.out_xn = s.out_xn;
    a.cso_mx = s.cso_mx;
    a.cso_sm = s.cso_sm;
    if (s.pro == PRO_GN_DW && (!s.dw_w || c0 + c1 > 256)) return fail(DDIF_ERR_INVALID, "%s: depthwise staging needs weights and <= 256 channels", s.name);
    if (s.pro == PRO_COLSM && (!s.cs_mx || !s.cs_sm || c0 % pc.ck != 0)) return fail(DDIF_ERR_INVALID, "%s: column-softmax prologue needs statistics and c0 %% %d == 0", s.name, pc.ck);
    a.n_chunks = var.wr ? 1 : pc.n_chunks;  // (resident weights: the whole input is one 32-channel stage)
    a.bias = (s.use_bias && pc.bias) ? pc.bias : zeros;
    a.tbias = zeros;  // strides 0: a row of zeros for every sample and step
    a.st0 = s.in0.st;
    a.np0 = s.in0.np;
    a.st1 = s.in1.st;
    a.np1 = s.in1.np;
    a.gamma = s.gamma;
    a.beta = s.beta;
    a.res = s.res;
    a.film = s.film;
    a.out = out->p;
    a.tiles_x = (Wout + var.tw - 1) / var.tw;
    a.tiles_y = (Hout + var.th - 1) / var.th;
    const int gy = (pc.cout + var.nt - 1) / var.nt;
    // the eight-wave 16 x 16 tilings write their statistics partials per 8 x 16 half tile, exactly as the four-wave 8 x 16 tiling of the same conv would (kernels_conv.h
    // ConvArgs::st_halves): the tiling may then depend on the number of work items without changing a single bit of what the consumer reads
    const bool st_halves = !var.lr && var.th == 16 && var.tw == 16 && var.nthr == 512;
    a.st_halves = st_halves ? 1 : 0;
    a.tiles_y8 = (Hout + 7) / 8;
    const int np_out = st_halves ? a.tiles_x * a.tiles_y8 : a.tiles_x * a.tiles_y;
    if (s.stats) {
        out->np = np_out * gy;
        if (int e = dalloc(&out->st, (size_t)B * out->np * 2)) return e;
        a.st_out = out->st;
    }
    a.n_ct = gy;
    if (xf) {
        if (gy != 1 || !s.stats) return fail(DDIF_ERR_STATE, "%s: x_conv fold needs one cout tile and output statistics", s.name);
        s.xf->out.np = np_out;
        if (int e = dalloc(&s.xf->out.st, (size_t)B * s.xf->out.np * 2)) return e;
        a.xf_w = s.xf->w;
        a.xf_b = s.xf->pc->bias;
        a.xf_film = s.xf->film;
        a.xf_out = s.xf->out.p;
        a.xf_st = s.xf->out.st;
        a.xf_cout = s.xf->pc->cout;
        s.xf->done = true;
    }
    a.xcd = (xcd_mask() & (var.lr ? 2 : 1)) ? 1 : 0;
    if (var.f16 && raw_in) {  // (the kernels only look at it in their raw-input f16x2 instantiations)
        a.range_flag = d_range;
        ++n_range_convs;
    }
    // persistent launch: a few workgroups per CU, each walking a contiguous range of (cout tile, pixel tile) items
    const long items_per_sample = (long)a.tiles_x * a.tiles_y * gy;
    const long nwork = (long)B * items_per_sample;
    const int gy0 = (pc.cout + var.nt - 1) / var.nt;
    const size_t smem = var.lr ? var.smem : var.smem + conv_smem_extra(s.pro, var.wr ? 1 : pc.n_chunks, var.wr ? 32 : pc.ck, gy0 * var.nt, xf ? s.xf->pc->cout : 0);
    long cap = (long)num_cus() * wg_per_cu(smem, var.wg_cap);
    if (g_debug_grid_cap > 0 && g_debug_grid_cap < cap) cap = g_debug_grid_cap;
    const dim3 grid((unsigned)(nwork < cap ? nwork : cap), 1u);
    const dim3 block((unsigned)var.nthr);
    if (var.smem + 8192 > 64 * 1024) {  // the attribute is per kernel function: set it to the variant's maximum
        DDIF_HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(var.fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(var.smem + 8192)));
    }
    if (smem > var.smem + 8192) return fail(DDIF_ERR_INVALID, "%s: %d input channels exceed the GroupNorm staging area", s.name, c0 + c1);
    // convs with a time bias: in the samplers every sample shares the step's row (bias + row live in LDS); forward() /
    // q_sample_forward with one t per sample use the EPI_TBS instantiation (rows loaded per work item)
    ConvKernelFn fn_tbs = nullptr;
    if (s.tb_off >= 0) {
        const ConvVariant vt = get_conv_variant(pc.ks, s.stride, s.ups, pc.ck, s.pro, cfg, vec, epi | EPI_TBS, math);
        if (!vt.fn || vt.smem != var.smem) return fail(DDIF_ERR_INVALID, "%s: no per-sample time-bias kernel variant", s.name);
        fn_tbs = lr_rows ? var.fn : vt.fn;  // (the low-resolution kernel reads its time-bias rows from memory either way: ddif_lr.cpp)
        if (var.smem + 8192 > 64 * 1024)
            DDIF_HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn_tbs), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(var.smem + 8192)));
    }
    // the final conv: the same tiling with the DDPM / DDIM update in its epilogue (split-operand tilings, vector output path)
    ConvKernelFn fn_samp = nullptr;
    if (s.samp && !train_mode && epi == 0 && s.tb_off < 0 && &prog == &step) {
        const ConvVariant vs = get_conv_variant(pc.ks, s.stride, s.ups, pc.ck, s.pro, cfg, vec, EPI_SAMP, math);
        if (vs.fn && vs.smem == var.smem) {
            fn_samp = vs.fn;
            if (var.smem + 8192 > 64 * 1024)
                DDIF_HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn_samp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(var.smem + 8192)));
            final_fused = true;
        }
    }
    const float* lms_p = lms.p;
    const bool dyn = s.dyn_input;
    const bool self_c = net->cfg.self_condition != 0;
    const int tb_off = s.tb_off;
    static const bool dump = getenv("DDIF_DUMP_PLAN") != nullptr;
    if (dump)
        fprintf(stderr, "[ddif plan] %-34s %-24s ks=%d s=%d u=%d  %3d+%3d -> %3d  @%3dx%-3d pro=%d epi=%d cfg=%d items=%ld grid=%u smem=%zu  in@%p out@%p\n", s.name, var.name, pc.ks, s.stride,
                s.ups, c0, c1, pc.cout, Hout, Wout, s.pro, epi, cfg, nwork, grid.x, smem, (const void*)s.in0.p, (const void*)out->p);
    Op op;
    op.name = var.name;
    {
        char lb[160];
        snprintf(lb, sizeof lb, "%s %dx%d %d+%d->%d @%dx%d cfg%d", s.name, pc.ks, pc.ks, c0, c1, pc.cout, Hout, Wout, cfg);
        op.label = lb;
    }
    op.flop = 2.0 * B * Hout * Wout * (double)pc.cout * (c0 + c1) * pc.ks * pc.ks;
    op.bytes = 4.0 * B * ((double)Hin * Win * (c0 + c1) + (double)Hout * Wout * pc.cout);
    if (xf) {  // + the folded 1x1 conv and its FiLM operands / output
        op.label += " + x_conv+FiLM";
        op.flop += 2.0 * B * Hout * Wout * 32.0 * s.xf->pc->cout;
        op.bytes += 4.0 * B * (double)Hout * Wout * 3.0 * s.xf->pc->cout;
    }
    op.cls = (Hout * Wout <= 256) ? 2 : (pc.ks == 3 ? 0 : 1);
    op.mfma_w = var.b1 ? 1 : (var.f16 ? 3 : (var.x3 ? 6 : 16));
    op.timed = op.cls == 0;
    if (pc.ks == 3 && &prog == &step) {
        ++n_conv3;
        if (var.x3) ++n_conv3_x3;
        if (var.f16) ++n_conv3_f16;
        if (var.b1) ++n_conv3_b1;
    }
    op.run = [a, var, fn_tbs, fn_samp, lms_p, grid, block, smem, dyn, self_c, tb_off](hipStream_t st, const StepCtx& ctx) {
        ConvArgs aa = a;
        const dim3 g = grid;
        if (dyn) {
            if (self_c) {
                aa.in0 = ctx.sc;
                aa.in1 = ctx.x;
            } else {
                aa.in0 = ctx.x;
            }
        }
        if (tb_off >= 0) {
            aa.tbias = ctx.tb + tb_off;
            aa.tbias_stride = ctx.tb_stride;
            aa.step_ptr = ctx.step_ptr;
            aa.tb_rowstride = ctx.tb_rowstride;
        }
        if (fn_samp && ctx.samp_out) {  // DDPM / DDIM loop: x_{t-1} straight from the epilogue
            aa.s_img = ctx.x;
            aa.s_lms = lms_p;
            aa.s_out = ctx.samp_out;
            aa.s_run = reinterpret_cast<const SamplerRun*>(ctx.samp_run);
            aa.s_step_next = ctx.step_next;
            aa.s_kind = ctx.samp_kind;
            aa.step_ptr = ctx.step_ptr;
            hipLaunchKernelGGL(fn_samp, g, block, smem, st, aa);
            return;
        }
        hipLaunchKernelGGL((tb_off >= 0 && ctx.tb_stride != 0) ? fn_tbs : var.fn, g, block, smem, st, aa);
    };
    prog.push_back(std::move(op));
    return 0;
}

// ------------------------------------------------------------------------------------------------ program
int Plan::build_impl() {
    const ddif_net_cfg& c = net->cfg;
    if (!net->committed) return fail(DDIF_ERR_STATE, "ddif_plan_create: ddif_net_commit has not been called");
    C = c.out_channel;
    P = c.pan_channel;
    const int Cl = c.lms_channel;
    CC = 2 * Cl + 4 * P;
    const int nlev = c.n_channel_mults;
    const int div = 1 << (nlev - 1);
    if (B < 1 || H < div || W < div || H % div || W % div)
        return fail(DDIF_ERR_INVALID, "ddif_plan_create: H=%d W=%d must be positive multiples of %d", H, W, div);
    LH.assign(1, H);
    LW.assign(1, W);
    for (int l = 1; l < nlev; ++l) {
        LH.push_back((LH.back() - 1) / 2 + 1);
        LW.push_back((LW.back() - 1) / 2 + 1);
    }
    auto V = [&](const std::string& k) -> const float* {
        auto it = net->vec.find(k);
        return it == net->vec.end() ? nullptr : it->second;
    };
    auto PC = [&](const std::string& k) -> const PackedConv* {
        auto it = net->conv.find(k);
        return it == net->conv.end() ? nullptr : &it->second;
    };
#define DDIF_TRY(x) do { if (int e__ = (x)) return e__; } while (0)

    // ---- boundary staging + sampler state
    DDIF_TRY(dalloc(&zeros, (size_t)1024));
    if (!dry) DDIF_HIPCHK(hipMemset(zeros, 0, 1024 * sizeof(float)));
    DDIF_TRY(alloc_tensor(&x_in, c.in_channel, H, W));
    DDIF_TRY(alloc_tensor(&sc_in, c.out_channel, H, W));
    DDIF_TRY(alloc_tensor(&lms, C, H, W));
    const size_t img_n = (size_t)B * H * W * C;
    for (int i = 0; i < 2; ++i) DDIF_TRY(dalloc(&img[i], img_n));
    for (int i = 0; i < 3; ++i) DDIF_TRY(dalloc(&mbuf[i], img_n));
    DDIF_TRY(dalloc(&io_nchw, (size_t)B * H * W * (c.in_channel > C ? c.in_channel : C)));
    DDIF_TRY(dalloc(&small, (size_t)4 * B + 64));

    // ---- cond-only program (set_cond)
    cenc.resize(nlev);
    cdec.resize(nlev);
    {
        Op op;
        op.name = "lms_nhwc";
        Tensor lm = lms;
        const int BB = B, CCc = CC, HW = H * W, Cc = C;
        op.run = [this, lm, BB, CCc, HW, Cc](hipStream_t s, const StepCtx&) {
            hipLaunchKernelGGL(nchw_to_nhwc_kernel, ew_grid((size_t)BB * HW * Cc), dim3(256), 0, s, (const float*)this->cond_nchw, BB, CCc, HW, 0, Cc, lm.p);
        };
        pre.push_back(std::move(op));
    }
    for (int l = 0; l < nlev; ++l) {
        DDIF_TRY(alloc_tensor(&cenc[l], Cl + P, LH[l], LW[l]));
        DDIF_TRY(alloc_tensor(&cdec[l], Cl + 3 * P, LH[l], LW[l]));
        for (int which = 0; which < 2; ++which) {
            Tensor t = which ? cdec[l] : cenc[l];
            const int cbeg = which ? CC - (Cl + 3 * P) : 0;
            Op op;
            op.name = "cond_resize";
            const int BB = B, CCc = CC, HH = H, WW = W;
            op.bytes = 4.0 * B * t.C * ((double)H * W + (double)t.H * t.W);
            op.run = [this, t, cbeg, BB, CCc, HH, WW](hipStream_t s, const StepCtx&) {
                hipLaunchKernelGGL(resize_bilinear_kernel, ew_grid((size_t)BB * t.H * t.W * t.C), dim3(256), 0, s,
                                   (const float*)this->cond_nchw, BB, CCc, HH, WW, cbeg, t.C, t.H, t.W, t.p);
            };
            pre.push_back(std::move(op));
        }
    }

    // helpers shared by the step program
    auto resblock = [&](const std::string& rb, Tensor in, Tensor* out, XfReq* xf = nullptr) -> int {
        const PackedConv *c1 = PC(rb + ".block1.block.3"), *c2 = PC(rb + ".block2.block.3");
        if (!c1 || !c2) return fail(DDIF_ERR_MISSING, "%s: conv weights missing", rb.c_str());
        Tensor h1;
        if (train_mode) {
            // train mode: y = silu(GN(h1)) * dropout mask is materialised (it is also what wgrad needs), conv2 runs on it
            auto dropped = [&](Tensor x, const float* g, const float* bt, Tensor* y) -> int {
                if (!x.st || !g || !bt) return fail(DDIF_ERR_STATE, "%s: GroupNorm without producer statistics / affine", rb.c_str());
                use(x.p);
                DDIF_TRY(alloc_tensor(y, x.C, x.H, x.W, true));
                DropSite site{nullptr, x.C, x.H, x.W};
                DDIF_TRY(dalloc(&site.mask, (size_t)B * x.H * x.W * x.C));
                drop_sites.push_back(site);
                const int HW = x.H * x.W, Cc = x.C, BB = B;
                const int chunks = (HW * Cc / 4 + 256 * 8 - 1) / (256 * 8);
                Tensor yy = *y;
                Op op;
                op.name = "gn_silu_dropout";
                op.cls = (HW <= 256) ? 2 : 5;
                op.bytes = 12.0 * B * HW * Cc;
                op.run = [x, g, bt, site, yy, HW, Cc, BB, chunks](hipStream_t s, const StepCtx&) {
                    hipLaunchKernelGGL(gn_silu_drop_kernel, dim3(chunks < 1 ? 1 : chunks, BB), dim3(256), 0, s, (const float*)x.p, (const double*)x.st, x.np, g, bt,
                                       (const float*)site.mask, HW, Cc, yy.p);
                };
                step.push_back(std::move(op));
                return 0;
            };
            // Dropout sits in block2 only (ResnetBlock.__init__, sr3_dwt.py:318-319): block1 keeps the fused GroupNorm + SiLU prologue
            Tensor y2;
            ConvSpec s1;
            s1.pc = c1;
            s1.in0 = in;
            s1.pro = PRO_GN_SILU;
            s1.gamma = V(rb + ".block1.block.0.weight");
            s1.beta = V(rb + ".block1.block.0.bias");
            s1.tb_off = net->slot_off.at(rb);
            s1.stats = true;
            s1.name = "res.conv1";
            DDIF_TRY(add_conv(step, s1, &h1));
            DDIF_TRY(dropped(h1, V(rb + ".block2.block.0.weight"), V(rb + ".block2.block.0.bias"), &y2));
            ConvSpec s2;
            s2.pc = c2;
            s2.in0 = y2;
            s2.res = in.p;
            s2.stats = true;
            s2.name = "res.conv2 (train)";
            DDIF_TRY(add_conv(step, s2, out));
            TrainMod m;
            m.kind = TrainMod::RES;
            m.key = rb;
            m.in = in;
            m.out = *out;
            m.t[0] = h1;
            m.t[1] = y2;
            m.mask = drop_sites.back().mask;
            m.slot = net->slot_off.at(rb);
            tmods.push_back(m);
            return 0;
        }
        ConvSpec s1;
        s1.pc = c1;
        s1.in0 = in;
        s1.pro = PRO_GN_SILU;
        s1.gamma = V(rb + ".block1.block.0.weight");
        s1.beta = V(rb + ".block1.block.0.bias");
        s1.tb_off = net->slot_off.at(rb);
        s1.stats = true;
        s1.name = "res.conv1";
        DDIF_TRY(add_conv(step, s1, &h1));
        ConvSpec s2;
        s2.pc = c2;
        s2.in0 = h1;
        s2.pro = PRO_GN_SILU;
        s2.gamma = V(rb + ".block2.block.0.weight");
        s2.beta = V(rb + ".block2.block.0.bias");
        s2.res = in.p;
        s2.stats = true;
        s2.xf = xf;
        s2.name = "res.conv2";
        return add_conv(step, s2, out);
    };
    auto attention = [&](const std::string& ap, Tensor in, Tensor* out) -> int {
        const PackedConv *cq = PC(ap + ".qkv"), *co = PC(ap + ".out");
        if (!cq || !co) return fail(DDIF_ERR_MISSING, "%s: conv weights missing", ap.c_str());
        if (!train_mode && in.H * in.W == 64 && in.C == 128 && x3_enabled() && lr_enabled() && cq->w_x3 && co->w_x3 && cq->ck == 32 && co->ck == 32 && in.st) {
            // 64-token tiles: GroupNorm -> qkv -> softmax(q k^T / sqrt(C)) v -> out + bias + x in ONE kernel, one workgroup per
            // sample (kernels_attn.h); other sizes take the three-launch path below
            use(in.p);
            DDIF_TRY(alloc_tensor(out, in.C, in.H, in.W, true));
            const int asplit = attn_block_split();  // workgroups (= statistics partials) per sample
            out->np = asplit;
            DDIF_TRY(dalloc(&out->st, (size_t)B * 2 * asplit));
            DDIF_TRY(attn_block_prepare());
            AttnBlockArgs a{};
            a.x = in.p;
            a.st = in.st;
            a.np = in.np;
            a.gamma = V(ap + ".norm.weight");
            a.beta = V(ap + ".norm.bias");
            a.wqkv = cq->w_x3;
            {   // qkv on f16x2 (three products) where the f16x2 conditions of a GroupNorm-prologue conv hold: half packs exist (|w| < 64), and the bound
                // sqrt(N) max|gamma| + max|beta| of GroupNorm's output stays inside the scaled half range; DDIF_ATTN_F16=0 keeps bf16x3 (tests/test_env_switches.py)
                static const bool af16 = [] { const char* e = getenv("DDIF_ATTN_F16"); return !e || atoi(e) != 0; }();
                auto ig = net->vec_absmax.find(a.gamma), ib = net->vec_absmax.find(a.beta);
                const bool okr = ig != net->vec_absmax.end() && ib != net->vec_absmax.end() &&
                                 std::sqrt((double)in.C * in.H * in.W) * ig->second + ib->second < DDIF_F16_AMAX;
                a.wqkv_f16 = (af16 && f16_enabled() && cq->w_f16 && okr) ? cq->w_f16 : nullptr;
            }
            a.wout = co->w_x3;
            a.bout = co->bias ? co->bias : zeros;
            a.scale = 1.0f / std::sqrt((float)in.C);  // 1/sqrt(C), not 1/sqrt(d)   (sr3_dwt.py:352)
            a.out = out->p;
            a.st_out = out->st;
            a.B = B;
            a.xcd = (xcd_mask() & 8) ? 1 : 0;
            if (!a.gamma || !a.beta) return fail(DDIF_ERR_MISSING, "%s: norm weights missing", ap.c_str());
            Op op;
            op.name = "attn_block";
            op.label = "attn_block (GN + qkv + attention + out + residual) @8x8";
            op.cls = 3;
            op.flop = 2.0 * B * 64 * 128.0 * (384 + 128) + 4.0 * B * 8 * 64.0 * 64 * 16;
            op.bytes = 8.0 * B * 64 * 128;
            const int ncu = num_cus();
            op.run = [a, ncu, asplit](hipStream_t s, const StepCtx&) { attn_block_launch(a, a.B * asplit < ncu ? a.B * asplit : ncu, s); };
            step.push_back(std::move(op));
            return 0;
        }
        Tensor qkv, o;
        ConvSpec s1;
        s1.pc = cq;
        s1.in0 = in;
        s1.pro = PRO_GN;
        s1.gamma = V(ap + ".norm.weight");
        s1.beta = V(ap + ".norm.bias");
        s1.use_bias = false;
        s1.name = "attn.qkv";
        DDIF_TRY(add_conv(step, s1, &qkv));
        DDIF_TRY(alloc_tensor(&o, in.C, in.H, in.W, true));
        use(qkv.p);
        {
            Op op;
            op.name = "self_attn";
            op.cls = 3;
            const int n = in.H * in.W, Cc = in.C, BB = B;
            const float scale = 1.0f / std::sqrt((float)Cc);  // 1/sqrt(C), not 1/sqrt(d)   (sr3_dwt.py:352)
            op.flop = 4.0 * B * 8 * (double)n * n * 16;
            op.bytes = 4.0 * B * n * 4.0 * Cc;
            op.run = [qkv, o, n, Cc, BB, scale](hipStream_t s, const StepCtx&) {
                hipLaunchKernelGGL(self_attn_mfma_kernel, dim3((n + 63) / 64, 8, BB), dim3(64), 0, s, (const float*)qkv.p, n, Cc, scale, o.p);
            };
            step.push_back(std::move(op));
        }
        ConvSpec s2;
        s2.pc = co;
        s2.in0 = o;
        s2.res = in.p;
        s2.stats = true;
        s2.name = "attn.out";
        DDIF_TRY(add_conv(step, s2, out));
        if (train_mode) {
            TrainMod m;
            m.kind = TrainMod::ATTN;
            m.key = ap;
            m.in = in;
            m.out = *out;
            m.t[0] = qkv;
            m.t[1] = o;
            tmods.push_back(m);
        }
        return 0;
    };

    // ---- step program
    Tensor cur;
    std::vector<Tensor> feats;
    std::vector<int> feat_mod;  // train: index in tmods of the module that produced each feature
    int lev = 0;
    // cond-only FiLM branch of an encoder block: body(cond) -> scale | shift (sr3_dwt.py:379-391); built when first asked for -- by the block itself, or one
    // layer earlier by the conv that folds the block's x_conv + FiLM into its epilogue (round 6)
    std::map<const Layer*, std::pair<Tensor, Tensor>> film_cache;  // block -> (hid, film)
    auto film_of = [&](const Layer& Lb, int at_lev, Tensor* hid_out, Tensor* film_out) -> int {
        auto it = film_cache.find(&Lb);
        if (it != film_cache.end()) {
            *hid_out = it->second.first;
            *film_out = it->second.second;
            return 0;
        }
        const std::string ci = Lb.p + ".cond_inj";
        Tensor hid, film;
        ConvSpec s;
        s.pc = PC(ci + ".body.0");
        if (!s.pc) return fail(DDIF_ERR_MISSING, "%s.body.0 missing", ci.c_str());
        s.in0 = cenc[at_lev];
        s.use_bias = false;
        s.stats = true;
        s.name = "film.body0";
        DDIF_TRY(add_conv(pre, s, &hid));
        ConvSpec s2;
        s2.pc = PC(ci + ".body.3");
        if (!s2.pc) return fail(DDIF_ERR_MISSING, "%s.body.3 missing", ci.c_str());
        s2.in0 = hid;
        s2.pro = PRO_GN_SILU;
        s2.gamma = V(ci + ".body.1.weight");
        s2.beta = V(ci + ".body.1.bias");
        s2.name = "film.body3";
        DDIF_TRY(add_conv(pre, s2, &film));
        film_cache[&Lb] = std::make_pair(hid, film);
        *hid_out = hid;
        *film_out = film;
        return 0;
    };
    // the fold request a producer conv at index li (stem / Downsample / the last conv of a block) hands to add_conv: the NEXT layer must be an encoder block
    XfReq xf_req;           // of the layer being built (add_conv fills out / done)
    bool xf_have = false;   // the previous layer's producer has already written this block's y = FiLM(x_conv(.))
    Tensor xf_y;
    auto xf_prepare = [&](size_t li, int next_lev) -> int {
        xf_req = XfReq();
        if (train_mode || li + 1 >= net->downs.size()) return 0;
        const Layer& Ln = net->downs[li + 1];
        if (Ln.kind == L_STEM || Ln.kind == L_DOWN) return 0;
        const std::string ci = Ln.p + ".cond_inj";
        const PackedConv* px = PC(ci + ".x_conv");
        const float* wxf = V(ci + ".x_conv.xf");
        if (!px || !wxf) return 0;
        Tensor hid, film;
        DDIF_TRY(film_of(Ln, next_lev, &hid, &film));
        xf_req.pc = px;
        xf_req.w = wxf;
        xf_req.film = film.p;
        return 0;
    };
    for (size_t li = 0; li < net->downs.size(); ++li) {
        auto& L = net->downs[li];
        if (L.kind == L_STEM) {
            const PackedConv* pc = PC(L.p);
            if (!pc) return fail(DDIF_ERR_MISSING, "%s missing", L.p.c_str());
            ConvSpec s;
            s.pc = pc;
            s.dyn_input = true;
            if (c.self_condition) {
                s.in0 = sc_in;
                s.in1 = x_in;
            } else {
                s.in0 = x_in;
            }
            s.stats = true;
            s.name = "stem";
            DDIF_TRY(xf_prepare(li, lev));
            s.xf = &xf_req;
            DDIF_TRY(add_conv(step, s, &cur));
            xf_have = xf_req.done;
            xf_y = xf_req.out;
            if (train_mode) {
                TrainMod m;
                m.kind = TrainMod::STEM;
                m.key = L.p;
                m.out = cur;
                tmods.push_back(m);
            }
        } else if (L.kind == L_DOWN) {
            ConvSpec s;
            s.pc = PC(L.p + ".conv");
            if (!s.pc) return fail(DDIF_ERR_MISSING, "%s.conv missing", L.p.c_str());
            s.in0 = cur;
            s.stride = 2;
            s.stats = true;
            s.name = "down";
            const Tensor din = cur;
            DDIF_TRY(xf_prepare(li, lev + 1));
            s.xf = &xf_req;
            DDIF_TRY(add_conv(step, s, &cur));
            xf_have = xf_req.done;
            xf_y = xf_req.out;
            ++lev;
            if (train_mode) {
                TrainMod m;
                m.kind = TrainMod::DOWN;
                m.key = L.p + ".conv";
                m.in = din;
                m.out = cur;
                tmods.push_back(m);
            }
        } else {
            const std::string ci = L.p + ".cond_inj";
            // cond-only: body(cond) -> FiLM scale|shift   (sr3_dwt.py:379-391)
            Tensor hid, film;
            DDIF_TRY(film_of(L, lev, &hid, &film));
            Tensor y;
            const bool y_folded = xf_have;  // the producer one layer up wrote y already (EPI_XF)
            xf_have = false;
            ConvSpec s;
            s.pc = PC(ci + ".x_conv");
            if (!s.pc) return fail(DDIF_ERR_MISSING, "%s.x_conv missing", ci.c_str());
            s.in0 = cur;
            if (train_mode) {
                // train: xc = x_conv(x) is kept (FiLM's backward needs it), the modulation is a launch of its own
                Tensor xc;
                s.name = "film.x_conv (train)";
                DDIF_TRY(add_conv(step, s, &xc));
                DDIF_TRY(alloc_tensor(&y, xc.C, xc.H, xc.W, true));
                const int HWl = xc.H * xc.W, Cc = xc.C, BB = B;
                const int chunks = tk::film_chunks(HWl, Cc);
                y.np = chunks;
                DDIF_TRY(dalloc(&y.st, (size_t)B * chunks * 2));
                Tensor yy = y;
                Op op;
                op.name = "film_apply";
                op.cls = (HWl <= 256) ? 2 : 5;
                op.bytes = 16.0 * B * HWl * Cc;
                op.run = [xc, film, yy, HWl, Cc, BB, chunks](hipStream_t st, const StepCtx&) {
                    tk::film_apply(st, xc.p, film.p, BB, HWl, Cc, yy.p, yy.st, chunks);
                };
                step.push_back(std::move(op));
                // the cond image zero-padded to 4 | channels: input of body.0's weight gradient (cond-only, part of set_cond)
                if ((int)cenc_pad.size() < nlev) cenc_pad.resize(nlev);
                if (!cenc_pad[lev].p) {
                    const int Cp = (cenc[lev].C + 3) & ~3;
                    DDIF_TRY(alloc_tensor(&cenc_pad[lev], Cp, cenc[lev].H, cenc[lev].W));
                }
                TrainMod m;
                m.kind = TrainMod::FILM;
                m.key = ci;
                m.in = cur;
                m.out = y;
                m.t[0] = xc;
                m.t[1] = film;
                m.t[2] = hid;
                m.t[3] = cenc[lev];
                m.t[4] = cenc_pad[lev];
                m.lev = lev;
                tmods.push_back(m);
            } else if (y_folded) {
                y = xf_y;
            } else {
            s.film = film.p;
            s.stats = true;
            s.name = "film.x_conv";
            DDIF_TRY(add_conv(step, s, &y));
            }
            // the block's last conv folds the NEXT block's x_conv + FiLM when nothing (attention) sits between them
            xf_req = XfReq();
            if (!L.attn) DDIF_TRY(xf_prepare(li, lev));
            DDIF_TRY(resblock(L.p + ".res_block", y, &cur, (!train_mode && !L.attn) ? &xf_req : nullptr));
            xf_have = xf_req.done;
            xf_y = xf_req.out;
            if (L.attn) {
                Tensor t2;
                DDIF_TRY(attention(L.p + ".attn", cur, &t2));
                cur = t2;
            }
        }
        feats.push_back(cur);
        if (train_mode) {
            tmods.back().pushes_feat = true;
            feat_mod.push_back((int)tmods.size() - 1);
        }
    }
    for (auto& L : net->mid) {
        Tensor t1;
        DDIF_TRY(resblock(L.p + ".res_block", cur, &t1));
        cur = t1;
        if (L.attn) {
            Tensor t2;
            DDIF_TRY(attention(L.p + ".attn", cur, &t2));
            cur = t2;
        }
    }
    if (train_mode) {  // the decoder-only half of the cond-only program ran on the side stream (set_cond): it must be complete from here on
        Op j;
        j.name = "join cond-only side work";
        j.run = [this](hipStream_t s, const StepCtx&) {  // (not under a stream capture: the samplers join before they start capturing)
            if (this->side_pending) {
                this->train_join(s);
                this->side_pending = false;
            }
        };
        step.push_back(std::move(j));
    }
    for (auto& L : net->ups) {
        if (L.kind == L_UP) {
            ConvSpec s;
            s.pc = PC(L.p + ".conv");
            if (!s.pc) return fail(DDIF_ERR_MISSING, "%s.conv missing", L.p.c_str());
            s.in0 = cur;
            s.ups = 1;
            s.stats = true;
            s.name = "up";
            const Tensor uin = cur;
            DDIF_TRY(add_conv(step, s, &cur));
            --lev;
            if (train_mode) {
                TrainMod m;
                m.kind = TrainMod::UP;
                m.key = L.p + ".conv";
                m.in = uin;
                m.out = cur;
                tmods.push_back(m);
            }
            continue;
        }
        const std::string ci = L.p + ".cond_inj";
        Tensor skip = feats.back();
        feats.pop_back();
        int skip_from = -1;
        if (train_mode) {
            skip_from = feat_mod.back();
            feat_mod.pop_back();
        }
        if (skip.C != L.cskip || cur.C != L.cx || skip.H != cur.H || skip.W != cur.W)
            return fail(DDIF_ERR_INVALID, "%s: skip/feature shape mismatch", L.p.c_str());
        const int fea = L.cin, d = fea / 8, Hl = cur.H, Wl = cur.W, BB = B;
        const int cd = Cl + 3 * P;
        // ---- cond-only: kv -> softmax_W(k) -> context   (sr3_dwt.py:514-517,541,546,563)
        const size_t pre_dec0 = pre.size();  // train: everything this block adds to the cond-only program is decoder-only -> the side stream
        float* ctx = nullptr;
        Tensor kdw, kv;
        {
            DDIF_TRY(alloc_tensor(&kdw, cd, Hl, Wl));
            {
                DwArgs a{};
                a.in0 = cdec[lev].p;
                a.c0 = cd;
                a.B = B;
                a.H = Hl;
                a.W = Wl;
                a.w = V(ci + ".kv.0.weight");
                if (!a.w) return fail(DDIF_ERR_MISSING, "%s.kv.0.weight missing", ci.c_str());
                a.out_dw = kdw.p;
                a.tiles_x = (Wl + 15) / 16;
                a.tiles_y = (Hl + 7) / 8;
                Op op;
                op.name = "kv.dw3x3";
                op.flop = 2.0 * 9 * B * Hl * Wl * cd;
                op.bytes = 8.0 * B * Hl * Wl * cd;
                op.run = [a, BB](hipStream_t s, const StepCtx&) {
                    launch_dw3x3(s, a);
                };
                pre.push_back(std::move(op));
            }
            ConvSpec s;
            s.pc = PC(ci + ".kv.1");
            if (!s.pc) return fail(DDIF_ERR_MISSING, "%s.kv.1 missing", ci.c_str());
            s.in0 = kdw;
            s.name = "kv.1x1";
            DDIF_TRY(add_conv(pre, s, &kv));
            float *kmx = nullptr, *ksm = nullptr;
            if (!train_mode) {
            DDIF_TRY(dalloc(&kmx, (size_t)B * Hl * fea));
            DDIF_TRY(dalloc(&ksm, (size_t)B * Hl * fea));
            DDIF_TRY(dalloc(&ctx, (size_t)B * 8 * d * d));
            {
                Op op;
                op.name = "k.softmax_stats";
                op.bytes = 8.0 * B * Hl * Wl * fea;
                op.run = [kv, kmx, ksm, fea, BB, Hl, Wl](hipStream_t s, const StepCtx&) {
                    hipLaunchKernelGGL(softmax_stats_kernel, ew_grid((size_t)BB * Hl * fea), dim3(256), 0, s, (const float*)kv.p, 2 * fea, 0, fea, BB, Hl, Wl, 1, kmx, ksm);
                };
                pre.push_back(std::move(op));
            }
            {
                Op op;
                op.name = "linattn_ctx";
                op.flop = 2.0 * B * 8 * d * d * (double)Hl * Wl;
                op.bytes = 4.0 * B * Hl * Wl * 2.0 * fea;
                op.run = [kv, kmx, ksm, ctx, fea, d, BB, Hl, Wl](hipStream_t s, const StepCtx&) {
                    hipLaunchKernelGGL(linattn_ctx_kernel, dim3(8, BB), dim3(256), 2 * 32 * d * sizeof(float), s, (const float*)kv.p, (const float*)kmx, (const float*)ksm, BB, Hl, Wl, fea, d, ctx);
                };
                pre.push_back(std::move(op));
            }
            }  // !train_mode (the train-mode forward runs the attention core itself)
        }
        if (train_mode) {
            // ---- train mode: the block op by op (models/sr3_dwt.py:536-577), every intermediate the reverse pass needs kept in memory:
            //   xn = GN(cat[h, skip]), dwq = dw3x3(xn)        one launch (dw3x3_kernel, two sources, GroupNorm from the producers' partials)
            //   q = q.1(dwq);  o = linear attention(q, kv)     (kv comes from set_cond);  a = attn_out(o) + attn_res(xn)  as one 1x1 conv over cat[o, xn]
            //   f0 = ffn.0(a); f1 = silu(f0); f2 = ffn.2(f1); f3c = ffn.3(f2) + b;  out = a + DropPath(f3c)
            const PackedConv* pq1t = PC(ci + ".q.1");
            const PackedConv* pmixt = PC(ci + ".attn_mix");
            if (!pq1t || !pmixt) return fail(DDIF_ERR_MISSING, "%s: q.1 / attn_out missing", ci.c_str());
            if (!cur.st || !skip.st) return fail(DDIF_ERR_STATE, "%s: prenorm without producer statistics", ci.c_str());
            const float *pn_g = V(ci + ".prenorm_x.weight"), *pn_b = V(ci + ".prenorm_x.bias"), *q0w = V(ci + ".q.0.weight");
            if (!pn_g || !pn_b || !q0w) return fail(DDIF_ERR_MISSING, "%s: prenorm/q.0 weights missing", ci.c_str());
            Tensor xn, dwq, q, o, a, f0, f1, f2, f3c, f3, kdw_pad;
            float *la_ctx = nullptr, *la_part = nullptr;
            DDIF_TRY(alloc_tensor(&xn, fea, Hl, Wl, true));
            DDIF_TRY(alloc_tensor(&dwq, fea, Hl, Wl, true));
            {
                DwArgs da{};
                da.in0 = cur.p;
                da.c0 = cur.C;
                da.in1 = skip.p;
                da.c1 = skip.C;
                da.B = B;
                da.H = Hl;
                da.W = Wl;
                da.st0 = cur.st;
                da.np0 = cur.np;
                da.st1 = skip.st;
                da.np1 = skip.np;
                da.gamma = pn_g;
                da.beta = pn_b;
                da.w = q0w;
                da.out_dw = dwq.p;
                da.out_xn = xn.p;
                da.use_gn = 1;
                da.tiles_x = (Wl + 15) / 16;
                da.tiles_y = (Hl + 7) / 8;
                Op op;
                op.name = "q.gn_dw3x3 (train)";
                op.cls = (Hl * Wl <= 256) ? 2 : 5;
                op.flop = 2.0 * 9 * B * Hl * Wl * fea;
                op.bytes = 12.0 * B * Hl * Wl * fea;
                op.run = [da, BB](hipStream_t s, const StepCtx&) {
                    launch_dw3x3(s, da);
                };
                step.push_back(std::move(op));
            }
            {
                ConvSpec s;
                s.pc = pq1t;
                s.in0 = dwq;
                s.name = "q.1x1 (train)";
                DDIF_TRY(add_conv(step, s, &q));
            }
            DDIF_TRY(alloc_tensor(&o, fea, Hl, Wl, true));
            {
                Op op;
                op.name = "linattn_fwd";
                op.cls = (Hl * Wl <= 256) ? 2 : 5;
                op.flop = 4.0 * B * 8 * d * d * (double)Hl * Wl;
                op.bytes = 16.0 * B * Hl * Wl * fea;
                Tensor qq = q, kk = kv, oo = o;
                if ((Hl > Wl ? Hl : Wl) * fea > 8192 || d > 32 || d % 4)
                    return fail(DDIF_ERR_INVALID, "%s: train-mode linear attention holds one image line x %d channels in LDS: lines of more than %d pixels are not supported",
                                ci.c_str(), fea, 8192 / fea);
                DDIF_TRY(dalloc(&la_ctx, (size_t)B * fea * d));
                DDIF_TRY(dalloc(&la_part, tk::linattn_part_floats(B, Hl, Wl, fea, d)));
                float *cx = la_ctx, *pt = la_part;
                {   // cond-only half: softmax_W(k), context = k v^T per head (sr3_dwt.py:541,546,563) -- part of set_cond, like the eval plan's
                    Op pc;
                    pc.name = "linattn_ctx (train)";
                    pc.bytes = 8.0 * B * Hl * Wl * fea;
                    pc.run = [kk, BB, d, Hl, Wl, cx, pt](hipStream_t s, const StepCtx&) { tk::linattn_ctx(s, kk.p, BB, 8, d, Hl, Wl, cx, pt); };
                    pre.push_back(std::move(pc));
                    for (size_t i = pre_dec0; i < pre.size(); ++i) pre[i].side = true;
                }
                op.run = [qq, oo, BB, d, Hl, Wl, fea, cx](hipStream_t s, const StepCtx&) { tk::linattn_apply(s, qq.p, cx, BB, 8, d, Hl, Wl, oo.p, fea); };
                step.push_back(std::move(op));
            }
            const bool has_res = pmixt->cin == 2 * fea;
            {
                ConvSpec s;
                s.pc = pmixt;
                s.in0 = o;
                if (has_res) s.in1 = xn;
                else s.res = xn.p;  // attn_res is Identity
                s.name = "attn_out+res (train)";
                DDIF_TRY(add_conv(step, s, &a));
            }
            {
                ConvSpec s;
                s.pc = PC(ci + ".ffn.0");
                if (!s.pc) return fail(DDIF_ERR_MISSING, "%s.ffn.0 missing", ci.c_str());
                s.in0 = a;
                s.use_bias = false;
                s.name = "ffn.0 (train)";
                DDIF_TRY(add_conv(step, s, &f0));
                DDIF_TRY(alloc_tensor(&f1, f0.C, f0.H, f0.W, true));
                Op op;
                op.name = "silu";
                op.cls = (Hl * Wl <= 256) ? 2 : 5;
                op.bytes = 8.0 * B * Hl * Wl * f0.C;
                Tensor ff0 = f0, ff1 = f1;
                const size_t n = (size_t)B * Hl * Wl * f0.C;
                op.run = [ff0, ff1, n](hipStream_t st, const StepCtx&) { tk::silu_fwd(st, ff0.p, n, ff1.p); };
                step.push_back(std::move(op));
                ConvSpec s2;
                s2.pc = PC(ci + ".ffn.2");
                if (!s2.pc) return fail(DDIF_ERR_MISSING, "%s.ffn.2 missing", ci.c_str());
                s2.in0 = f1;
                s2.use_bias = false;
                s2.name = "ffn.2 (train)";
                DDIF_TRY(add_conv(step, s2, &f2));
                ConvSpec s3;
                s3.pc = PC(ci + ".ffn.3");
                if (!s3.pc) return fail(DDIF_ERR_MISSING, "%s.ffn.3 missing", ci.c_str());
                s3.in0 = f2;
                s3.name = "ffn.3 (train)";
                DDIF_TRY(add_conv(step, s3, &f3c));
            }
            float* scale = nullptr;
            {
                DDIF_TRY(alloc_tensor(&f3, f3c.C, f3c.H, f3c.W, true));
                DDIF_TRY(dalloc(&scale, (size_t)B));
                path_sites.push_back(scale);
                const int HW = f3c.H * f3c.W, Cc = f3c.C;
                int chunks = (HW * Cc / 4 + 256 * 8 - 1) / (256 * 8);
                if (chunks < 1) chunks = 1;
                f3.np = chunks;
                DDIF_TRY(dalloc(&f3.st, (size_t)B * chunks * 2));
                Tensor fo = f3, fc = f3c, aa = a;
                Op op;
                op.name = "droppath_add";
                op.cls = (HW <= 256) ? 2 : 5;
                op.bytes = 12.0 * B * HW * Cc;
                op.run = [fc, scale, aa, fo, HW, Cc, BB, chunks](hipStream_t s, const StepCtx&) {
                    hipLaunchKernelGGL(droppath_add_kernel, dim3(chunks, BB), dim3(256), 64, s, (const float*)fc.p, (const float*)scale, (const float*)aa.p, HW, Cc, fo.p, fo.st);
                };
                step.push_back(std::move(op));
            }
            // kv.0's output zero-padded to 4 | channels: input of kv.1's weight gradient (cond-only)
            {
                const int Cp = (cd + 3) & ~3;
                DDIF_TRY(alloc_tensor(&kdw_pad, Cp, Hl, Wl));
            }
            TrainMod m;
            m.kind = TrainMod::DEC;
            m.key = ci;
            m.in = cur;
            m.out = f3;
            m.t[0] = skip;
            m.t[1] = xn;
            m.t[2] = dwq;
            m.t[3] = q;
            m.t[4] = kv;
            m.t[5] = kdw;
            m.t[6] = kdw_pad;
            m.t[7] = o;
            m.t[8] = a;
            m.t[9] = f0;
            m.t[10] = f1;
            m.t[11] = f2;
            m.scale = scale;
            m.ctx = la_ctx;
            m.la_part = la_part;
            m.has_res = has_res;
            m.skip_from = skip_from;
            m.lev = lev;
            tmods.push_back(m);
            // f3c is only needed by the reverse pass through DropPath's scale: d(f3c) = scale * d(out); not stored in the record
            DDIF_TRY(resblock(L.p + ".res_block", f3, &cur));
            if (L.attn) {
                Tensor t2;
                DDIF_TRY(attention(L.p + ".attn", cur, &t2));
                cur = t2;
            }
            continue;
        }
        // ---- per step
        Tensor xn, q, amix, f1, f2, f3;
        const PackedConv* pq1 = PC(ci + ".q.1");
        if (!pq1) return fail(DDIF_ERR_MISSING, "%s.q.1 missing", ci.c_str());
        if (!cur.st || !skip.st) return fail(DDIF_ERR_STATE, "%s: prenorm without producer statistics", ci.c_str());
        const float *pn_g = V(ci + ".prenorm_x.weight"), *pn_b = V(ci + ".prenorm_x.bias"), *q0w = V(ci + ".q.0.weight");
        if (!pn_g || !pn_b || !q0w) return fail(DDIF_ERR_MISSING, "%s: prenorm/q.0 weights missing", ci.c_str());
        // ---- the whole attention half in ONE launch where the level keeps whole image columns inside a workgroup (kernels_lafuse.h):
        //      xn = GN(cat[h, skip]) -> q = q.1(q.0(xn)) -> softmax over H -> M_b p + W_res xn + bias; q and xn never reach memory
        bool fused_attn = false;
        {
            const PackedConv* pm = PC(ci + ".attn_mix");
            const float* wr = V(ci + ".attn_res.weight");
            // (round 6) the 8 x 8 level has a kernel of its own: half a sample per workgroup, the waves split the output channels (kernels_lafuse8.h)
            const bool la8 = la8_enabled() && lafuse8_supported(Hl, Wl, cur.C, skip.C, pm ? pm->cout : 0);
            bool ok = lafuse_enabled() && f16_enabled() && x3_enabled() && lr_enabled() && pm && wr && pq1->w_f16 && pq1->bias && pm->bias && (Hl * Wl >= 256 || la8) &&
                      pq1->cout == fea && pq1->ck == 32 && pm->ck == 32 && pm->cin == 2 * fea && cur.C % 16 == 0 && skip.C % 16 == 0 && (la8 || lafuse_supported(Hl, fea, pm->cout));
            if (ok) {  // f16x2 range of depthwise(GroupNorm(.)): (sqrt(N) max|gamma| + max|beta|) * 9 max|w_dw| inside the scaled half range
                auto ig = net->vec_absmax.find(pn_g), ib = net->vec_absmax.find(pn_b), iw = net->vec_absmax.find(q0w);
                ok = ig != net->vec_absmax.end() && ib != net->vec_absmax.end() && iw != net->vec_absmax.end() &&
                     (std::sqrt((double)fea * Hl * Wl) * ig->second + ib->second) * 9.0 * iw->second < DDIF_F16_AMAX;
            }
            if (ok) {
                const float* wo = V(ci + ".attn_out.weight");
                if (!wo) return fail(DDIF_ERR_MISSING, "%s.attn_out.weight missing", ci.c_str());
                const int nb_pad = (((pm->cout + 31) / 32) + 3) & ~3;
                const size_t per = (size_t)nb_pad * pm->n_chunks * 2 * 3 * 256;  // bf16x3 planes, 32-channel chunks
                float* wmix = nullptr;
                DDIF_TRY(dalloc(&wmix, per * B));
                {
                    Op op;
                    op.name = "pack_mix_weights";
                    const float scale = 1.0f / std::sqrt((float)d);
                    const int co_n = pm->cout, nch = pm->n_chunks;
                    op.flop = 2.0 * B * co_n * (double)fea * d;
                    op.run = [wo, wr, ctx, wmix, BB, co_n, fea, d, scale, nch, nb_pad, per](hipStream_t s, const StepCtx&) {
                        hipLaunchKernelGGL(pack_mix_weights_x3_kernel, ew_grid(per * BB), dim3(256), 0, s, wo, wr, (const float*)ctx, BB, co_n, fea, d, scale, 32, nch, nb_pad, wmix);
                    };
                    pre.push_back(std::move(op));
                }
                use(cur.p);
                use(skip.p);
                DDIF_TRY(alloc_tensor(&amix, pm->cout, Hl, Wl, true));
                LaFuseArgs a{};
                a.xcd = (xcd_mask() & 4) ? 1 : 0;
                a.in0 = cur.p;
                a.c0 = cur.C;
                a.in1 = skip.p;
                a.c1 = skip.C;
                a.B = B;
                a.H = Hl;
                a.W = Wl;
                a.st0 = cur.st;
                a.np0 = cur.np;
                a.st1 = skip.st;
                a.np1 = skip.np;
                a.gamma = pn_g;
                a.beta = pn_b;
                a.dw_w = q0w;
                a.wq = pq1->w_f16;
                a.nchq = pq1->n_chunks;
                a.bq = pq1->bias;
                a.wmix = wmix;
                a.wmix_bstride = (long long)per;
                a.nch_mix = pm->n_chunks;
                a.bias = pm->bias;
                a.out = amix.p;
                a.dout = pm->cout;
                // four-wave workgroups of 128 pixels instead of eight-wave ones of 256 when the latter would not fill the CUs (round 6; same results either way:
                // kernels_lafuse.h).  DDIF_LA_NW = 8 / 4 forces one form (tests/test_env_switches.py).
                int la_nw = 8;
                if (!la8) {
                    static const int nw_env = [] { const char* e = getenv("DDIF_LA_NW"); return e ? atoi(e) : 0; }();
                    const long wg8 = (long)B * ((Wl + lafuse_strip(Hl, 8) - 1) / lafuse_strip(Hl, 8));
                    la_nw = nw_env == 4 || nw_env == 8 ? nw_env : (wg8 < num_cus() ? 4 : 8);
                }
                if (!dry) DDIF_TRY(la8 ? lafuse8_launch(a, 1, nullptr, true) : lafuse_launch(a, 1, nullptr, true, la_nw));
                const int nstrips = la8 ? 2 : (Wl + lafuse_strip(Hl, la_nw) - 1) / lafuse_strip(Hl, la_nw);
                long cap = num_cus();
                if (g_debug_grid_cap > 0 && g_debug_grid_cap < cap) cap = g_debug_grid_cap;
                Op op;
                op.name = "linattn_fused";
                {
                    char lb[160];
                    snprintf(lb, sizeof lb, "linattn_fused GN+dw3x3+q.1+softmax_H+attn_out+res %d+%d->%d @%dx%d", cur.C, skip.C, pm->cout, Hl, Wl);
                    op.label = lb;
                }
                op.flop = 2.0 * B * Hl * Wl * ((double)fea * fea + 9.0 * fea + 2.0 * fea * pm->cout);
                op.bytes = 4.0 * B * Hl * Wl * ((double)fea + pm->cout);
                op.cls = (Hl * Wl <= 256) ? 2 : 1;
                // q.1 on f16x2 (x3), attn_out / attn_res on bf16x3 (x6): weight of the sum
                op.mfma_w = (3.0 * fea * fea + 6.0 * 2.0 * fea * pm->cout) / ((double)fea * fea + 9.0 * fea + 2.0 * fea * pm->cout);
                if (la8) op.name = "linattn8_fused";
                op.run = [a, nstrips, cap, la8, la_nw](hipStream_t st, const StepCtx&) {
                    const long nw = (long)a.B * nstrips;
                    if (la8) (void)lafuse8_launch(a, (int)(nw < cap ? nw : cap), st, false);
                    else (void)lafuse_launch(a, (int)(nw < cap ? nw : cap), st, false, la_nw);
                };
                step.push_back(std::move(op));
                fused_attn = true;
            }
        }
        if (!fused_attn) {
        DDIF_TRY(alloc_tensor(&xn, fea, Hl, Wl, true));
        if (pq1->ck != 32 || cur.C % 4 != 0 || skip.C % 4 != 0 || fea > 256)
            return fail(DDIF_ERR_INVALID, "%s: the fused q = 1x1(dw3x3(GN(cat))) kernel needs 4 | channels and <= 256 of them (got %d+%d)", ci.c_str(), cur.C, skip.C);
        float *qmx = nullptr, *qsm = nullptr;
        if (pick_cfg(1, 32, PRO_NONE, 1, 1, 0, Hl, Wl, pq1->cout, B, fea, fea) >= 20) {
            // low-resolution levels: xn = GN(cat), dwq = depthwise3x3(xn) from ONE small kernel (whole sample per workgroup),
            // then q = q.1(dwq) on the split-K kernel -- fused into the 1x1 conv the depthwise pass would be recomputed by
            // every 32-cout tile (sr3_dwt.py:507-513,537,540)
            Tensor dwq;
            DDIF_TRY(alloc_tensor(&dwq, fea, Hl, Wl, true));
            use(cur.p);
            use(skip.p);
            use(xn.p);
            DwArgs a{};
            a.in0 = cur.p;
            a.c0 = cur.C;
            a.in1 = skip.p;
            a.c1 = skip.C;
            a.B = B;
            a.H = Hl;
            a.W = Wl;
            a.st0 = cur.st;
            a.np0 = cur.np;
            a.st1 = skip.st;
            a.np1 = skip.np;
            a.gamma = pn_g;
            a.beta = pn_b;
            a.w = q0w;
            a.out_dw = dwq.p;
            a.out_xn = xn.p;
            a.use_gn = 1;
            Op op;
            op.name = "q.gn_dw3x3";
            op.cls = 2;
            {
                char lb[160];
                snprintf(lb, sizeof lb, "q.gn_dw3x3 %d+%d @%dx%d", cur.C, skip.C, Hl, Wl);
                op.label = lb;
            }
            op.flop = 2.0 * 9 * B * Hl * Wl * fea;
            op.bytes = 4.0 * B * Hl * Wl * 3.0 * fea;
            const size_t sm = (size_t)(Hl + 2) * (Wl + 2) * 36 * sizeof(float);
            if (sm > 64 * 1024) DDIF_HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(gn_dw3x3_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
            a.xcd = (xcd_mask() & 8) ? 1 : 0;
            const dim3 gdw = a.xcd ? dim3(BB, (fea + 31) / 32) : dim3((fea + 31) / 32, BB);
            op.run = [a, gdw, sm](hipStream_t s, const StepCtx&) { hipLaunchKernelGGL(gn_dw3x3_small_kernel, gdw, dim3(256), sm, s, a); };
            step.push_back(std::move(op));
            ConvSpec s;
            s.pc = pq1;
            s.in0 = dwq;
            s.name = "q.1x1";
            if (Hl <= 16) {  // whole columns inside one tile: softmax_H statistics of q from the conv's epilogue
                DDIF_TRY(dalloc(&qmx, (size_t)B * Wl * fea));
                DDIF_TRY(dalloc(&qsm, (size_t)B * Wl * fea));
                s.cso_mx = qmx;
                s.cso_sm = qsm;
                s.name = "q.1x1 (+softmax_H stats)";
            }
            DDIF_TRY(add_conv(step, s, &q));
        } else {
            // q = q.1(depthwise3x3(GroupNorm(cat[h, skip]))) in ONE kernel; also emits xn (sr3_dwt.py:507-513,537,540)
            ConvSpec s;
            s.pc = pq1;
            s.in0 = cur;
            s.in1 = skip;
            s.pro = PRO_GN_DW;
            s.gamma = pn_g;
            s.beta = pn_b;
            s.dw_w = q0w;
            s.out_xn = xn.p;
            s.name = "q = 1x1(dw3x3(GN(cat)))";
            DDIF_TRY(add_conv(step, s, &q));
        }
        if (!qmx) {
            DDIF_TRY(dalloc(&qmx, (size_t)B * Wl * fea));
            DDIF_TRY(dalloc(&qsm, (size_t)B * Wl * fea));
            use(q.p);
            Op op;
            op.name = "q.softmax_stats";
            op.cls = 4;
            op.bytes = 8.0 * B * Hl * Wl * fea;
            op.run = [q, qmx, qsm, fea, BB, Hl, Wl](hipStream_t s, const StepCtx&) {
                hipLaunchKernelGGL(softmax_stats_kernel, ew_grid((size_t)BB * Wl * fea), dim3(256), 0, s, (const float*)q.p, fea, 0, fea, BB, Hl, Wl, 0, qmx, qsm);
            };
            step.push_back(std::move(op));
        }
        const PackedConv* pmix = PC(ci + ".attn_mix");
        if (!pmix) return fail(DDIF_ERR_MISSING, "%s.attn_out missing", ci.c_str());
        // the linear-attention context is folded into per-sample attn_out weights (needs 32-channel chunks)
        if (fea % 32 != 0 || pmix->ck != 32) return fail(DDIF_ERR_INVALID, "%s: linear attention over %d channels needs 32 | channels", ci.c_str(), fea);
        {
            // cond-only: M_b = scale * W_out . blockdiag(ctx_b^T), packed per sample next to W_res
            const int nb_pad = (((pmix->cout + 31) / 32) + 3) & ~3;
            // the layout follows the instantiation add_conv() will pick for this conv (bf16x3 planes or fp32 fragments)
            const bool mix_x3 = pick_cfg(1, pmix->ck, PRO_COLSM, 1, 1, 0, Hl, Wl, pmix->cout, B, pmix->cin, pmix->cin == 2 * fea ? fea : pmix->cin) >= 12;
            const size_t per = mix_x3 ? (size_t)nb_pad * pmix->n_chunks * (pmix->ck / 16) * 3 * 256 : (size_t)nb_pad * pmix->n_chunks * (pmix->ck / 8) * 256;
            float* wmix = nullptr;
            DDIF_TRY(dalloc(&wmix, per * B));
            const float* wo = V(ci + ".attn_out.weight");
            const float* wr = V(ci + ".attn_res.weight");  // null: attn_res is Identity
            if (!wo) return fail(DDIF_ERR_MISSING, "%s.attn_out.weight missing", ci.c_str());
            {
                Op op;
                op.name = "pack_mix_weights";
                const float scale = 1.0f / std::sqrt((float)d);
                const int co_n = pmix->cout, ck = pmix->ck, nch = pmix->n_chunks;
                op.flop = 2.0 * B * co_n * (double)fea * d;
                op.run = [wo, wr, ctx, wmix, BB, co_n, fea, d, scale, ck, nch, nb_pad, per, mix_x3](hipStream_t s, const StepCtx&) {
                    if (mix_x3)
                        hipLaunchKernelGGL(pack_mix_weights_x3_kernel, ew_grid(per * BB), dim3(256), 0, s, wo, wr, (const float*)ctx, BB, co_n, fea, d, scale, ck, nch, nb_pad, wmix);
                    else
                        hipLaunchKernelGGL(pack_mix_weights_kernel, ew_grid(per * BB), dim3(256), 0, s, wo, wr, (const float*)ctx, BB, co_n, fea, d, scale, ck, nch, nb_pad, wmix);
                };
                pre.push_back(std::move(op));
            }
            ConvSpec s;
            s.pc = pmix;
            s.in0 = q;
            if (pmix->cin == 2 * fea) s.in1 = xn;
            else s.res = xn.p;  // attn_res is Identity
            s.pro = PRO_COLSM;
            s.cs_mx = qmx;
            s.cs_sm = qsm;
            s.w_override = wmix;
            s.w_bstride = (long long)per;
            s.name = "softmax_H(q).ctx.attn_out+res";
            DDIF_TRY(add_conv(step, s, &amix));
        }
        }  // !fused_attn
        {
            ConvSpec s;
            s.pc = PC(ci + ".ffn.0");
            if (!s.pc) return fail(DDIF_ERR_MISSING, "%s.ffn.0 missing", ci.c_str());
            s.in0 = amix;
            s.use_bias = false;
            s.silu = true;
            s.name = "ffn.0";
            DDIF_TRY(add_conv(step, s, &f1));
            ConvSpec s2;
            s2.pc = PC(ci + ".ffn.2");
            if (!s2.pc) return fail(DDIF_ERR_MISSING, "%s.ffn.2 missing", ci.c_str());
            s2.in0 = f1;
            s2.use_bias = false;
            s2.name = "ffn.2";
            ConvSpec s3;
            s3.pc = PC(ci + ".ffn.3");
            if (!s3.pc) return fail(DDIF_ERR_MISSING, "%s.ffn.3 missing", ci.c_str());
            // eval: ffn[3] o ffn[2] is one 3x3 conv (no nonlinearity between them; merged weights ".ffn.23", ddif_net.cpp)
            bool fused_ffn3 = false;
            if (const PackedConv* pm = train_mode ? nullptr : PC(ci + ".ffn.23")) {
                ConvSpec sf = s2;
                sf.pc = pm;
                sf.use_bias = true;
                sf.res = amix.p;
                sf.stats = true;
                sf.name = "ffn.3(ffn.2) merged + res";
                DDIF_TRY(add_conv(step, sf, &f3));
                fused_ffn3 = true;
            }
            if (!fused_ffn3) DDIF_TRY(add_conv(step, s2, &f2));
            s3.in0 = f2;
            if (fused_ffn3) {
            } else if (train_mode) {
                // ffn_drop_path(ffn(a)) + a  (sr3_dwt.py:576): the per-sample DropPath scale sits between the conv and the residual
                Tensor f3c;
                s3.name = "ffn.3 (train)";
                DDIF_TRY(add_conv(step, s3, &f3c));
                DDIF_TRY(alloc_tensor(&f3, f3c.C, f3c.H, f3c.W, true));
                use(f3c.p);
                use(amix.p);
                float* scale = nullptr;
                DDIF_TRY(dalloc(&scale, (size_t)B));
                path_sites.push_back(scale);
                const int HW = f3c.H * f3c.W, Cc = f3c.C;
                int chunks = (HW * Cc / 4 + 256 * 8 - 1) / (256 * 8);
                if (chunks < 1) chunks = 1;
                f3.np = chunks;
                DDIF_TRY(dalloc(&f3.st, (size_t)B * chunks * 2));
                Tensor fo = f3;
                Op op;
                op.name = "droppath_add";
                op.cls = (HW <= 256) ? 2 : 5;
                op.bytes = 12.0 * B * HW * Cc;
                op.run = [f3c, scale, amix, fo, HW, Cc, BB, chunks](hipStream_t s, const StepCtx&) {
                    hipLaunchKernelGGL(droppath_add_kernel, dim3(chunks, BB), dim3(256), 64, s, (const float*)f3c.p, (const float*)scale, (const float*)amix.p, HW, Cc, fo.p, fo.st);
                };
                step.push_back(std::move(op));
            } else {
                s3.res = amix.p;
                s3.stats = true;
                s3.name = "ffn.3";
                DDIF_TRY(add_conv(step, s3, &f3));
            }
        }
        DDIF_TRY(resblock(L.p + ".res_block", f3, &cur));
        if (L.attn) {
            Tensor t2;
            DDIF_TRY(attention(L.p + ".attn", cur, &t2));
            cur = t2;
        }
    }
    {
        ConvSpec s;
        s.pc = PC("final_conv.block.3");
        if (!s.pc) return fail(DDIF_ERR_MISSING, "final_conv.block.3 missing");
        s.in0 = cur;
        s.pro = PRO_GN_SILU;
        s.gamma = V("final_conv.block.0.weight");
        s.beta = V("final_conv.block.0.bias");
        s.name = "final";
        s.samp = true;
        DDIF_TRY(add_conv(step, s, &net_out));
        if (train_mode) {
            TrainMod m;
            m.kind = TrainMod::FINAL;
            m.key = "final_conv";
            m.in = cur;
            m.out = net_out;
            tmods.push_back(m);
        }
    }
    DDIF_TRY(ensure_tb(B));
    return 0;
#undef DDIF_TRY
}

void Plan::drop_graphs() {
#ifndef DDIF_EMU
    for (auto& g : graph_exec) {
        if (g) (void)hipGraphExecDestroy((hipGraphExec_t)g);
        g = nullptr;
    }
#endif
}

int Plan::ensure_tb(int rows) {
    if (rows <= tb_rows) return 0;
    drop_graphs();  // captured launches point at the old table
    if (int e = dalloc(&tb, (size_t)rows * net->nslots)) return e;  // older table stays in `allocs` until destroy
    if (int e = dalloc(&tvals, (size_t)rows)) return e;
    tb_rows = rows;
    return 0;
}

int Plan::time_rows(const float* t_host, int rows, hipStream_t s) {
    if (int e = ensure_tb(rows)) return e;
    DDIF_HIPCHK(hipMemcpyAsync(tvals, t_host, (size_t)rows * sizeof(float), hipMemcpyDefault, s));  // host or device source
    const int inner = net->cfg.inner_channel;
    hipLaunchKernelGGL(time_embed_kernel, dim3(rows), dim3(128), (size_t)6 * inner * sizeof(float), s, (const float*)tvals,
                       net->freqs, net->w1, net->b1, net->w3, net->b3, net->wall, net->ball, inner, net->nslots, tb, (float*)nullptr);
    return 0;
}

static int check_sampler_net(const Net* n);

int Plan::time_rows_aux(const float* t_host, int rows, float* aux, hipStream_t s) {
    if (int e = ensure_tb(rows)) return e;
    DDIF_HIPCHK(hipMemcpyAsync(tvals, t_host, (size_t)rows * sizeof(float), hipMemcpyDefault, s));  // host or device source
    const int inner = net->cfg.inner_channel;
    hipLaunchKernelGGL(time_embed_kernel, dim3(rows), dim3(128), (size_t)6 * inner * sizeof(float), s, (const float*)tvals,
                       net->freqs, net->w1, net->b1, net->w3, net->b3, net->wall, net->ball, inner, net->nslots, tb, aux);
    return 0;
}

namespace tk {
void dw3x3_plain(hipStream_t s, const float* in, int C, int B, int H, int W, const float* w9c, float* out, bool flip) {
    DwArgs a{};
    a.flip = flip ? 1 : 0;
    a.in0 = in;
    a.c0 = C;
    a.B = B;
    a.H = H;
    a.W = W;
    a.w = w9c;
    a.out_dw = out;
    a.tiles_x = (W + 15) / 16;
    a.tiles_y = (H + 7) / 8;
    launch_dw3x3(s, a);
}
}  // namespace tk

// One training iteration's device work (reference diffusion_ddpm_pan.py:692-766 main pass + diffusion_engine.py:233 loss.backward()):
// x_t = a x0 + s noise, train-mode forward (masks as set), L1 loss against x0, reverse program -> gradients in the bound tensors.
int Plan::train_step(const float* x0, const float* noise, const float* a_h, const float* s_h, const float* t_h, const float* sc, float* loss_dev, float* pred,
                     hipStream_t s) {
    if (!train_mode || bwd.empty()) return fail(DDIF_ERR_STATE, "ddif_plan_train_step: not a train-mode plan");
    if (!cond_set) return fail(DDIF_ERR_STATE, "ddif_plan_train_step before ddif_plan_set_cond");
    if (!x0 || !noise || !a_h || !s_h || !t_h) return fail(DDIF_ERR_INVALID, "ddif_plan_train_step: NULL argument");
    for (auto& kv : grad_slots)
        if (!kv.second) return fail(DDIF_ERR_STATE, "ddif_plan_train_step: gradient tensors are not bound (ddif_plan_train_bind); first missing: %s", kv.first.c_str());
    if (int e = check_sampler_net(net)) return e;
    const int HW = H * W;
    const size_t n = (size_t)B * HW * C;
    DDIF_HIPCHK(hipMemcpyAsync(small, a_h, (size_t)B * sizeof(float), hipMemcpyDefault, s));  // host or device source
    DDIF_HIPCHK(hipMemcpyAsync(small + B, s_h, (size_t)B * sizeof(float), hipMemcpyDefault, s));
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, ew_grid(n), dim3(256), 0, s, x0, B, C, HW, 0, C, img[0]);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, ew_grid(n), dim3(256), 0, s, noise, B, C, HW, 0, C, img[1]);
    hipLaunchKernelGGL(q_sample_kernel, ew_grid(n), dim3(256), 0, s, (const float*)img[0], (const float*)img[1], (const float*)small, (const float*)(small + B), B, (size_t)HW * C, x_in.p);
    if (sc) hipLaunchKernelGGL(nchw_to_nhwc_kernel, ew_grid(n), dim3(256), 0, s, sc, B, C, HW, 0, C, sc_in.p);
    return train_core(t_h, sc != nullptr, img[0], loss_dev, pred, s);
}

// forward + loss + reverse pass on a GIVEN network input (parity tests feed the reference's own x / target): x, target (B,C,H,W)
int Plan::train_forward_backward(const float* x, const float* t_h, const float* sc, const float* target, float* loss_dev, float* pred, hipStream_t s) {
    if (!train_mode || bwd.empty()) return fail(DDIF_ERR_STATE, "ddif_plan_train_forward_backward: not a train-mode plan");
    if (!cond_set) return fail(DDIF_ERR_STATE, "ddif_plan_train_forward_backward before ddif_plan_set_cond");
    if (!x || !t_h || !target) return fail(DDIF_ERR_INVALID, "ddif_plan_train_forward_backward: NULL argument");
    for (auto& kv : grad_slots)
        if (!kv.second) return fail(DDIF_ERR_STATE, "ddif_plan_train_forward_backward: gradient tensors are not bound; first missing: %s", kv.first.c_str());
    if (int e = check_sampler_net(net)) return e;
    const int HW = H * W;
    const size_t n = (size_t)B * HW * C;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, ew_grid(n), dim3(256), 0, s, x, B, C, HW, 0, C, x_in.p);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, ew_grid(n), dim3(256), 0, s, target, B, C, HW, 0, C, img[0]);
    if (sc) hipLaunchKernelGGL(nchw_to_nhwc_kernel, ew_grid(n), dim3(256), 0, s, sc, B, C, HW, 0, C, sc_in.p);
    return train_core(t_h, sc != nullptr, img[0], loss_dev, pred, s);
}

int Plan::train_core(const float* t_h, bool has_sc, const float* target_nhwc, float* loss_dev, float* pred, hipStream_t s) {
    const int HW = H * W;
    const size_t n = (size_t)B * HW * C;
    if (int e = time_rows_aux(t_h, B, taux, s)) return e;
    StepCtx ctx;
    ctx.x = x_in.p;
    ctx.sc = has_sc ? sc_in.p : x_in.p;
    ctx.tb = tb;
    ctx.tb_stride = net->nslots;
    train_set_stem_source(ctx.sc);
    run_prog(step, s, ctx, false);
    if (int e = train_backward(target_nhwc, 1.0f, loss_dev, s)) return e;
    if (pred) hipLaunchKernelGGL(nhwc_to_nchw_kernel, ew_grid(n), dim3(256), 0, s, (const float*)net_out.p, B, C, HW, pred);
    DDIF_HIPCHK(hipGetLastError());
    return 0;
}

void Plan::run_prog(std::vector<Op>& prog, hipStream_t s, const StepCtx& ctx, bool prof) {
    static const char* op_timing = getenv("DDIF_OP_TIMING");
    if (op_timing && prof && !op_timing_done) {  // development aid: every op of ONE step between its own pair of events
        op_timing_done = true;
        hipEvent_t e0, e1;
        if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
            if (FILE* f = fopen(op_timing, "w")) {
                fprintf(f, "op,kernel,us,gflop,mbytes\n");
                for (auto& op : prog) {
                    (void)hipEventRecord(e0, s);
                    op.run(s, ctx);
                    (void)hipEventRecord(e1, s);
                    (void)hipEventSynchronize(e1);
                    float ms = 0.f;
                    (void)hipEventElapsedTime(&ms, e0, e1);
                    fprintf(f, "\"%s\",%s,%.2f,%.4f,%.3f\n", op.label.empty() ? op.name : op.label.c_str(), op.name, ms * 1e3, op.flop / 1e9, op.bytes / 1e6);
                }
                fclose(f);
                (void)hipEventDestroy(e0);
                (void)hipEventDestroy(e1);
                return;
            }
        }
    }
    // a step is profiled completely or not at all: per-step figures divide by prof_steps
    if (prof && ev_used + (int)prog.size() > (int)ev0.size()) prof = false;
    if (prof) ++prof_steps;
    for (auto& op : prog) {
        const bool t = prof && (op.timed || prof_all) && ev_used < (int)ev0.size();
        if (t) (void)hipEventRecord(ev0[ev_used], s);
        op.run(s, ctx);
        if (t) {
            (void)hipEventRecord(ev1[ev_used], s);
            ev_flop[ev_used] = op.flop;
            ev_mflop[ev_used] = op.flop * op.mfma_w;
            ev_bytes[ev_used] = op.bytes;
            ev_cls[ev_used] = op.cls;
            ++ev_used;
        }
    }
}

// ------------------------------------------------------------------------------------------------ entry points
int Plan::set_cond(const float* cond, hipStream_t s) {
    if (!cond) return fail(DDIF_ERR_INVALID, "ddif_plan_set_cond: cond is NULL");
    cond_nchw = const_cast<float*>(cond);
    StepCtx ctx;
    if (train_mode && wg_async) {
        // encoder half (and everything shared) on the caller's stream, then the decoder-only half -- kv convs, contexts, padded copies for the weight
        // gradients -- on the side stream: it overlaps the stem / encoder / middle of the forward pass that follows, which joins in front of its
        // first decoder block.  (Encoder ops all precede decoder ops in `pre`; the side ops depend on the resized cond images only.)
        train_join(s);  // a previous set_cond's side work may still be reading what the ops below rewrite
        for (auto& op : pre)
            if (!op.side) op.run(s, ctx);
        hipStream_t ws = train_fork(s);
        for (auto& op : pre)
            if (op.side) op.run(ws, ctx);
        (void)hipEventRecord(side_read, ws);  // ddif_net_refresh waits for this before it rewrites the weight packs these ops read
        side_pending = true;
    } else {
        run_prog(pre, s, ctx, false);
    }
    cond_set = true;
    DDIF_HIPCHK(hipGetLastError());
    return 0;
}

int Plan::forward(const float* x, const float* t_host, const float* sc, float* out, hipStream_t s) {
    if (!cond_set) return fail(DDIF_ERR_STATE, "ddif_plan_forward before ddif_plan_set_cond");
    if (!x || !t_host || !out) return fail(DDIF_ERR_INVALID, "ddif_plan_forward: NULL argument");
    const int HW = H * W, Cx = net->cfg.in_channel;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, ew_grid((size_t)B * HW * Cx), dim3(256), 0, s, x, B, Cx, HW, 0, Cx, x_in.p);
    if (sc) hipLaunchKernelGGL(nchw_to_nhwc_kernel, ew_grid((size_t)B * HW * C), dim3(256), 0, s, sc, B, C, HW, 0, C, sc_in.p);
    if (int e = time_rows(t_host, B, s)) return e;
    StepCtx ctx;
    ctx.x = x_in.p;
    ctx.sc = sc ? sc_in.p : x_in.p;
    ctx.tb = tb;
    ctx.tb_stride = net->nslots;
    run_prog(step, s, ctx, false);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, ew_grid((size_t)B * HW * C), dim3(256), 0, s, (const float*)net_out.p, B, C, HW, out);
    DDIF_HIPCHK(hipGetLastError());
    return 0;
}

// The sticky range flag of the plan's raw-input f16x2 convs: read (and cleared) once per sampler / forward call by the caller, never inside a loop.
int Plan::range_status(hipStream_t s, int* overflow) {
    int v = 0;
    if (d_range && n_range_convs > 0) {
        DDIF_HIPCHK(hipMemcpyAsync(&v, d_range, sizeof(int), hipMemcpyDeviceToHost, s));
        DDIF_HIPCHK(hipStreamSynchronize(s));
        if (v) DDIF_HIPCHK(hipMemsetAsync(d_range, 0, sizeof(int), s));
    }
    if (overflow) *overflow = v ? 1 : 0;
    return 0;
}

static int check_sampler_net(const Net* n) {
    if (n->cfg.in_channel != n->cfg.out_channel)
        return fail(DDIF_ERR_INVALID, "samplers need in_channel == out_channel (x_start prediction over the image channels)");
    return 0;
}

// Shared loop of the DDPM (kind 0) and DDIM (kind 1) samplers.  Everything that changes from step to step is read
// from device memory (step counter, coefficient tables, SamplerRun), so two consecutive steps (img0 -> img1 -> img0)
// are captured ONCE into a hipGraph and replayed: ~270 launches per replay instead of per-kernel host launches.
int Plan::run_sampler(int kind, int n_steps, const float* const* tabs_host, int n_tabs, const float* t_model, const float* xT,
                      const float* noise, uint64_t seed, uint64_t tile0, float lo, float hi, int do_clamp, float* out, hipStream_t s) {
    if (int e = check_sampler_net(net)) return e;
    if (side_pending) {  // a train-mode plan: the cond-only side work must not be waited for inside a captured step
        train_join(s);
        side_pending = false;
    }
    const int HW = H * W;
    const size_t n = (size_t)B * HW * C;
    if (xT) hipLaunchKernelGGL(nchw_to_nhwc_kernel, ew_grid(n), dim3(256), 0, s, xT, B, C, HW, 0, C, img[0]);
    else hipLaunchKernelGGL(randn_nhwc_kernel, ew_grid(n), dim3(256), 0, s, img[0], B, C, HW, (unsigned long long)seed, 0u, (unsigned long long)tile0);
    if (int e = time_rows(t_model, n_steps, s)) return e;
    if (!d_step) {
        if (int e = dalloc(&d_step, 16)) return e;
        SamplerRun* r = nullptr;
        if (int e = dalloc(&r, 1)) return e;
        d_run = r;
    }
    if (tabs_cap < n_steps) {
        drop_graphs();
        if (int e = dalloc(&d_tabs, (size_t)6 * n_steps)) return e;
        tabs_cap = n_steps;
    }
    SamplerRun run{};
    run.noise = noise;
    run.seed = seed;
    run.tile0 = tile0;
    run.lo = lo;
    run.hi = hi;
    run.do_clamp = do_clamp;
    run.n_steps = n_steps;
    for (int i = 0; i < n_tabs; ++i) {
        run.tab[i] = d_tabs + (size_t)i * tabs_cap;
        DDIF_HIPCHK(hipMemcpyAsync(d_tabs + (size_t)i * tabs_cap, tabs_host[i], (size_t)n_steps * sizeof(float), hipMemcpyHostToDevice, s));
    }
    DDIF_HIPCHK(hipMemcpyAsync(d_run, &run, sizeof(run), hipMemcpyHostToDevice, s));
    DDIF_HIPCHK(hipMemsetAsync(d_step, 0, 2 * sizeof(int), s));
    // the host copies above must have been consumed before `run` (stack) goes away: pageable H2D copies are staged
    // synchronously by the runtime, so returning after the enqueue is safe.

    auto one_step = [&](int parity, hipStream_t st, bool prof) {
        StepCtx ctx;
        ctx.x = ctx.sc = img[parity];  // self-conditioning == current image (diffusion_ddpm_pan.py:491,502; sr3_dwt.py:173)
        ctx.tb = tb;
        ctx.tb_stride = 0;
        ctx.step_ptr = d_step + parity;
        ctx.step_next = d_step + (parity ^ 1);
        ctx.samp_run = d_run;
        ctx.samp_kind = kind;
        ctx.samp_out = final_fused ? img[parity ^ 1] : nullptr;
        ctx.tb_rowstride = net->nslots;
        run_prog(step, st, ctx, prof);
        StepArgs a{};
        a.x0 = net_out.p;
        a.img = img[parity];
        a.lms = lms.p;
        a.out = img[parity ^ 1];
        a.B = B;
        a.C = C;
        a.HW = HW;
        a.run = reinterpret_cast<const SamplerRun*>(d_run);
        if (final_fused) return;  // the update ran in the final conv's epilogue, which also wrote the next step's counter
        a.step = d_step + parity;
        if (kind == 0) hipLaunchKernelGGL(ddpm_step_kernel, ew_grid(n), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(ddim_step_kernel, ew_grid(n), dim3(256), 0, st, a);
        hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, st, (const int*)(d_step + parity), d_step + (parity ^ 1));
    };

    bool graph_ok = false;
    (void)graph_ok;
#ifndef DDIF_EMU
    static const bool graph_env = [] { const char* e = getenv("DDIF_GRAPH"); return !e || atoi(e) != 0; }();
    if (use_graph && graph_env && n_steps >= 4) {
        if (!graph_exec[kind]) {
            if (!cap_stream) DDIF_HIPCHK(hipStreamCreateWithFlags(&cap_stream, hipStreamNonBlocking));
            hipGraph_t g = nullptr;
            hipError_t e1 = hipStreamBeginCapture(cap_stream, hipStreamCaptureModeThreadLocal);
            if (e1 == hipSuccess) {
                one_step(0, cap_stream, false);
                one_step(1, cap_stream, false);
                e1 = hipStreamEndCapture(cap_stream, &g);
            }
            hipGraphExec_t ge = nullptr;
            if (e1 == hipSuccess && g) e1 = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            if (g) (void)hipGraphDestroy(g);
            if (e1 == hipSuccess && ge) graph_exec[kind] = ge;
            else {
                (void)hipGetLastError();
                use_graph = false;  // capture unsupported here: plain stream launches (same kernels, same results)
            }
        }
        graph_ok = graph_exec[kind] != nullptr;
    }
#endif
    int k = 0;
    while (k < n_steps) {
        const bool pair = k + 1 < n_steps;
        const bool prof = prof_every > 0 && ((k % prof_every) == 0 || (pair && ((k + 1) % prof_every) == 0));
        (void)prof;
#ifndef DDIF_EMU
        if (graph_ok && pair && !prof) {
            DDIF_HIPCHK(hipGraphLaunch((hipGraphExec_t)graph_exec[kind], s));
            k += 2;
            continue;
        }
#endif
        one_step(0, s, prof_every > 0 && (k % prof_every) == 0);
        ++k;
        if (pair) {
            one_step(1, s, prof_every > 0 && (k % prof_every) == 0);
            ++k;
        }
    }
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, ew_grid(n), dim3(256), 0, s, (const float*)img[n_steps & 1], B, C, HW, out);
    DDIF_HIPCHK(hipGetLastError());
    return 0;
}

// ---- train mode: dropout / DropPath masks
int Plan::train_set_dropout(int site, const float* mask_nchw, hipStream_t s) {
    if (!train_mode) return fail(DDIF_ERR_STATE, "not a train-mode plan (ddif_plan_create_train)");
    if (site < 0 || site >= (int)drop_sites.size() || !mask_nchw) return fail(DDIF_ERR_INVALID, "ddif_plan_train_set_dropout: bad site %d of %d", site, (int)drop_sites.size());
    const DropSite& d = drop_sites[site];
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, ew_grid((size_t)B * d.H * d.W * d.C), dim3(256), 0, s, mask_nchw, B, d.C, d.H * d.W, 0, d.C, d.mask);
    DDIF_HIPCHK(hipGetLastError());
    return 0;
}
int Plan::train_set_droppath(const float* scales_host, hipStream_t s) {
    if (!train_mode) return fail(DDIF_ERR_STATE, "not a train-mode plan (ddif_plan_create_train)");
    if (!scales_host) return fail(DDIF_ERR_INVALID, "ddif_plan_train_set_droppath: NULL");
    for (size_t k = 0; k < path_sites.size(); ++k)
        DDIF_HIPCHK(hipMemcpyAsync(path_sites[k], scales_host + k * B, (size_t)B * sizeof(float), hipMemcpyHostToDevice, s));
    return 0;
}
int Plan::train_get_dropout(int site, float* mask_nchw, hipStream_t s) {
    if (!train_mode) return fail(DDIF_ERR_STATE, "not a train-mode plan (ddif_plan_create_train)");
    if (site < 0 || site >= (int)drop_sites.size() || !mask_nchw) return fail(DDIF_ERR_INVALID, "ddif_plan_train_get_dropout: bad site %d of %d", site, (int)drop_sites.size());
    const DropSite& d = drop_sites[site];
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, ew_grid((size_t)B * d.H * d.W * d.C), dim3(256), 0, s, (const float*)d.mask, B, d.C, d.H * d.W, mask_nchw);
    DDIF_HIPCHK(hipGetLastError());
    return 0;
}
int Plan::train_get_droppath(float* scales_dev, hipStream_t s) {
    if (!train_mode) return fail(DDIF_ERR_STATE, "not a train-mode plan (ddif_plan_create_train)");
    if (!scales_dev) return fail(DDIF_ERR_INVALID, "ddif_plan_train_get_droppath: NULL");
    for (size_t k = 0; k < path_sites.size(); ++k)
        DDIF_HIPCHK(hipMemcpyAsync(scales_dev + k * B, path_sites[k], (size_t)B * sizeof(float), hipMemcpyDeviceToDevice, s));
    return 0;
}
int Plan::train_random_masks(uint64_t seed, uint64_t tile0, float p_drop, float p_path, hipStream_t s) {
    if (!train_mode) return fail(DDIF_ERR_STATE, "not a train-mode plan (ddif_plan_create_train)");
    if (!(p_drop >= 0.f && p_drop < 1.f && p_path >= 0.f && p_path < 1.f)) return fail(DDIF_ERR_INVALID, "drop probabilities must be in [0, 1)");
    if (!mask_recs) {  // the table of sites (fixed for the life of the plan): dropout sites first, DropPath sites after them, numbered in that order
        std::vector<MaskRec> tab;
        unsigned long long blk = 0;
        auto push = [&](float* m, int C_, int HW_, int path) {
            MaskRec r{};
            r.mask = m;
            r.C = C_;
            r.HW = HW_;
            r.site = (unsigned)tab.size();
            r.path = path;
            r.blk0 = blk;
            const size_t units = (HW_ & 3) == 0 ? (size_t)B * (HW_ >> 2) * C_ : (size_t)B * HW_ * C_;
            blk += (units + MASK_QPB - 1) / MASK_QPB;
            tab.push_back(r);
        };
        for (const DropSite& d : drop_sites) push(d.mask, d.C, d.H * d.W, 0);
        for (float* p : path_sites) push(p, 1, 1, 1);
        if (tab.empty()) return 0;
        float* raw = nullptr;
        if (int e = dalloc(&raw, (tab.size() * sizeof(MaskRec) + sizeof(float) - 1) / sizeof(float))) return e;
        DDIF_HIPCHK(hipMemcpy(raw, tab.data(), tab.size() * sizeof(MaskRec), hipMemcpyHostToDevice));
        mask_recs = raw;
        n_mask_recs = (int)tab.size();
        mask_blocks = blk;
    }
    hipLaunchKernelGGL(train_masks_kernel, dim3((unsigned)mask_blocks), dim3(256), 0, s, reinterpret_cast<const MaskRec*>(mask_recs), n_mask_recs, B,
                       (unsigned long long)seed, (unsigned long long)tile0, 1.f - p_drop, 1.f - p_path);
    DDIF_HIPCHK(hipGetLastError());
    return 0;
}

int Plan::sample_ddpm(const ddif_ddpm_tables* t, const float* xT, const float* noise, uint64_t seed, uint64_t tile0,
                      float lo, float hi, int do_clamp, float* out, hipStream_t s) {
    if (!cond_set) return fail(DDIF_ERR_STATE, "ddif_plan_sample_ddpm before ddif_plan_set_cond");
    if (!t || t->n_steps < 1 || !t->t_model || !t->coef_x0 || !t->coef_xt || !t->coef_z || !out)
        return fail(DDIF_ERR_INVALID, "ddif_plan_sample_ddpm: bad tables");
    const float* tabs[3] = {t->coef_x0, t->coef_xt, t->coef_z};
    return run_sampler(0, t->n_steps, tabs, 3, t->t_model, xT, noise, seed, tile0, lo, hi, do_clamp, out, s);
}

int Plan::sample_ddim(const ddif_ddim_tables* t, const float* xT, const float* noise, uint64_t seed, uint64_t tile0,
                      float lo, float hi, int do_clamp, float* out, hipStream_t s) {
    if (!cond_set) return fail(DDIF_ERR_STATE, "ddif_plan_sample_ddim before ddif_plan_set_cond");
    if (!t || t->n_steps < 1 || !t->t_model || !t->sqrt_recip || !t->sqrt_recipm1 || !t->sqrt_ap || !t->dir_coef || !t->sigma || !out)
        return fail(DDIF_ERR_INVALID, "ddif_plan_sample_ddim: bad tables");
    const float* tabs[5] = {t->sqrt_recip, t->sqrt_recipm1, t->sqrt_ap, t->dir_coef, t->sigma};
    return run_sampler(1, t->n_steps, tabs, 5, t->t_model, xT, noise, seed, tile0, lo, hi, do_clamp, out, s);
}

int Plan::sample_dpmpp(const ddif_dpm_tables* t, const float* xT, float lo, float hi, int do_clamp, float* out, hipStream_t s) {
    if (!cond_set) return fail(DDIF_ERR_STATE, "ddif_plan_sample_dpmpp before ddif_plan_set_cond");
    if (!t || t->n_evals < 1 || t->order < 1 || t->order > 3 || !t->t_model || !t->alpha || !t->sigma || !t->ord || !t->cx || !t->a_phi1 || !xT || !out)
        return fail(DDIF_ERR_INVALID, "ddif_plan_sample_dpmpp: bad tables");
    if (int e = check_sampler_net(net)) return e;
    if (side_pending) {
        train_join(s);
        side_pending = false;
    }
    const int HW = H * W;
    const size_t n = (size_t)B * HW * C;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, ew_grid(n), dim3(256), 0, s, xT, B, C, HW, 0, C, img[0]);
    if (int e = time_rows(t->t_model, t->n_evals, s)) return e;
    int cur = 0;
    float* hist[3] = {nullptr, nullptr, nullptr};  // newest first
    int nhist = 0, slot = 0;
    for (int k = 0; k < t->n_evals; ++k) {
        StepCtx ctx;
        ctx.x = ctx.sc = img[cur];  // model_wrapper never passes self_cond (dpm_solver.py:295) -> x
        ctx.tb = tb + (size_t)k * net->nslots;
        const bool prof = prof_every > 0 && (k % prof_every) == 0;
        run_prog(step, s, ctx, prof);
        float* mnew = mbuf[slot];
        slot = (slot + 1) % 3;
        hipLaunchKernelGGL(dpm_x0_kernel, ew_grid(n), dim3(256), 0, s, (const float*)net_out.p, (const float*)img[cur], (const float*)lms.p,
                           t->alpha[k], t->sigma[k], lo, hi, do_clamp, n, mnew);
        hist[2] = hist[1];
        hist[1] = hist[0];
        hist[0] = mnew;
        if (nhist < 3) ++nhist;
        const int o = t->ord[k];
        if (o < 1 || o > nhist) return fail(DDIF_ERR_INVALID, "ddif_plan_sample_dpmpp: update %d has order %d with %d model values", k, o, nhist);
        DpmUpdArgs u{};
        u.x = img[cur];
        u.m0 = hist[0];
        u.m1 = o >= 2 ? hist[1] : nullptr;
        u.m2 = o >= 3 ? hist[2] : nullptr;
        u.out = img[cur ^ 1];
        u.n = n;
        u.order = o;
        u.cx = t->cx[k];
        u.a1 = t->a_phi1[k];
        u.inv_r0 = (o >= 2 && t->inv_r0) ? t->inv_r0[k] : 0.f;
        u.inv_r1 = (o >= 3 && t->inv_r1) ? t->inv_r1[k] : 0.f;
        u.r0_frac = (o >= 3 && t->r0_frac) ? t->r0_frac[k] : 0.f;
        u.inv_r01 = (o >= 3 && t->inv_r01) ? t->inv_r01[k] : 0.f;
        u.a2 = (o >= 3 && t->a_phi2) ? t->a_phi2[k] : 0.f;
        u.a3 = (o >= 3 && t->a_phi3) ? t->a_phi3[k] : 0.f;
        hipLaunchKernelGGL(dpm_update_kernel, ew_grid(n), dim3(256), 0, s, u);
        cur ^= 1;
    }
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, ew_grid(n), dim3(256), 0, s, (const float*)img[cur], B, C, HW, out);
    DDIF_HIPCHK(hipGetLastError());
    return 0;
}

int Plan::q_sample_forward(const float* x0, const float* noise, const float* a_h, const float* s_h, const float* t_h,
                           const float* sc, float* pred, hipStream_t s) {
    if (!cond_set) return fail(DDIF_ERR_STATE, "ddif_plan_q_sample_forward before ddif_plan_set_cond");
    if (!x0 || !noise || !a_h || !s_h || !t_h || !pred) return fail(DDIF_ERR_INVALID, "ddif_plan_q_sample_forward: NULL argument");
    if (int e = check_sampler_net(net)) return e;
    const int HW = H * W;
    const size_t n = (size_t)B * HW * C;
    DDIF_HIPCHK(hipMemcpyAsync(small, a_h, (size_t)B * sizeof(float), hipMemcpyDefault, s));  // host or device source
    DDIF_HIPCHK(hipMemcpyAsync(small + B, s_h, (size_t)B * sizeof(float), hipMemcpyDefault, s));
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, ew_grid(n), dim3(256), 0, s, x0, B, C, HW, 0, C, img[0]);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, ew_grid(n), dim3(256), 0, s, noise, B, C, HW, 0, C, img[1]);
    hipLaunchKernelGGL(q_sample_kernel, ew_grid(n), dim3(256), 0, s, (const float*)img[0], (const float*)img[1], (const float*)small, (const float*)(small + B), B, (size_t)HW * C, x_in.p);
    if (sc) hipLaunchKernelGGL(nchw_to_nhwc_kernel, ew_grid(n), dim3(256), 0, s, sc, B, C, HW, 0, C, sc_in.p);
    if (int e = time_rows(t_h, B, s)) return e;
    StepCtx ctx;
    ctx.x = x_in.p;
    ctx.sc = sc ? sc_in.p : x_in.p;
    ctx.tb = tb;
    ctx.tb_stride = net->nslots;
    run_prog(step, s, ctx, false);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, ew_grid(n), dim3(256), 0, s, (const float*)net_out.p, B, C, HW, pred);
    DDIF_HIPCHK(hipGetLastError());
    return 0;
}

}  // namespace ddif
