// extern "C" surface of libddif (include/ddif.h).  No exceptions cross this boundary.
#include "ddif_plan.h"

struct ddif_net {
    ddif::Net n;
};
struct ddif_plan {
    ddif::Plan p;
};

namespace ddif {
extern int g_debug_grid_cap;
}

// Every entry point runs with the handle's device current and restores the caller's device afterwards: a process
// that drives several GPUs (test_fn(device="cuda:1") without torch.cuda.set_device) must not have the lazily
// allocated sampler state land on whatever device happened to be current.
struct DeviceScope {
    int prev = -1;
    bool ok = true;
    explicit DeviceScope(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceScope() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};
#define DDIF_PLAN_ENTER(plan, what)                                                                                   \
    if (!(plan)) return ddif::fail(DDIF_ERR_INVALID, what ": NULL plan");                                             \
    DeviceScope dscope__((plan)->p.net->device);                                                                      \
    if (!dscope__.ok) return ddif::fail(DDIF_ERR_HIP, what ": hipSetDevice(%d) failed", (plan)->p.net->device);      \
    if ((plan)->p.generation != (plan)->p.net->generation)                                                            \
        return ddif::fail(DDIF_ERR_STATE, what ": the network's weights were re-committed after this plan was built " \
                                               "(its launch program points into the old weight blob); create a new plan"); \
    if ((plan)->p.net->merged_stale && !(plan)->p.train_mode)                                                         \
        return ddif::fail(DDIF_ERR_STATE, what ": the weights were refreshed on the device for training (ddif_net_refresh), which leaves the "   \
                                               "inference-only merged FFN weights stale; load + commit the weights again before inference")

#define DDIF_GUARD_BEGIN try {
#define DDIF_GUARD_END                                                                    \
    }                                                                                     \
    catch (const std::exception& e) {                                                     \
        return ddif::fail(DDIF_ERR_INVALID, "unexpected C++ exception: %s", e.what());    \
    }                                                                                     \
    catch (...) {                                                                         \
        return ddif::fail(DDIF_ERR_INVALID, "unexpected C++ exception");                  \
    }

extern "C" {

int ddif_net_create(ddif_net_t* out, const ddif_net_cfg* cfg, int device) {
    DDIF_GUARD_BEGIN
    if (!out || !cfg) return ddif::fail(DDIF_ERR_INVALID, "ddif_net_create: NULL argument");
    *out = nullptr;
    DeviceScope ds(device);
    if (!ds.ok) return ddif::fail(DDIF_ERR_HIP, "ddif_net_create: hipSetDevice(%d) failed", device);
    std::unique_ptr<ddif_net> h(new ddif_net());
    h->n.cfg = *cfg;
    h->n.device = device;
    if (int e = h->n.build_layers()) return e;
    h->n.owner = h.get();
    *out = h.release();
    return DDIF_OK;
    DDIF_GUARD_END
}

void ddif_net_destroy(ddif_net_t net) {
    if (!net) return;
    {
        std::lock_guard<std::mutex> lk(net->n.life_mu);
        if (net->n.live_plans > 0) {  // plans point into the net (weight blobs, reader events): the last of them frees it (ddif_plan_destroy)
            net->n.orphaned = true;
            return;
        }
    }
    delete net;
}

int ddif_net_load(ddif_net_t net, const char* key, const float* data, const int64_t* shape, int ndim) {
    DDIF_GUARD_BEGIN
    if (!net) return ddif::fail(DDIF_ERR_INVALID, "ddif_net_load: NULL net");
    return net->n.load(key, data, shape, ndim);
    DDIF_GUARD_END
}

int ddif_net_commit(ddif_net_t net, void* stream) {
    DDIF_GUARD_BEGIN
    if (!net) return ddif::fail(DDIF_ERR_INVALID, "ddif_net_commit: NULL net");
    DeviceScope ds(net->n.device);
    if (!ds.ok) return ddif::fail(DDIF_ERR_HIP, "ddif_net_commit: hipSetDevice(%d) failed", net->n.device);
    return net->n.commit((hipStream_t)stream);
    DDIF_GUARD_END
}

int ddif_net_refresh(ddif_net_t net, int n, const char* const* keys, const float* const* params_dev, void* stream) {
    DDIF_GUARD_BEGIN
    if (!net) return ddif::fail(DDIF_ERR_INVALID, "ddif_net_refresh: NULL net");
    DeviceScope ds(net->n.device);
    if (!ds.ok) return ddif::fail(DDIF_ERR_HIP, "ddif_net_refresh: hipSetDevice(%d) failed", net->n.device);
    return net->n.refresh_device(n, keys, params_dev, (hipStream_t)stream);
    DDIF_GUARD_END
}

int64_t ddif_net_num_params(ddif_net_t net) { return net ? net->n.num_params() : 0; }

int ddif_plan_create(ddif_plan_t* out, ddif_net_t net, int B, int H, int W) {
    DDIF_GUARD_BEGIN
    if (!out || !net) return ddif::fail(DDIF_ERR_INVALID, "ddif_plan_create: NULL argument");
    *out = nullptr;
    DeviceScope ds(net->n.device);
    if (!ds.ok) return ddif::fail(DDIF_ERR_HIP, "ddif_plan_create: hipSetDevice(%d) failed", net->n.device);
    std::unique_ptr<ddif_plan> h(new ddif_plan());
    h->p.net = &net->n;
    h->p.generation = net->n.generation;
    h->p.B = B;
    h->p.H = H;
    h->p.W = W;
    if (int e = h->p.build()) return e;
    {
        std::lock_guard<std::mutex> lk(net->n.life_mu);
        ++net->n.live_plans;
    }
    *out = h.release();
    return DDIF_OK;
    DDIF_GUARD_END
}

int ddif_plan_create_train(ddif_plan_t* out, ddif_net_t net, int B, int H, int W) {
    DDIF_GUARD_BEGIN
    if (!out || !net) return ddif::fail(DDIF_ERR_INVALID, "ddif_plan_create_train: NULL argument");
    *out = nullptr;
    DeviceScope ds(net->n.device);
    if (!ds.ok) return ddif::fail(DDIF_ERR_HIP, "ddif_plan_create_train: hipSetDevice(%d) failed", net->n.device);
    std::unique_ptr<ddif_plan> h(new ddif_plan());
    h->p.net = &net->n;
    h->p.generation = net->n.generation;
    h->p.B = B;
    h->p.H = H;
    h->p.W = W;
    h->p.train_mode = true;
    h->p.use_graph = false;
    if (int e = h->p.build()) return e;
    // identity masks until the caller provides some: a train-mode plan then computes the eval network
    if (int e = h->p.train_random_masks(0, 0, 0.f, 0.f, nullptr)) return e;
    {
        std::lock_guard<std::mutex> lk(net->n.life_mu);
        ++net->n.live_plans;
    }
    *out = h.release();
    return DDIF_OK;
    DDIF_GUARD_END
}

int ddif_plan_train_bind(ddif_plan_t plan, int n, const char* const* keys, float* const* grads_dev) {
    DDIF_GUARD_BEGIN
    DDIF_PLAN_ENTER(plan, "ddif_plan_train_bind");
    return plan->p.train_bind(n, keys, grads_dev);
    DDIF_GUARD_END
}

int ddif_plan_train_num_grads(ddif_plan_t plan, int* n) {
    if (!plan || !plan->p.train_mode || !n) return ddif::fail(DDIF_ERR_STATE, "ddif_plan_train_num_grads: not a train-mode plan");
    *n = (int)plan->p.grad_slots.size();
    return DDIF_OK;
}

int ddif_plan_train_step(ddif_plan_t plan, const float* x0, const float* noise, const float* sqrt_ac_host, const float* sqrt_1mac_host, const float* time_host,
                         const float* self_cond, float* loss_dev, float* pred, void* stream) {
    DDIF_GUARD_BEGIN
    DDIF_PLAN_ENTER(plan, "ddif_plan_train_step");
    if (!plan->p.net->dgrad_filled) return ddif::fail(DDIF_ERR_STATE, "ddif_plan_train_step: the gradient-conv weight packs are still empty -- call ddif_net_refresh after creating the train-mode plan (and after every re-commit)");
    return plan->p.train_step(x0, noise, sqrt_ac_host, sqrt_1mac_host, time_host, self_cond, loss_dev, pred, (hipStream_t)stream);
    DDIF_GUARD_END
}

int ddif_plan_train_forward_backward(ddif_plan_t plan, const float* x, const float* time_host, const float* self_cond, const float* target, float* loss_dev,
                                     float* pred, void* stream) {
    DDIF_GUARD_BEGIN
    DDIF_PLAN_ENTER(plan, "ddif_plan_train_forward_backward");
    if (!plan->p.net->dgrad_filled) return ddif::fail(DDIF_ERR_STATE, "ddif_plan_train_forward_backward: the gradient-conv weight packs are still empty -- call ddif_net_refresh after creating the train-mode plan (and after every re-commit)");
    return plan->p.train_forward_backward(x, time_host, self_cond, target, loss_dev, pred, (hipStream_t)stream);
    DDIF_GUARD_END
}

int ddif_plan_train_info(ddif_plan_t plan, int* n_dropout_sites, int* n_droppath_sites) {
    if (!plan || !plan->p.train_mode) return ddif::fail(DDIF_ERR_STATE, "ddif_plan_train_info: not a train-mode plan");
    if (n_dropout_sites) *n_dropout_sites = (int)plan->p.drop_sites.size();
    if (n_droppath_sites) *n_droppath_sites = (int)plan->p.path_sites.size();
    return DDIF_OK;
}

int ddif_plan_train_site(ddif_plan_t plan, int site, int* C, int* H, int* W) {
    if (!plan || !plan->p.train_mode || site < 0 || site >= (int)plan->p.drop_sites.size()) return ddif::fail(DDIF_ERR_INVALID, "ddif_plan_train_site: bad plan / site");
    const auto& d = plan->p.drop_sites[site];
    if (C) *C = d.C;
    if (H) *H = d.H;
    if (W) *W = d.W;
    return DDIF_OK;
}

int ddif_plan_train_set_dropout(ddif_plan_t plan, int site, const float* mask, void* stream) {
    DDIF_GUARD_BEGIN
    DDIF_PLAN_ENTER(plan, "ddif_plan_train_set_dropout");
    return plan->p.train_set_dropout(site, mask, (hipStream_t)stream);
    DDIF_GUARD_END
}

int ddif_plan_train_set_droppath(ddif_plan_t plan, const float* scales_host, void* stream) {
    DDIF_GUARD_BEGIN
    DDIF_PLAN_ENTER(plan, "ddif_plan_train_set_droppath");
    return plan->p.train_set_droppath(scales_host, (hipStream_t)stream);
    DDIF_GUARD_END
}

int ddif_plan_train_get_dropout(ddif_plan_t plan, int site, float* mask, void* stream) {
    DDIF_GUARD_BEGIN
    DDIF_PLAN_ENTER(plan, "ddif_plan_train_get_dropout");
    return plan->p.train_get_dropout(site, mask, (hipStream_t)stream);
    DDIF_GUARD_END
}

int ddif_plan_train_get_droppath(ddif_plan_t plan, float* scales_dev, void* stream) {
    DDIF_GUARD_BEGIN
    DDIF_PLAN_ENTER(plan, "ddif_plan_train_get_droppath");
    return plan->p.train_get_droppath(scales_dev, (hipStream_t)stream);
    DDIF_GUARD_END
}

int ddif_plan_train_random_masks(ddif_plan_t plan, uint64_t seed, uint64_t tile0, float p_dropout, float p_droppath, void* stream) {
    DDIF_GUARD_BEGIN
    DDIF_PLAN_ENTER(plan, "ddif_plan_train_random_masks");
    return plan->p.train_random_masks(seed, tile0, p_dropout, p_droppath, (hipStream_t)stream);
    DDIF_GUARD_END
}

void ddif_plan_destroy(ddif_plan_t plan) {
    if (!plan) return;
    ddif::Net* n = plan->p.net;
    delete plan;
    ddif_net* last = nullptr;
    if (n) {
        std::lock_guard<std::mutex> lk(n->life_mu);
        if (--n->live_plans == 0 && n->orphaned) last = static_cast<ddif_net*>(n->owner);
    }
    delete last;  // outside the lock it lives in
}

int ddif_plan_set_cond(ddif_plan_t plan, const float* cond, void* stream) {
    DDIF_GUARD_BEGIN
    DDIF_PLAN_ENTER(plan, "ddif_plan call");
    return plan->p.set_cond(cond, (hipStream_t)stream);
    DDIF_GUARD_END
}

int ddif_plan_forward(ddif_plan_t plan, const float* x, const float* time_host, const float* self_cond, float* out, void* stream) {
    DDIF_GUARD_BEGIN
    DDIF_PLAN_ENTER(plan, "ddif_plan call");
    return plan->p.forward(x, time_host, self_cond, out, (hipStream_t)stream);
    DDIF_GUARD_END
}

int ddif_plan_sample_ddpm(ddif_plan_t plan, const ddif_ddpm_tables* tabs, const float* x_T, const float* noise, uint64_t seed,
                          uint64_t tile0, float clamp_lo, float clamp_hi, int do_clamp, float* out, void* stream) {
    DDIF_GUARD_BEGIN
    DDIF_PLAN_ENTER(plan, "ddif_plan call");
    return plan->p.sample_ddpm(tabs, x_T, noise, seed, tile0, clamp_lo, clamp_hi, do_clamp, out, (hipStream_t)stream);
    DDIF_GUARD_END
}

int ddif_plan_sample_ddim(ddif_plan_t plan, const ddif_ddim_tables* tabs, const float* x_T, const float* noise, uint64_t seed,
                          uint64_t tile0, float clamp_lo, float clamp_hi, int do_clamp, float* out, void* stream) {
    DDIF_GUARD_BEGIN
    DDIF_PLAN_ENTER(plan, "ddif_plan call");
    return plan->p.sample_ddim(tabs, x_T, noise, seed, tile0, clamp_lo, clamp_hi, do_clamp, out, (hipStream_t)stream);
    DDIF_GUARD_END
}

int ddif_plan_sample_dpmpp(ddif_plan_t plan, const ddif_dpm_tables* tabs, const float* x_T, float clamp_lo, float clamp_hi,
                           int do_clamp, float* out, void* stream) {
    DDIF_GUARD_BEGIN
    DDIF_PLAN_ENTER(plan, "ddif_plan call");
    return plan->p.sample_dpmpp(tabs, x_T, clamp_lo, clamp_hi, do_clamp, out, (hipStream_t)stream);
    DDIF_GUARD_END
}

int ddif_plan_q_sample_forward(ddif_plan_t plan, const float* x0, const float* noise, const float* sqrt_ac_host,
                               const float* sqrt_1mac_host, const float* time_host, const float* self_cond, float* pred,
                               void* stream) {
    DDIF_GUARD_BEGIN
    DDIF_PLAN_ENTER(plan, "ddif_plan call");
    return plan->p.q_sample_forward(x0, noise, sqrt_ac_host, sqrt_1mac_host, time_host, self_cond, pred, (hipStream_t)stream);
    DDIF_GUARD_END
}

int ddif_prof_begin(ddif_plan_t plan, int every_n_steps, int max_events) {
    DDIF_GUARD_BEGIN
    if (!plan || every_n_steps < 0 || max_events < 0) return ddif::fail(DDIF_ERR_INVALID, "ddif_prof_begin: bad arguments");
    ddif::Plan& p = plan->p;
    while ((int)p.ev0.size() < max_events) {
        hipEvent_t a, b;
        DDIF_HIPCHK(hipEventCreate(&a));
        DDIF_HIPCHK(hipEventCreate(&b));
        p.ev0.push_back(a);
        p.ev1.push_back(b);
    }
    p.ev_flop.assign(p.ev0.size(), 0.0);
    p.ev_bytes.assign(p.ev0.size(), 0.0);
    p.ev_mflop.assign(p.ev0.size(), 0.0);
    p.ev_cls.assign(p.ev0.size(), 5);
    p.ev_used = 0;
    p.prof_steps = 0;
    p.prof_all = true;
    p.prof_every = every_n_steps;
    p.prof_max = max_events;
    return DDIF_OK;
    DDIF_GUARD_END
}

int ddif_prof_collect(ddif_plan_t plan, ddif_prof_result* out) {
    DDIF_GUARD_BEGIN
    if (!plan || !out) return ddif::fail(DDIF_ERR_INVALID, "ddif_prof_collect: NULL argument");
    ddif::Plan& p = plan->p;
    std::memset(out, 0, sizeof(*out));
    for (int k = 0; k < 6; ++k) p.cls_res[k] = ddif_prof_class{};
    for (int i = 0; i < p.ev_used; ++i) {
        float ms = 0.f;
        DDIF_HIPCHK(hipEventSynchronize(p.ev1[i]));
        DDIF_HIPCHK(hipEventElapsedTime(&ms, p.ev0[i], p.ev1[i]));
        const int k = p.ev_cls[i] >= 0 && p.ev_cls[i] < 6 ? p.ev_cls[i] : 5;
        p.cls_res[k].launches += 1;
        p.cls_res[k].total_ms += ms;
        p.cls_res[k].total_flop += p.ev_flop[i];
        p.cls_res[k].total_bytes += p.ev_bytes[i];
        p.cls_res[k].total_mfma_flop += p.ev_mflop[i];
        if (k != 0) continue;  // the aggregate result is the dominant class (3x3 convs of the high-resolution levels)
        out->launches += 1;
        out->total_ms += ms;
        out->total_flop += p.ev_flop[i];
        out->total_bytes += p.ev_bytes[i];
        out->total_mfma_flop += p.ev_mflop[i];
    }
    if (p.n_conv3_b1 > 0)
        std::snprintf(out->kernel_name, sizeof(out->kernel_name), "ddif::conv_mfma_kernel<3,...> (3x3 implicit-GEMM convs; THROUGHPUT variant: one bf16 product on %d of %d)", p.n_conv3_b1,
                      p.n_conv3);
    else if (p.n_conv3_f16 > 0)
        std::snprintf(out->kernel_name, sizeof(out->kernel_name), "ddif::conv_mfma_kernel<3,...> (3x3 implicit-GEMM convs; f16x2 split products on %d of %d, bf16x3 on %d)", p.n_conv3_f16,
                      p.n_conv3, p.n_conv3_x3 - p.n_conv3_f16);
    else if (p.n_conv3_x3 > 0)
        std::snprintf(out->kernel_name, sizeof(out->kernel_name), "ddif::conv_mfma_kernel<3,...> (3x3 implicit-GEMM convolutions; bf16x3 split products on %d of %d)", p.n_conv3_x3, p.n_conv3);
    else
        std::snprintf(out->kernel_name, sizeof(out->kernel_name), "ddif::conv_mfma_kernel<3,...> (3x3 implicit-GEMM convolutions; exact fp32 MFMA)");
    out->steps_recorded = p.prof_steps;
    out->launches_per_step = (int64_t)p.step.size();
    p.prof_every = 0;
    p.ev_used = 0;
    p.prof_steps = 0;
    return DDIF_OK;
    DDIF_GUARD_END
}

int ddif_prof_classes(ddif_plan_t plan, ddif_prof_class* out6) {
    if (!plan || !out6) return ddif::fail(DDIF_ERR_INVALID, "ddif_prof_classes: NULL argument");
    static const char* names[6] = {"conv3x3 (> 256 px / sample)", "conv1x1 (> 256 px / sample)", "low-resolution levels (<= 256 px / sample)",
                                   "bottleneck attention", "softmax statistics", "other"};
    for (int k = 0; k < 6; ++k) {
        out6[k] = plan->p.cls_res[k];
        std::snprintf(out6[k].name, sizeof(out6[k].name), "%s", names[k]);
    }
    return DDIF_OK;
}

int ddif_plan_num_launches(ddif_plan_t plan, int* step_launches, int* cond_launches) {
    if (!plan) return ddif::fail(DDIF_ERR_INVALID, "NULL plan");
    if (step_launches) *step_launches = (int)plan->p.step.size();
    if (cond_launches) *cond_launches = (int)plan->p.pre.size();
    return DDIF_OK;
}

int ddif_plan_cost(ddif_plan_t plan, double* step_flop, double* step_bytes, double* cond_flop, double* cond_bytes) {
    DDIF_PLAN_ENTER(plan, "ddif_plan call");
    double sf = 0, sb = 0, cf = 0, cb = 0;
    for (auto& op : plan->p.step) {
        sf += op.flop;
        sb += op.bytes;
    }
    for (auto& op : plan->p.pre) {
        cf += op.flop;
        cb += op.bytes;
    }
    if (step_flop) *step_flop = sf;
    if (step_bytes) *step_bytes = sb;
    if (cond_flop) *cond_flop = cf;
    if (cond_bytes) *cond_bytes = cb;
    return DDIF_OK;
}

int ddif_plan_memory(ddif_plan_t plan, int64_t* total_bytes, int64_t* arena_bytes, int64_t* unaliased_bytes) {
    if (!plan) return ddif::fail(DDIF_ERR_INVALID, "NULL plan");
    if (total_bytes) *total_bytes = (int64_t)plan->p.bytes_allocated;
    if (arena_bytes) *arena_bytes = (int64_t)plan->p.arena_bytes;
    if (unaliased_bytes) *unaliased_bytes = (int64_t)plan->p.unaliased_bytes;
    return DDIF_OK;
}

int ddif_set_math_mode(int mode) {
    if (mode != DDIF_MATH_SPLIT && mode != DDIF_MATH_BF16) return ddif::fail(DDIF_ERR_INVALID, "math mode %d (DDIF_MATH_SPLIT or DDIF_MATH_BF16)", mode);
    ddif::g_math_mode = mode;
    return DDIF_OK;
}
int ddif_get_math_mode(void) { return ddif::g_math_mode; }

int ddif_set_f16_raw(int on) {
    ddif::g_f16_raw = on ? 1 : 0;
    return DDIF_OK;
}
int ddif_get_f16_raw(void) { return ddif::g_f16_raw; }
int ddif_plan_range_status(ddif_plan_t plan, void* stream, int* overflow) {
    DDIF_GUARD_BEGIN
    if (!plan || !overflow) return ddif::fail(DDIF_ERR_INVALID, "ddif_plan_range_status: NULL argument");
    DeviceScope ds(plan->p.net->device);
    if (!ds.ok) return ddif::fail(DDIF_ERR_HIP, "ddif_plan_range_status: hipSetDevice failed");
    return plan->p.range_status(reinterpret_cast<hipStream_t>(stream), overflow);
    DDIF_GUARD_END
}

int ddif_debug_set_grid_cap(int max_workgroups) {
    ddif::g_debug_grid_cap = max_workgroups > 0 ? max_workgroups : 0;
    return DDIF_OK;
}

const char* ddif_last_error(void) { return ddif::g_err.c_str(); }
const char* ddif_version(void) { return "ddif 0.1 (gfx950)"; }
int ddif_is_emulated(void) {
#ifdef DDIF_EMU
    return 1;
#else
    return 0;
#endif
}

}  // extern "C"
