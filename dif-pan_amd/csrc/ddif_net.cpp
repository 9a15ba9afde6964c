// Network construction and weight repacking (host side).
// Layer list = UNetSR3.__init__ (reference models/sr3_dwt.py:69-163); checkpoint keys = SURVEY.md appendix C.
#include "ddif_net.h"
#include <cstdint>
#include <tuple>
#include "kernels_refresh.h"

namespace ddif {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

static bool in_list(const int32_t* a, int n, int v) {
    for (int i = 0; i < n; ++i)
        if (a[i] == v) return true;
    return false;
}

int Net::build_layers() {
    const ddif_net_cfg& c = cfg;
    if (c.norm_groups != 1) return fail(DDIF_ERR_INVALID, "norm_groups=%d: only GroupNorm(1 group) is implemented", c.norm_groups);
    if (c.inner_channel != 32) return fail(DDIF_ERR_INVALID, "inner_channel=%d: only 32 is implemented", c.inner_channel);
    if (c.n_channel_mults < 1 || c.n_channel_mults > 8) return fail(DDIF_ERR_INVALID, "bad channel_mults");
    if (c.in_channel < 1 || c.out_channel < 1 || c.res_blocks < 1) return fail(DDIF_ERR_INVALID, "bad channel counts");
    const int inner = c.inner_channel;
    int pre = inner, res = c.image_size;
    std::vector<int> skips{pre};
    downs.clear();
    mid.clear();
    ups.clear();
    Layer stem;
    stem.kind = L_STEM;
    stem.cin = c.in_channel + (c.self_condition ? c.out_channel : 0);
    stem.cout = inner;
    stem.p = "downs.0";
    downs.push_back(stem);
    for (int i = 0; i < c.n_channel_mults; ++i) {
        const int ch = inner * c.channel_mults[i];
        const bool attn = in_list(c.attn_res, c.n_attn_res, res);
        for (int r = 0; r < c.res_blocks; ++r) {
            Layer L;
            L.kind = L_ENC;
            L.cin = pre;
            L.cout = ch;
            L.attn = attn;
            L.p = "downs." + std::to_string(downs.size());
            downs.push_back(L);
            skips.push_back(ch);
            pre = ch;
        }
        if (i != c.n_channel_mults - 1) {
            Layer L;
            L.kind = L_DOWN;
            L.cin = L.cout = pre;
            L.p = "downs." + std::to_string(downs.size());
            downs.push_back(L);
            skips.push_back(pre);
            res /= 2;
        }
    }
    for (int i = 0; i < 2; ++i) {
        Layer L;
        L.kind = L_MID;
        L.cin = L.cout = pre;
        L.attn = (i == 0);
        L.p = "mid." + std::to_string(i);
        mid.push_back(L);
    }
    for (int i = c.n_channel_mults - 1; i >= 0; --i) {
        const int ch = inner * c.channel_mults[i];
        const bool attn = in_list(c.attn_res, c.n_attn_res, res);
        for (int r = 0; r < c.res_blocks + 1; ++r) {
            Layer L;
            L.kind = L_DEC;
            L.cskip = skips.back();
            skips.pop_back();
            L.cx = pre;
            L.cin = pre + L.cskip;
            L.cout = ch;
            L.attn = attn;
            L.p = "ups." + std::to_string(ups.size());
            ups.push_back(L);
            pre = ch;
        }
        if (i >= 1) {
            Layer L;
            L.kind = L_UP;
            L.cin = L.cout = pre;
            L.p = "ups." + std::to_string(ups.size());
            ups.push_back(L);
            res *= 2;
        }
    }
    final_in = pre;
    auto check = [&](const Layer& L) -> int {
        if (L.attn && L.cout != 128)
            return fail(DDIF_ERR_INVALID, "%s: self-attention with %d channels (head dim %d): only 128 (head dim 16) is implemented",
                        L.p.c_str(), L.cout, L.cout / 8);
        if (L.kind == L_DEC && (L.cin % 8 != 0 || L.cin / 8 > 32))
            return fail(DDIF_ERR_INVALID, "%s: linear attention over %d channels needs 8 | channels and head dim <= 32", L.p.c_str(), L.cin);
        return 0;
    };
    for (auto& L : downs)
        if (int e = check(L)) return e;
    for (auto& L : mid)
        if (int e = check(L)) return e;
    for (auto& L : ups)
        if (int e = check(L)) return e;
    return 0;
}

int Net::load(const char* key, const float* data, const int64_t* shape, int ndim) {
    if (!key || !data || ndim < 0 || ndim > 4) return fail(DDIF_ERR_INVALID, "ddif_net_load: bad arguments");
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] < 0) return fail(DDIF_ERR_INVALID, "ddif_net_load(%s): negative dim", key);
        t.shape.push_back(shape[i]);
        n *= (size_t)shape[i];
    }
    t.v.assign(data, data + n);
    host[key] = std::move(t);
    committed = false;
    return 0;
}

int64_t Net::num_params() const {
    int64_t n = 0;
    for (auto& kv : host)
        if (kv.first != "noise_level_mlp.0.freqs") n += (int64_t)kv.second.v.size();
    return n;
}

namespace {

struct Blob {
    std::vector<float> v;
    size_t add(const float* p, size_t n) {
        size_t off = (v.size() + 63) & ~(size_t)63;  // 256-byte alignment
        v.resize(off + n, 0.f);
        if (p) std::memcpy(v.data() + off, p, n * sizeof(float));
        return off;
    }
};

// Kernel order (kernels_conv.h): [n-block of 32 couts][chunk of ck cins][tap][k8][half h][cout j][4 cins],
// cin = chunk*ck + k8*8 + 4*h + i.  Each wave's B fragment for one (tap, k8) is one contiguous 1 KiB read.
size_t pack_conv(Blob& b, const float* w, int cout, int cin, int ks, int ck, int* n_chunks_out) {
    const int n_chunks = (cin + ck - 1) / ck, nb = (cout + 31) / 32, nb_pad = (nb + 3) & ~3;
    const int taps = ks * ks, K8 = ck / 8;
    const size_t n = (size_t)nb_pad * n_chunks * taps * K8 * 256;
    const size_t off = b.add(nullptr, n);
    float* o = b.v.data() + off;
    for (int nbi = 0; nbi < nb; ++nbi)
        for (int ch = 0; ch < n_chunks; ++ch)
            for (int tap = 0; tap < taps; ++tap)
                for (int k8 = 0; k8 < K8; ++k8)
                    for (int h = 0; h < 2; ++h)
                        for (int j = 0; j < 32; ++j)
                            for (int i = 0; i < 4; ++i) {
                                const int ci = ch * ck + k8 * 8 + 4 * h + i, co = nbi * 32 + j;
                                float val = 0.f;
                                if (co < cout && ci < cin) val = w[((size_t)co * cin + ci) * taps + tap];
                                o[(((((size_t)nbi * n_chunks + ch) * taps + tap) * K8 + k8) * 2 + h) * 128 + j * 4 + i] = val;
                            }
    *n_chunks_out = n_chunks;
    return off;
}

// bf16x3 order (kernels_conv.h, MATH = 1; 3x3 convs with 16-channel chunks, 1x1 convs with 32-channel chunks):
//   [n-block of 32 couts][chunk][tap][16-channel slab k16][plane: hi, mid, lo][half h][cout j][8 bf16: cin = chunk*ck + 16*k16 + 8*h + t]
// = 1 KiB per (tap, plane) in exactly the per-lane A-operand order of v_mfma_f32_32x32x16_bf16.  The split is the same
// round-to-nearest-even three-way split the kernel applies to the activations (ddif_dev.h dd_split3).
inline unsigned bf16_bits_host(float x) {
    unsigned u;
    std::memcpy(&u, &x, 4);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
inline float bf16_val_host(unsigned b) {
    const unsigned u = b << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
size_t pack_conv_x3(Blob& b, const float* w, int cout, int cin, int ks, int ck) {
    const int taps = ks * ks, K16 = ck / 16;
    const int n_chunks = (cin + ck - 1) / ck, nb = (cout + 31) / 32, nb_pad = (nb + 3) & ~3;
    const size_t per = (size_t)taps * K16 * 3 * 256;  // floats per (n-block, chunk)
    const size_t off = b.add(nullptr, (size_t)nb_pad * n_chunks * per);
    uint16_t* o = reinterpret_cast<uint16_t*>(b.v.data() + off);
    for (int nbi = 0; nbi < nb; ++nbi)
        for (int ch = 0; ch < n_chunks; ++ch)
            for (int tap = 0; tap < taps; ++tap)
              for (int k16 = 0; k16 < K16; ++k16)
                for (int h = 0; h < 2; ++h)
                    for (int j = 0; j < 32; ++j)
                        for (int t = 0; t < 8; ++t) {
                            const int ci = ch * ck + 16 * k16 + 8 * h + t, co = nbi * 32 + j;
                            float val = 0.f;
                            if (co < cout && ci < cin) val = w[((size_t)co * cin + ci) * taps + tap];
                            unsigned q[3];
                            q[0] = bf16_bits_host(val);
                            const float r1 = val - bf16_val_host(q[0]);
                            q[1] = bf16_bits_host(r1);
                            q[2] = bf16_bits_host(r1 - bf16_val_host(q[1]));
                            for (int pl = 0; pl < 3; ++pl) {
                                const size_t fl = (((((size_t)nbi * n_chunks + ch) * taps + tap) * K16 + k16) * 3 + pl) * 256 + (size_t)(h * 32 + j) * 4;  // float index of the lane's 16 bytes
                                o[fl * 2 + t] = (uint16_t)q[pl];
                            }
                        }
    return off;
}

// f16x2 order (kernels_conv.h MATH = 3, kernels_lr.h F16; inference plans only -- never refreshed on the device):
//   [n-block of 32 couts][chunk][tap][16-channel slab k16][plane: hi, lo][half h][cout j][8 halves: cin = chunk*ck + 16*k16 + 8*h + t]
// of w * 2^10 (ddif_dev.h DDIF_F16_WSCALE): hi = half(w S) round to nearest even, lo = half(w S - hi).  Returns (size_t)-1 without
// packing when a weight is too large for the scaled half range (the conv then stays on bf16x3).
inline uint16_t f16_bits_host(float x) {
    const _Float16 hv = (_Float16)x;
    uint16_t b;
    std::memcpy(&b, &hv, 2);
    return b;
}
inline float f16_val_host(uint16_t b) {
    _Float16 hv;
    std::memcpy(&hv, &b, 2);
    return (float)hv;
}
size_t pack_conv_f16(Blob& b, const float* w, int cout, int cin, int ks, int ck) {
    const int taps = ks * ks, K16 = ck / 16;
    for (size_t i = 0; i < (size_t)cout * cin * taps; ++i)
        if (!(std::fabs(w[i]) <= DDIF_F16_WMAX)) return (size_t)-1;
    const int n_chunks = (cin + ck - 1) / ck, nb = (cout + 31) / 32, nb_pad = (nb + 3) & ~3;
    const size_t per = (size_t)taps * K16 * 2 * 256;  // floats per (n-block, chunk)
    const size_t off = b.add(nullptr, (size_t)nb_pad * n_chunks * per);
    uint16_t* o = reinterpret_cast<uint16_t*>(b.v.data() + off);
    for (int nbi = 0; nbi < nb; ++nbi)
        for (int ch = 0; ch < n_chunks; ++ch)
            for (int tap = 0; tap < taps; ++tap)
                for (int k16 = 0; k16 < K16; ++k16)
                    for (int h = 0; h < 2; ++h)
                        for (int j = 0; j < 32; ++j)
                            for (int t = 0; t < 8; ++t) {
                                const int ci = ch * ck + 16 * k16 + 8 * h + t, co = nbi * 32 + j;
                                float val = 0.f;
                                if (co < cout && ci < cin) val = w[((size_t)co * cin + ci) * taps + tap] * DDIF_F16_WSCALE;
                                uint16_t q[2];
                                q[0] = f16_bits_host(val);
                                q[1] = f16_bits_host(val - f16_val_host(q[0]));
                                for (int pl = 0; pl < 2; ++pl) {
                                    const size_t fl = (((((size_t)nbi * n_chunks + ch) * taps + tap) * K16 + k16) * 2 + pl) * 256 + (size_t)(h * 32 + j) * 4;
                                    o[fl * 2 + t] = q[pl];
                                }
                            }
    return off;
}

// bf16x1 order (kernels_conv.h MATH = 4, kernels_lr.h; the throughput variant, inference plans only): the hi plane of the bf16x3 pack on its own --
//   [n-block of 32 couts][chunk][tap][16-channel slab k16][half h][cout j][8 bf16]  = RNE bf16 of the weight.
size_t pack_conv_b1(Blob& b, size_t x3_off, int cout, int cin, int ks, int ck) {
    const size_t steps = (size_t)((((cout + 31) / 32 + 3) & ~3)) * ((cin + ck - 1) / ck) * (ks * ks) * (ck / 16);
    const size_t off = b.add(nullptr, steps * 256);
    for (size_t st = 0; st < steps; ++st) std::memcpy(b.v.data() + off + st * 256, b.v.data() + x3_off + st * 3 * 256, 256 * sizeof(float));
    return off;
}

}  // namespace

int Net::commit(hipStream_t stream) {
    Blob b;
    struct PendConv { std::string name; size_t w_off; long bias_off; int cin, cout, ks, ck, n_chunks; long x3_off = -1; long f16_off = -1; long b1_off = -1; };
    std::vector<PendConv> pend;
    std::map<std::string, size_t> vec_off;
    std::string missing;
    recipes.clear();
    auto rec_pack = [&](int kind, const std::string& s0, const std::string& s1, size_t off, int cout, int cin0, int cin1, int ks, int ck, int n_chunks) {
        Recipe r;
        r.kind = kind;
        r.src0 = s0;
        r.src1 = s1;
        r.dst_off = off;
        r.cout = cout;
        r.cin0 = cin0;
        r.cin1 = cin1;
        r.ks = ks;
        r.ck = ck;
        r.n_chunks = n_chunks;
        const size_t nb_pad = (size_t)(((cout + 31) / 32 + 3) & ~3);
        r.n_out = kind == RF_PACK_F32 ? nb_pad * n_chunks * ks * ks * (ck / 8) * 256 : nb_pad * n_chunks * ks * ks * (ck / 16) * 3 * 256;
        recipes.push_back(r);
    };
    auto rec_copy = [&](int kind, const std::string& s0, const std::string& s1, size_t off, size_t n, int C = 0) {
        Recipe r;
        r.kind = kind;
        r.src0 = s0;
        r.src1 = s1;
        r.dst_off = off;
        r.n_out = n;
        r.cout = C;
        recipes.push_back(r);
    };

    auto need = [&](const std::string& key) -> const HostTensor* {
        const HostTensor* t = get(key);
        if (!t && missing.empty()) missing = key;
        return t;
    };
    auto add_conv = [&](const std::string& name, bool has_bias) {
        const HostTensor* w = need(name + ".weight");
        const HostTensor* bs = has_bias ? need(name + ".bias") : nullptr;
        if (!w || (has_bias && !bs)) return;
        if (w->shape.size() != 4) { if (missing.empty()) missing = name + ".weight (expected 4-d)"; return; }
        PendConv p;
        p.name = name;
        p.cout = (int)w->shape[0];
        p.cin = (int)w->shape[1];
        p.ks = (int)w->shape[2];
        p.ck = (p.ks == 3 || p.cin <= 16) ? 16 : 32;  // 3x3: 16-channel chunks (A + W double-buffered = 65 KB of LDS)
        p.w_off = pack_conv(b, w->v.data(), p.cout, p.cin, p.ks, p.ck, &p.n_chunks);
        rec_pack(RF_PACK_F32, name + ".weight", "", p.w_off, p.cout, p.cin, 0, p.ks, p.ck, p.n_chunks);
        if (p.ck % 16 == 0 && (p.ks == 3 || p.ck == 32)) {
            p.x3_off = (long)pack_conv_x3(b, w->v.data(), p.cout, p.cin, p.ks, p.ck);
            rec_pack(RF_PACK_X3, name + ".weight", "", (size_t)p.x3_off, p.cout, p.cin, 0, p.ks, p.ck, p.n_chunks);
            p.f16_off = (long)pack_conv_f16(b, w->v.data(), p.cout, p.cin, p.ks, p.ck);  // inference only: no refresh recipe (merged_stale)
            p.b1_off = (long)pack_conv_b1(b, (size_t)p.x3_off, p.cout, p.cin, p.ks, p.ck);  // likewise
        }
        p.bias_off = bs ? (long)b.add(bs->v.data(), bs->v.size()) : -1;
        if (bs) rec_copy(RF_COPY, name + ".bias", "", (size_t)p.bias_off, bs->v.size());
        pend.push_back(p);
    };
    auto add_vec = [&](const std::string& key) {
        const HostTensor* t = need(key);
        if (t) {
            vec_off[key] = b.add(t->v.data(), t->v.size());
            rec_copy(RF_COPY, key, "", vec_off[key], t->v.size());
        }
    };
    auto add_dw = [&](const std::string& key) {  // (C,1,3,3) -> [9][C]
        const HostTensor* t = need(key);
        if (!t) return;
        const int C = (int)t->shape[0];
        std::vector<float> r((size_t)9 * C);
        for (int c = 0; c < C; ++c)
            for (int k = 0; k < 9; ++k) r[(size_t)k * C + c] = t->v[(size_t)c * 9 + k];
        vec_off[key] = b.add(r.data(), r.size());
        rec_copy(RF_DW, key, "", vec_off[key], r.size(), C);
    };
    auto add_resblock = [&](const std::string& p) {
        add_vec(p + ".block1.block.0.weight");
        add_vec(p + ".block1.block.0.bias");
        add_conv(p + ".block1.block.3", true);
        add_vec(p + ".block2.block.0.weight");
        add_vec(p + ".block2.block.0.bias");
        add_conv(p + ".block2.block.3", true);
    };
    auto add_attn = [&](const std::string& p) {
        add_vec(p + ".norm.weight");
        add_vec(p + ".norm.bias");
        add_conv(p + ".qkv", false);
        add_conv(p + ".out", true);
    };

    // time embedding: concatenated FeatureWiseAffine matrices
    std::vector<float> wall_h, ball_h;
    std::vector<std::tuple<std::string, size_t, size_t, size_t, size_t>> slot_keys;  // res_block prefix, offset in wall, offset in ball, sizes
    slot_off.clear();
    nslots = 0;
    auto add_slot = [&](const std::string& rb) {
        const HostTensor* w = need(rb + ".noise_func.noise_func.0.weight");
        const HostTensor* bs = need(rb + ".noise_func.noise_func.0.bias");
        if (!w || !bs) return;
        slot_off[rb] = nslots;
        slot_keys.emplace_back(rb, wall_h.size(), (size_t)nslots, w->v.size(), bs->v.size());
        wall_h.insert(wall_h.end(), w->v.begin(), w->v.end());
        ball_h.insert(ball_h.end(), bs->v.begin(), bs->v.end());
        nslots += (int)bs->v.size();
    };

    for (auto& L : downs) {
        if (L.kind == L_STEM) add_conv(L.p, true);
        else if (L.kind == L_DOWN) add_conv(L.p + ".conv", true);
        else {
            add_slot(L.p + ".res_block");
            add_resblock(L.p + ".res_block");
            if (L.attn) add_attn(L.p + ".attn");
            add_conv(L.p + ".cond_inj.body.0", false);
            add_vec(L.p + ".cond_inj.body.1.weight");
            add_vec(L.p + ".cond_inj.body.1.bias");
            add_conv(L.p + ".cond_inj.body.3", true);
            add_conv(L.p + ".cond_inj.x_conv", true);
            // round 6, inference only (no refresh recipe, like the f16 packs): the x_conv weights in the per-lane A-operand order of the register GEMM that folds
            // x_conv + FiLM into the PRODUCER's epilogue (kernels_conv.h EPI_XF): [32-cout block][lane = 32 h + j][s = 4 g + i] = W_x[32 nb + j][8 g + 4 h + i]
            if (const HostTensor* wx = get(L.p + ".cond_inj.x_conv.weight")) {
                if (wx->shape.size() == 4 && wx->shape[1] == 32 && wx->shape[0] % 32 == 0 && wx->shape[2] == 1) {
                    const int co = (int)wx->shape[0];
                    std::vector<float> r((size_t)co * 32);
                    for (int nb = 0; nb < co / 32; ++nb)
                        for (int ln = 0; ln < 64; ++ln)
                            for (int sx = 0; sx < 16; ++sx)
                                r[((size_t)nb * 64 + ln) * 16 + sx] = wx->v[(size_t)(32 * nb + (ln & 31)) * 32 + 8 * (sx >> 2) + 4 * (ln >> 5) + (sx & 3)];
                    vec_off[L.p + ".cond_inj.x_conv.xf"] = b.add(r.data(), r.size());
                }
            }
        }
    }
    for (auto& L : mid) {
        add_slot(L.p + ".res_block");
        add_resblock(L.p + ".res_block");
        if (L.attn) add_attn(L.p + ".attn");
    }
    for (auto& L : ups) {
        if (L.kind == L_UP) { add_conv(L.p + ".conv", true); continue; }
        add_slot(L.p + ".res_block");
        add_resblock(L.p + ".res_block");
        if (L.attn) add_attn(L.p + ".attn");
        const std::string ci = L.p + ".cond_inj";
        add_vec(ci + ".prenorm_x.weight");
        add_vec(ci + ".prenorm_x.bias");
        add_dw(ci + ".q.0.weight");
        add_conv(ci + ".q.1", true);
        add_dw(ci + ".kv.0.weight");
        add_conv(ci + ".kv.1", true);
        // attn_out(o) + attn_res(xn) as ONE 1x1 conv over cat[o, xn] (K = 2*fea); biases summed
        const HostTensor *wo = need(ci + ".attn_out.weight"), *bo = need(ci + ".attn_out.bias");
        const HostTensor* wr = get(ci + ".attn_res.weight");
        const HostTensor* br = get(ci + ".attn_res.bias");
        if (wo) {
            vec_off[ci + ".attn_out.weight"] = b.add(wo->v.data(), wo->v.size());
            rec_copy(RF_COPY, ci + ".attn_out.weight", "", vec_off[ci + ".attn_out.weight"], wo->v.size());
        }
        if (wr) {
            vec_off[ci + ".attn_res.weight"] = b.add(wr->v.data(), wr->v.size());
            rec_copy(RF_COPY, ci + ".attn_res.weight", "", vec_off[ci + ".attn_res.weight"], wr->v.size());
        }
        if (wo && bo) {
            const int co = (int)wo->shape[0], fea = (int)wo->shape[1];
            PendConv p;
            p.name = ci + ".attn_mix";
            p.cout = co;
            p.ks = 1;
            std::vector<float> bsum(bo->v);
            if (wr && br) {
                std::vector<float> cat((size_t)co * 2 * fea);
                for (int o = 0; o < co; ++o) {
                    std::memcpy(&cat[(size_t)o * 2 * fea], &wo->v[(size_t)o * fea], fea * sizeof(float));
                    std::memcpy(&cat[(size_t)o * 2 * fea + fea], &wr->v[(size_t)o * fea], fea * sizeof(float));
                    bsum[o] = bo->v[o] + br->v[o];
                }
                p.cin = 2 * fea;
                p.ck = 32;
                p.w_off = pack_conv(b, cat.data(), co, p.cin, 1, p.ck, &p.n_chunks);
                rec_pack(RF_PACK_F32, ci + ".attn_out.weight", ci + ".attn_res.weight", p.w_off, co, fea, fea, 1, p.ck, p.n_chunks);
                // (train-mode plans run this conv with the SHARED weights -- the per-sample folded copies belong to inference -- on the bf16x3 path)
                p.x3_off = (long)pack_conv_x3(b, cat.data(), co, p.cin, 1, p.ck);
                rec_pack(RF_PACK_X3, ci + ".attn_out.weight", ci + ".attn_res.weight", (size_t)p.x3_off, co, fea, fea, 1, p.ck, p.n_chunks);
                p.b1_off = (long)pack_conv_b1(b, (size_t)p.x3_off, co, p.cin, 1, p.ck);
            } else {  // attn_res is Identity (fea == dim_out): xn is added as a residual
                p.cin = fea;
                p.ck = fea <= 16 ? 16 : 32;
                p.w_off = pack_conv(b, wo->v.data(), co, fea, 1, p.ck, &p.n_chunks);
                rec_pack(RF_PACK_F32, ci + ".attn_out.weight", "", p.w_off, co, fea, 0, 1, p.ck, p.n_chunks);
                if (p.ck == 32) {
                    p.x3_off = (long)pack_conv_x3(b, wo->v.data(), co, fea, 1, p.ck);
                    rec_pack(RF_PACK_X3, ci + ".attn_out.weight", "", (size_t)p.x3_off, co, fea, 0, 1, p.ck, p.n_chunks);
                    p.b1_off = (long)pack_conv_b1(b, (size_t)p.x3_off, co, fea, 1, p.ck);
                }
            }
            p.bias_off = (long)b.add(bsum.data(), bsum.size());
            if (wr && br) rec_copy(RF_SUM, ci + ".attn_out.bias", ci + ".attn_res.bias", (size_t)p.bias_off, bsum.size());
            else rec_copy(RF_COPY, ci + ".attn_out.bias", "", (size_t)p.bias_off, bsum.size());
            pend.push_back(p);
        }
        add_conv(ci + ".ffn.0", false);
        add_conv(ci + ".ffn.2", false);
        add_conv(ci + ".ffn.3", true);
        // ffn[2] (conv3x3, no bias) and ffn[3] (conv1x1 + bias) have NO nonlinearity between them (models/sr3_dwt.py:530-533): in eval
        // mode they are ONE 3x3 conv with W'[o][i][tap] = sum_k W3[o][k] W2[k][i][tap] (formed in fp64, rounded once) and bias b3.
        // Same algebra as the reference, one fp32 rounding pattern instead of two; removes 16 launches per denoising step.
        {
            const HostTensor *w2 = get(ci + ".ffn.2.weight"), *w3 = get(ci + ".ffn.3.weight"), *b3 = get(ci + ".ffn.3.bias");
            if (w2 && w3 && b3 && w2->shape.size() == 4 && w3->shape.size() == 4 && w3->shape[1] == w2->shape[0] && w3->shape[2] == 1) {
                const int co = (int)w3->shape[0], mid = (int)w2->shape[0], cin = (int)w2->shape[1];
                std::vector<float> wm((size_t)co * cin * 9);
                std::vector<double> row((size_t)cin * 9);
                for (int o = 0; o < co; ++o) {
                    std::fill(row.begin(), row.end(), 0.0);
                    for (int k = 0; k < mid; ++k) {
                        const double a = (double)w3->v[(size_t)o * mid + k];
                        const float* src = &w2->v[(size_t)k * cin * 9];
                        for (int e = 0; e < cin * 9; ++e) row[e] += a * (double)src[e];
                    }
                    for (int e = 0; e < cin * 9; ++e) wm[(size_t)o * cin * 9 + e] = (float)row[e];
                }
                PendConv p;
                p.name = ci + ".ffn.23";
                p.cout = co;
                p.cin = cin;
                p.ks = 3;
                p.ck = 16;
                p.w_off = pack_conv(b, wm.data(), co, cin, 3, p.ck, &p.n_chunks);
                p.x3_off = (long)pack_conv_x3(b, wm.data(), co, cin, 3, p.ck);
                p.f16_off = (long)pack_conv_f16(b, wm.data(), co, cin, 3, p.ck);
                p.b1_off = (long)pack_conv_b1(b, (size_t)p.x3_off, co, cin, 3, p.ck);
                p.bias_off = (long)b.add(b3->v.data(), b3->v.size());
                pend.push_back(p);
            }
        }
    }
    add_vec("final_conv.block.0.weight");
    add_vec("final_conv.block.0.bias");
    add_conv("final_conv.block.3", true);

    const HostTensor *hw1 = need("noise_level_mlp.1.weight"), *hb1 = need("noise_level_mlp.1.bias");
    const HostTensor *hw3 = need("noise_level_mlp.3.weight"), *hb3 = need("noise_level_mlp.3.bias");
    if (!missing.empty()) return fail(DDIF_ERR_MISSING, "ddif_net_commit: weight '%s' was never loaded", missing.c_str());

    const int half = cfg.inner_channel / 2;
    std::vector<float> fr(half);
    if (const HostTensor* f = get("noise_level_mlp.0.freqs")) {
        if ((int)f->v.size() != half) return fail(DDIF_ERR_INVALID, "noise_level_mlp.0.freqs must have %d entries", half);
        fr = f->v;
    } else {
        const float k = -(float)std::log(1e4);
        for (int j = 0; j < half; ++j) fr[j] = std::exp(k * ((float)j / (float)half));
    }
    const size_t o_fr = b.add(fr.data(), fr.size());
    const size_t o_w1 = b.add(hw1->v.data(), hw1->v.size()), o_b1 = b.add(hb1->v.data(), hb1->v.size());
    const size_t o_w3 = b.add(hw3->v.data(), hw3->v.size()), o_b3 = b.add(hb3->v.data(), hb3->v.size());
    const size_t o_wall = b.add(wall_h.data(), wall_h.size()), o_ball = b.add(ball_h.data(), ball_h.size());
    rec_copy(RF_COPY, "noise_level_mlp.1.weight", "", o_w1, hw1->v.size());
    rec_copy(RF_COPY, "noise_level_mlp.1.bias", "", o_b1, hb1->v.size());
    rec_copy(RF_COPY, "noise_level_mlp.3.weight", "", o_w3, hw3->v.size());
    rec_copy(RF_COPY, "noise_level_mlp.3.bias", "", o_b3, hb3->v.size());
    for (auto& sk : slot_keys) {
        rec_copy(RF_COPY, std::get<0>(sk) + ".noise_func.noise_func.0.weight", "", o_wall + std::get<1>(sk), std::get<3>(sk));
        rec_copy(RF_COPY, std::get<0>(sk) + ".noise_func.noise_func.0.bias", "", o_ball + std::get<2>(sk), std::get<4>(sk));
    }
    merged_stale = false;
    last_ptrs.clear();
    refresh_keys.clear();
    dconv.clear();  // re-derived on demand (build_dgrad_packs) from the recipes of THIS commit
    dgrad_filled = false;
    // plans of the previous generation may still have launches in flight on their own (side) streams that read the blobs freed below
    // (their own streams and events only -- not hipDeviceSynchronize, which fails while ANY stream of the process is being captured; the hipFree
    //  calls below wait for whatever else still uses the allocations)
    if (blob || dgrad_blob) {
        for (hipEvent_t e : reader_events) DDIF_HIPCHK(hipEventSynchronize(e));
        DDIF_HIPCHK(hipStreamSynchronize(stream));
    }
    if (dgrad_blob) {
        DDIF_HIPCHK(hipFree(dgrad_blob));
        dgrad_blob = nullptr;
    }

    if (blob) {
        DDIF_HIPCHK(hipFree(blob));
        blob = nullptr;
    }
    blob_floats = b.v.size() + 64;
    DDIF_HIPCHK(hipMalloc((void**)&blob, blob_floats * sizeof(float)));
    DDIF_HIPCHK(hipMemcpy(blob, b.v.data(), b.v.size() * sizeof(float), hipMemcpyHostToDevice));
    (void)stream;

    conv.clear();
    vec.clear();
    for (auto& p : pend) {
        PackedConv pc;
        pc.w = blob + p.w_off;
        pc.w_x3 = p.x3_off >= 0 ? blob + p.x3_off : nullptr;
        pc.w_f16 = p.f16_off >= 0 ? blob + p.f16_off : nullptr;
        pc.w_b1 = p.b1_off >= 0 ? blob + p.b1_off : nullptr;
        pc.bias = p.bias_off >= 0 ? blob + p.bias_off : nullptr;
        pc.cin = p.cin;
        pc.cout = p.cout;
        pc.ks = p.ks;
        pc.ck = p.ck;
        pc.n_chunks = p.n_chunks;
        conv[p.name] = pc;
    }
    vec_absmax.clear();
    for (auto& kv : vec_off) {
        vec[kv.first] = blob + kv.second;
        float m = 0.f;
        if (const HostTensor* t = get(kv.first))
            for (float x : t->v) m = std::fabs(x) > m || std::isnan(x) ? (std::isnan(x) ? INFINITY : std::fabs(x)) : m;
        vec_absmax[blob + kv.second] = m;
    }
    freqs = blob + o_fr;
    w1 = blob + o_w1;
    b1 = blob + o_b1;
    w3 = blob + o_w3;
    b3 = blob + o_b3;
    wall = blob + o_wall;
    ball = blob + o_ball;
    committed = true;
    ++generation;
    return 0;
}

// Rewrites every packed tensor of the blob in place from DEVICE parameter tensors (reference layouts), one launch.  Plans built
// on this net stay valid (no pointer moves, same generation).  The eval-only merged ffn weights are left stale (merged_stale).
int Net::refresh_device(int n, const char* const* keys, const float* const* ptrs, hipStream_t stream) {
    if (!committed) return fail(DDIF_ERR_STATE, "ddif_net_refresh: ddif_net_commit has not been called");
    if (n < 1 || !keys || !ptrs) return fail(DDIF_ERR_INVALID, "ddif_net_refresh: bad arguments");
#ifndef DDIF_EMU
    for (hipEvent_t e : reader_events) (void)hipStreamWaitEvent(stream, e, 0);  // (a never-recorded or completed event does not block)
#endif
    // the training loop calls this every iteration with the same keys and (almost always) the same pointers: the key -> pointer map of 702 strings is only
    // rebuilt when either moved (the call sits in front of every iteration's first launch, with the GPU idle behind it)
    bool same = (int)refresh_keys.size() == n && d_recs != nullptr;
    for (int i = 0; i < n && same; ++i) {
        if (!keys[i] || !ptrs[i]) return fail(DDIF_ERR_INVALID, "ddif_net_refresh: NULL key / pointer at %d", i);
        same = refresh_in_ptrs[i] == ptrs[i] && refresh_keys[i] == keys[i];
    }
    if (same) {
        hipLaunchKernelGGL(refresh_blob_kernel, dim3((unsigned)refresh_blocks), dim3(256), 0, stream, (const RefreshRec*)d_recs, n_recs);
        DDIF_HIPCHK(hipGetLastError());
        merged_stale = true;
        if (dgrad_blob) dgrad_filled = true;
        return 0;
    }
    std::map<std::string, const float*> by_key;
    for (int i = 0; i < n; ++i) {
        if (!keys[i] || !ptrs[i]) return fail(DDIF_ERR_INVALID, "ddif_net_refresh: NULL key / pointer at %d", i);
        by_key[keys[i]] = ptrs[i];
    }
    refresh_keys.clear();  // (set again below, once the table is known to match these arguments)
    std::vector<const float*> flat;
    flat.reserve(recipes.size() * 2);
    for (auto& r : recipes) {
        auto i0 = by_key.find(r.src0);
        if (i0 == by_key.end()) return fail(DDIF_ERR_MISSING, "ddif_net_refresh: parameter '%s' was not given", r.src0.c_str());
        const float* p1 = nullptr;
        if (!r.src1.empty()) {
            auto i1 = by_key.find(r.src1);
            if (i1 == by_key.end()) return fail(DDIF_ERR_MISSING, "ddif_net_refresh: parameter '%s' was not given", r.src1.c_str());
            p1 = i1->second;
        }
        flat.push_back(i0->second);
        flat.push_back(p1);
    }
    if (flat != last_ptrs || !d_recs) {  // (re)build the device table: only when the parameter tensors moved
        std::vector<RefreshRec> recs(recipes.size());
        long long blk = 0;
        for (size_t k = 0; k < recipes.size(); ++k) {
            const Recipe& r = recipes[k];
            RefreshRec& d = recs[k];
            d.kind = r.kind;
            d.src0 = flat[2 * k];
            d.src1 = flat[2 * k + 1];
            d.dst = (r.dblob ? dgrad_blob : blob) + r.dst_off;
            d.tr = r.tr;
            d.fc0 = r.fc0;
            d.fc1 = r.fc1;
            d.cout = r.cout;
            d.cin0 = r.cin0;
            d.cin1 = r.cin1;
            d.ks = r.ks;
            d.ck = r.ck;
            d.n_chunks = r.n_chunks;
            d.n_out = (long long)r.n_out;
            d.blk0 = blk;
            blk += (long long)((r.n_out + 255) / 256);
        }
        if (d_recs) {
            DDIF_HIPCHK(hipStreamSynchronize(stream));  // an earlier refresh may still read the old table
            DDIF_HIPCHK(hipFree(d_recs));
            d_recs = nullptr;
        }
        DDIF_HIPCHK(hipMalloc(&d_recs, recs.size() * sizeof(RefreshRec)));
        DDIF_HIPCHK(hipMemcpy(d_recs, recs.data(), recs.size() * sizeof(RefreshRec), hipMemcpyHostToDevice));
        n_recs = (int)recs.size();
        refresh_blocks = blk;
        last_ptrs = flat;
    }
    refresh_keys.assign(keys, keys + n);
    refresh_in_ptrs.assign(ptrs, ptrs + n);
    hipLaunchKernelGGL(refresh_blob_kernel, dim3((unsigned)refresh_blocks), dim3(256), 0, stream, (const RefreshRec*)d_recs, n_recs);
    DDIF_HIPCHK(hipGetLastError());
    merged_stale = true;
    if (dgrad_blob) dgrad_filled = true;
    return 0;
}

// Training: allocate + describe the dgrad packs of every conv the reverse pass differentiates through (called once by the first
// train-mode plan after a commit).  The packs are FILLED by refresh_device() -- a train-mode plan therefore needs one refresh before
// its first backward pass.
int Net::build_dgrad_packs() {
    if (!committed) return fail(DDIF_ERR_STATE, "build_dgrad_packs: ddif_net_commit has not been called");
    if (dgrad_blob) return 0;
    size_t off = 0;
    std::vector<Recipe> extra;
    auto add = [&](const std::string& name, const std::string& s0, const std::string& s1, int cout_f, int fc0, int fc1, int ks) {
        // dgrad conv: contraction over the forward couts, output = the forward cins (padded to a multiple of 4 for float4 stores)
        PackedConv pc;
        pc.cin = cout_f;
        pc.cout = (fc0 + fc1 + 3) & ~3;
        pc.ks = ks;
        pc.ck = (ks == 3 || pc.cin <= 16) ? 16 : 32;
        pc.n_chunks = (pc.cin + pc.ck - 1) / pc.ck;
        const size_t nb_pad = (size_t)(((pc.cout + 31) / 32 + 3) & ~3);
        const bool x3 = pc.ck % 16 == 0 && (ks == 3 || pc.ck == 32);
        Recipe r;
        r.kind = RF_PACK_F32;
        r.src0 = s0;
        r.src1 = s1;
        r.cout = pc.cout;
        r.cin0 = cout_f;
        r.ks = ks;
        r.ck = pc.ck;
        r.n_chunks = pc.n_chunks;
        r.tr = 1;
        r.fc0 = fc0;
        r.fc1 = fc1;
        r.dblob = true;
        r.dst_off = off;
        r.n_out = nb_pad * pc.n_chunks * ks * ks * (pc.ck / 8) * 256;
        const size_t w_off = off;
        off += (r.n_out + 63) & ~(size_t)63;
        extra.push_back(r);
        size_t x3_off = 0;
        if (x3) {
            Recipe q = r;
            q.kind = RF_PACK_X3;
            q.dst_off = off;
            q.n_out = nb_pad * pc.n_chunks * ks * ks * (pc.ck / 16) * 3 * 256;
            x3_off = off;
            off += (q.n_out + 63) & ~(size_t)63;
            extra.push_back(q);
        }
        // pointers are fixed up below, once the blob exists
        pc.w = reinterpret_cast<const float*>(w_off + 1);
        pc.w_x3 = x3 ? reinterpret_cast<const float*>(x3_off + 1) : nullptr;
        pc.bias = nullptr;
        dconv[name] = pc;
    };
    for (auto& kv : conv) {
        const std::string& name = kv.first;
        const PackedConv& f = kv.second;
        const size_t n = name.size();
        auto ends = [&](const char* suf) { const size_t m = strlen(suf); return n >= m && name.compare(n - m, m, suf) == 0; };
        if (ends(".ffn.23") || ends(".body.0") || name == "downs.0") continue;  // eval-only merge; convs whose input needs no gradient
        if (ends(".attn_mix")) {
            const std::string ci = name.substr(0, n - strlen(".attn_mix"));
            const bool has_res = get(ci + ".attn_res.weight") != nullptr;
            const int fea = has_res ? f.cin / 2 : f.cin;
            add(name, ci + ".attn_out.weight", has_res ? ci + ".attn_res.weight" : "", f.cout, fea, has_res ? fea : 0, 1);
            continue;
        }
        add(name, name + ".weight", "", f.cout, f.cin, 0, f.ks);
    }
    dgrad_floats = off + 64;
    DDIF_HIPCHK(hipMalloc((void**)&dgrad_blob, dgrad_floats * sizeof(float)));
    DDIF_HIPCHK(hipMemset(dgrad_blob, 0, dgrad_floats * sizeof(float)));
    dgrad_filled = false;
    for (auto& kv : dconv) {
        PackedConv& pc = kv.second;
        pc.w = dgrad_blob + (reinterpret_cast<size_t>(pc.w) - 1);
        if (pc.w_x3) pc.w_x3 = dgrad_blob + (reinterpret_cast<size_t>(pc.w_x3) - 1);
    }
    recipes.insert(recipes.end(), extra.begin(), extra.end());
    last_ptrs.clear();  // the device table must be rebuilt
    refresh_keys.clear();
    return 0;
}

}  // namespace ddif
