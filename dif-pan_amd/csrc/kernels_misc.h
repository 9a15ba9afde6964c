// The non-GEMM kernels of the DDIF denoising step (NHWC fp32, gfx950): depthwise 3x3 with GroupNorm prologue,
// linear-attention softmax statistics / context / apply, bottleneck self-attention, time embedding, cond resize,
// layout conversion and the sampler update kernels.  Reference line numbers are for models/sr3_dwt.py and
// diffusion/diffusion_ddpm_pan.py.
#pragma once
#include "ddif_dev.h"
#include "sampler_dev.h"

namespace ddif {

// ----------------------------------------------------------------------------------------------------------------
// depthwise 3x3 (FastAttnCondInjection.q[0] / kv[0], sr3_dwt.py:510-517) over xn = GroupNorm(cat[in0, in1])
// (prenorm_x, :507,537).  Also writes xn itself (consumed by attn_res, :573).  8x16 pixel tile, 32-channel chunks.
struct DwArgs {
    const float* in0;
    const float* in1;
    int c0, c1;
    int B, H, W;
    const double* st0;
    int np0;
    const double* st1;
    int np1;
    const float* gamma;
    const float* beta;
    const float* w;      // [9][C] tap-major
    float* out_dw;       // [B,H,W,C]
    float* out_xn;       // [B,H,W,C] or null
    int tiles_x, tiles_y;
    int use_gn;
    int b0;              // gn_dw3x3_small_kernel: first sample (blockIdx.y counts from b0)
    int xcd;             // gn_dw3x3_small_kernel: grid = (samples, chunks) with the samples XCD-contiguous (ddif_dev.h wg_work_range) instead of (chunks, samples)
    int flip;            // dw3x3(_q4)_kernel: taps mirrored (w[8 - k]): the input gradient of the same depthwise conv
};

__global__ __launch_bounds__(256) void dw3x3_kernel(DwArgs a) {
    dd_touch_kernargs<sizeof(DwArgs)>();  // (ddif_dev.h: the argument block in one round trip)
    constexpr int TH = 8, TW = 16, IH = TH + 2, IW = TW + 2, CK = 32;
    DDIF_DYN_SMEM(smem);
    float* As = reinterpret_cast<float*>(smem);  // [IH*IW][CK]
    const int tid = threadIdx.x;
    const int tiles = a.tiles_x * a.tiles_y;
    const int b = blockIdx.x / tiles, t = blockIdx.x % tiles;
    const int oy0 = (t / a.tiles_x) * TH, ox0 = (t % a.tiles_x) * TW;
    const int C = a.c0 + a.c1;
    float mean = 0.f, rstd = 1.f;
    if (a.use_gn) {
        if (tid < 64) {
            gn_finalize_wave(a.st0, a.np0, a.st1, a.np1, b, (double)C * a.H * a.W, &mean, &rstd);
            if (tid == 0) {
                As[0] = mean;
                As[1] = rstd;
            }
        }
        __syncthreads();
        mean = As[0];
        rstd = As[1];
        __syncthreads();
    }
    const int cl = tid & 31;
    for (int cb = 0; cb < C; cb += CK) {
        const int c = cb + cl;
        const bool cok = c < C;
        float ga = 1.f, gb = 0.f;
        if (a.use_gn && cok) {
            ga = a.gamma[c] * rstd;
            gb = a.beta[c] - mean * ga;
        }
        if (cb) __syncthreads();
        // (the 180 halo pixels of this thread's channel, eight loads in flight at a time: a load-then-store loop pays one memory latency per pixel)
        for (int p0 = tid >> 5; p0 < IH * IW; p0 += 64) {
            float v[8];
            bool ok[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int pix = p0 + 8 * u;
                const int iy = oy0 - 1 + pix / IW, ix = ox0 - 1 + pix % IW;
                ok[u] = cok && pix < IH * IW && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                const size_t sp = ok[u] ? ((size_t)b * a.H + iy) * a.W + ix : (size_t)b * a.H * a.W;
                const int cc = ok[u] ? c : 0;
                v[u] = (cc < a.c0) ? a.in0[sp * a.c0 + cc] : a.in1[sp * a.c1 + (cc - a.c0)];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int pix = p0 + 8 * u;
                if (pix < IH * IW) As[pix * CK + cl] = ok[u] ? (a.use_gn ? fmaf(v[u], ga, gb) : v[u]) : 0.f;
            }
        }
        __syncthreads();
        float wv[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) wv[k] = cok ? a.w[(a.flip ? 8 - k : k) * C + c] : 0.f;
        for (int p = tid >> 5; p < TH * TW; p += 8) {
            const int ty = p / TW, tx = p % TW;
            const int oy = oy0 + ty, ox = ox0 + tx;
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) s = fmaf(As[((ty + k / 3) * IW + tx + k % 3) * CK + cl], wv[k], s);
            if (cok && oy < a.H && ox < a.W) {
                const size_t op = (((size_t)b * a.H + oy) * a.W + ox) * C + c;
                a.out_dw[op] = s;
                if (a.out_xn) a.out_xn[op] = As[((ty + 1) * IW + tx + 1) * CK + cl];
            }
        }
    }
}

// The same op on QUADS of channels (4 | c0, 4 | c1): one 16-byte access and one set of index arithmetic per four elements -- the scalar form above
// spends ~3600 vector instructions per wavefront on addressing (rocprofv3 SQ_INSTS_VALU) and is instruction-bound, not memory-bound.  Same tile
// (8x16 pixels, 32-channel chunks = 8 quads), same arithmetic per element (bit-identical results).
__global__ __launch_bounds__(256) void dw3x3_q4_kernel(DwArgs a) {
    dd_touch_kernargs<sizeof(DwArgs)>();  // (ddif_dev.h: the argument block in one round trip)
    constexpr int TH = 8, TW = 16, IH = TH + 2, IW = TW + 2, CK = 32, NIT = (IH * IW * 8 + 255) / 256;
    DDIF_DYN_SMEM(smem);
    float* As = reinterpret_cast<float*>(smem);  // [IH*IW][CK]
    const int tid = threadIdx.x;
    const int tiles = a.tiles_x * a.tiles_y;
    const int b = blockIdx.x / tiles, t = blockIdx.x % tiles;
    const int oy0 = (t / a.tiles_x) * TH, ox0 = (t % a.tiles_x) * TW;
    const int C = a.c0 + a.c1;
    float mean = 0.f, rstd = 1.f;
    if (a.use_gn) {
        if (tid < 64) {
            gn_finalize_wave(a.st0, a.np0, a.st1, a.np1, b, (double)C * a.H * a.W, &mean, &rstd);
            if (tid == 0) {
                As[0] = mean;
                As[1] = rstd;
            }
        }
        __syncthreads();
        mean = As[0];
        rstd = As[1];
        __syncthreads();
    }
    const int q = tid & 7;  // this thread's quad of the chunk, in both phases (256 % 8 == 0)
    const size_t img = (size_t)b * a.H * a.W;
    for (int cb = 0; cb < C; cb += CK) {
        const int c = cb + q * 4;
        const bool cok = c < C;
        float4 ga = make_float4(1.f, 1.f, 1.f, 1.f), gb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.use_gn && cok) {
            const float4 g = *reinterpret_cast<const float4*>(a.gamma + c), bt = *reinterpret_cast<const float4*>(a.beta + c);
            ga = make_float4(g.x * rstd, g.y * rstd, g.z * rstd, g.w * rstd);
            gb = make_float4(bt.x - mean * ga.x, bt.y - mean * ga.y, bt.z - mean * ga.z, bt.w - mean * ga.w);
        }
        const float* src = (c < a.c0) ? a.in0 + c : a.in1 + (c - a.c0);
        const int ld = (c < a.c0) ? a.c0 : a.c1;
        if (cb) __syncthreads();
        float4 v[NIT];
        bool ok[NIT];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {  // all of this thread's halo loads in flight together
            const int pix = (tid >> 3) + 32 * u;
            const int iy = oy0 - 1 + pix / IW, ix = ox0 - 1 + pix % IW;
            ok[u] = cok && pix < IH * IW && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            v[u] = ok[u] ? *reinterpret_cast<const float4*>(src + (img + (size_t)iy * a.W + ix) * ld) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int pix = (tid >> 3) + 32 * u;
            if (pix < IH * IW) {
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok[u]) o = a.use_gn ? make_float4(fmaf(v[u].x, ga.x, gb.x), fmaf(v[u].y, ga.y, gb.y), fmaf(v[u].z, ga.z, gb.z), fmaf(v[u].w, ga.w, gb.w)) : v[u];
                *reinterpret_cast<float4*>(As + pix * CK + q * 4) = o;
            }
        }
        __syncthreads();
        float4 wv[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) wv[k] = cok ? *reinterpret_cast<const float4*>(a.w + (a.flip ? 8 - k : k) * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = (tid >> 3) + 32 * u;  // 0 .. 127
            const int ty = p / TW, tx = p % TW;
            const int oy = oy0 + ty, ox = ox0 + tx;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float4 x = *reinterpret_cast<const float4*>(As + ((ty + k / 3) * IW + tx + k % 3) * CK + q * 4);
                s = make_float4(fmaf(x.x, wv[k].x, s.x), fmaf(x.y, wv[k].y, s.y), fmaf(x.z, wv[k].z, s.z), fmaf(x.w, wv[k].w, s.w));
            }
            if (cok && oy < a.H && ox < a.W) {
                const size_t op = ((img + (size_t)oy * a.W + ox)) * C + c;
                *reinterpret_cast<float4*>(a.out_dw + op) = s;
                if (a.out_xn) *reinterpret_cast<float4*>(a.out_xn + op) = *reinterpret_cast<const float4*>(As + ((ty + 1) * IW + tx + 1) * CK + q * 4);
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------
// The same op for samples of <= 256 pixels (the 8x8 / 16x16 levels): one workgroup per (32-channel chunk, sample) holds
// the WHOLE normalised image of its chunk (+ zero border) in LDS, float4 channel groups, one pass.  Emits dw3x3(xn) and
// xn = GroupNorm(cat[in0, in1]); channel counts are multiples of 4.
__global__ __launch_bounds__(256) void gn_dw3x3_small_kernel(DwArgs a) {
    dd_touch_kernargs<sizeof(DwArgs)>();  // (ddif_dev.h: the argument block in one round trip)
    constexpr int CK = 32, HP = CK + 4;  // floats per staged pixel (16 B pad: conflict-free float4 rows)
    DDIF_DYN_SMEM(smem);
    float* Hs = reinterpret_cast<float*>(smem);  // [(H+2)*(W+2)][HP]
    const int tid = threadIdx.x;
    // xcd: blockIdx.x = sample slot (workgroup s runs on XCD s % 8: the samples of one XCD are made contiguous, as the conv kernels' work partition does), blockIdx.y = chunk
    const unsigned gs = gridDim.x;
    const int bs = (a.xcd && (gs & 7u) == 0u) ? (int)((blockIdx.x & 7u) * (gs >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
    const int b = a.b0 + (a.xcd ? bs : (int)blockIdx.y), cb = (a.xcd ? (int)blockIdx.y : (int)blockIdx.x) * CK;
    const int C = a.c0 + a.c1, IW = a.W + 2, n = a.H * a.W, nh = (a.H + 2) * IW;
    const int c4 = tid & 7, c = cb + c4 * 4;
    const bool cok = c < C;
    const bool s0 = !cok || c < a.c0;
    const float* src = s0 ? a.in0 + (cok ? c : 0) : a.in1 + (c - a.c0);
    const int cs = s0 ? a.c0 : a.c1;
    // loads first: the tile rows of this thread (<= 8 pixels), affine parameters, taps -- then the statistics
    float4 v[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int p = (tid >> 3) + it * 32;
        const int pc = p < n ? p : n - 1;
        v[it] = *reinterpret_cast<const float4*>(src + ((size_t)b * n + pc) * cs);
    }
    const float4 gq = *reinterpret_cast<const float4*>(a.gamma + (cok ? c : 0));
    const float4 bq = *reinterpret_cast<const float4*>(a.beta + (cok ? c : 0));
    float4 wk[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wk[k] = *reinterpret_cast<const float4*>(a.w + (size_t)k * C + (cok ? c : 0));
    float mean, rstd;
    gn_finalize_wave(a.st0, a.np0, a.st1, a.np1, b, (double)C * a.H * a.W, &mean, &rstd);  // every wavefront for itself
    for (int i = tid; i < nh; i += 256) {  // zero border (and interior, overwritten below after the barrier)
        const int y = i / IW, x = i - y * IW;
        if (y == 0 || y == a.H + 1 || x == 0 || x == a.W + 1)
            for (int k = 0; k < CK / 4; ++k) *reinterpret_cast<float4*>(&Hs[i * HP + k * 4]) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float ga[4], gb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ga[i] = (&gq.x)[i] * rstd;
        gb[i] = (&bq.x)[i] - mean * ga[i];
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int p = (tid >> 3) + it * 32;
        if (p < n) {
            const int y = p / a.W, x = p - y * a.W;
            float4 xn;
            xn.x = cok ? fmaf(v[it].x, ga[0], gb[0]) : 0.f;
            xn.y = cok ? fmaf(v[it].y, ga[1], gb[1]) : 0.f;
            xn.z = cok ? fmaf(v[it].z, ga[2], gb[2]) : 0.f;
            xn.w = cok ? fmaf(v[it].w, ga[3], gb[3]) : 0.f;
            *reinterpret_cast<float4*>(&Hs[((y + 1) * IW + x + 1) * HP + c4 * 4]) = xn;
            if (cok && a.out_xn) *reinterpret_cast<float4*>(a.out_xn + ((size_t)b * n + p) * C + c) = xn;
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int p = (tid >> 3) + it * 32;
        if (p < n && cok) {
            const int y = p / a.W, x = p - y * a.W;
            float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float4 hv = *reinterpret_cast<const float4*>(&Hs[((y + k / 3) * IW + x + k % 3) * HP + c4 * 4]);
                s[0] = fmaf(hv.x, wk[k].x, s[0]);
                s[1] = fmaf(hv.y, wk[k].y, s[1]);
                s[2] = fmaf(hv.z, wk[k].z, s[2]);
                s[3] = fmaf(hv.w, wk[k].w, s[3]);
            }
            *reinterpret_cast<float4*>(a.out_dw + ((size_t)b * n + p) * C + c) = make_float4(s[0], s[1], s[2], s[3]);
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------
// softmax statistics along one image axis for channels [coff, coff+C) of an NHWC tensor with row stride ld:
//   axis 0: over H (q.softmax(dim=-2), :545) -> mx/sm indexed [b][w][c];  axis 1: over W (k.softmax(dim=-1), :546)
//   -> [b][h][c].  Two passes (max, then sum of exp(x - max)) like torch.softmax.
__global__ void softmax_stats_kernel(const float* in, int ld, int coff, int C, int B, int H, int W, int axis,
                                     float* mx, float* sm) {
    const int keep = axis == 0 ? W : H, red = axis == 0 ? H : W;
    const size_t n = (size_t)B * keep * C;
    const size_t rstride = (size_t)(axis == 0 ? W : 1) * ld;  // floats between consecutive elements of the reduced axis
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int k = (int)((i / C) % keep);
        const int b = (int)(i / ((size_t)C * keep));
        const float* p0 = in + ((size_t)b * H * W + (axis == 0 ? (size_t)k : (size_t)k * W)) * ld + coff + c;
        float m = -INFINITY, s = 0.f;
        if (red <= 64) {
            // the whole line lives in registers: ONE pass over memory, same two-pass arithmetic (and order) as below
            float v[64];
#pragma unroll
            for (int r = 0; r < 64; ++r) v[r] = p0[(size_t)(r < red ? r : red - 1) * rstride];
#pragma unroll
            for (int r = 0; r < 64; ++r) m = fmaxf(m, v[r]);  // the clamped tail repeats the last element: max unchanged
#pragma unroll
            for (int r = 0; r < 64; ++r)
                if (r < red) s += dd_exp(v[r] - m);
        } else {
            for (int r = 0; r < red; ++r) m = fmaxf(m, p0[(size_t)r * rstride]);
            for (int r = 0; r < red; ++r) s += dd_exp(p0[(size_t)r * rstride] - m);
        }
        mx[i] = m;
        sm[i] = s;
    }
}

// ----------------------------------------------------------------------------------------------------------------
// linear-attention context (cond-only, once per tile): ctx[b,hd,i,e] = sum_n softmax_W(k)[i,n] * v[e,n]  (:563)
// kv: [B,H,W,2*Cq] (k | v), channel = hd*d + i.  One workgroup per (hd, b).
__global__ __launch_bounds__(256) void linattn_ctx_kernel(const float* kv, const float* kmx, const float* ksm, int B, int H,
                                                          int W, int Cq, int d, float* ctx) {
    DDIF_DYN_SMEM(smem);
    constexpr int PB = 32;
    float* ks = reinterpret_cast<float*>(smem);  // [PB][d]
    float* vs = ks + PB * d;                     // [PB][d]
    const int hd = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int n = H * W;
    const int nout = d * d;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};  // outputs tid, tid+256, ... (d <= 32)
    for (int p0 = 0; p0 < n; p0 += PB) {
        __syncthreads();
        for (int it = tid; it < PB * d; it += 256) {
            const int pl = it / d, i = it % d, p = p0 + pl;
            float kvv = 0.f, vv = 0.f;
            if (p < n) {
                const int hh = p / W;
                const size_t row = ((size_t)b * n + p) * 2 * Cq;
                const size_t si = ((size_t)b * H + hh) * Cq + hd * d + i;
                kvv = dd_exp(kv[row + hd * d + i] - kmx[si]) / ksm[si];
                vv = kv[row + Cq + hd * d + i];
            }
            ks[it] = kvv;
            vs[it] = vv;
        }
        __syncthreads();
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int idx = tid + o * 256;
            if (idx < nout) {
                const int i = idx / d, e = idx % d;
                float s = acc[o];
                for (int pl = 0; pl < PB; ++pl) s = fmaf(ks[pl * d + i], vs[pl * d + e], s);
                acc[o] = s;
            }
        }
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const int idx = tid + o * 256;
        if (idx < nout) ctx[((size_t)b * gridDim.x + hd) * nout + idx] = acc[o];
    }
}

// Folds the cond-only linear-attention context into the attn_out weights (once per tile batch):
//   attn_out(ctx^T (q_n * scale)) + attn_res(xn) = [M_b | W_res] . cat[q_n, xn],   M_b[co][hd*d+i] = scale * sum_e
//   W_out[co][hd*d+e] * ctx[b,hd,i,e]       (FastAttnCondInjection, models/sr3_dwt.py:561-573)
// and writes the per-sample 1x1 weights in the conv kernel's packed B-fragment order (ks = 1, chunk ck).
__global__ void pack_mix_weights_kernel(const float* wo, const float* wr, const float* ctx, int B, int co_n, int fea, int d,
                                        float scale, int ck, int n_chunks, int nb_pad, float* out) {
    const int K8 = ck / 8;
    const size_t per = (size_t)nb_pad * n_chunks * K8 * 256;
    const int cin_n = wr ? 2 * fea : fea;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < per * B; idx += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(idx / per);
        size_t r = idx % per;
        const int i = (int)(r % 4); r /= 4;
        const int j = (int)(r % 32); r /= 32;
        const int h = (int)(r % 2); r /= 2;
        const int k8 = (int)(r % K8); r /= K8;
        const int ch = (int)(r % n_chunks); r /= n_chunks;
        const int nbi = (int)r;
        const int ci = ch * ck + k8 * 8 + 4 * h + i, co = nbi * 32 + j;
        float v = 0.f;
        if (co < co_n && ci < cin_n) {
            if (ci < fea) {
                const int hd = ci / d, ii = ci % d;
                const float* cx = ctx + (((size_t)b * (fea / d) + hd) * d + ii) * d;
                float s = 0.f;
                for (int e = 0; e < d; ++e) s = fmaf(wo[(size_t)co * fea + hd * d + e], cx[e], s);
                v = s * scale;
            } else {
                v = wr[(size_t)co * fea + (ci - fea)];
            }
        }
        out[idx] = v;
    }
}

// The same fold, written as the three bf16 planes of the bf16x3 conv path (kernels_conv.h MATH = 1):
//   [n-block][chunk][16-channel slab][plane hi|mid|lo][half h][cout j][8 bf16: cin = chunk*ck + 16*k16 + 8*h + t]
__global__ void pack_mix_weights_x3_kernel(const float* wo, const float* wr, const float* ctx, int B, int co_n, int fea, int d,
                                           float scale, int ck, int n_chunks, int nb_pad, float* out) {
    const int K16 = ck / 16;
    const size_t per = (size_t)nb_pad * n_chunks * K16 * 3 * 256;  // floats per sample
    const size_t nel = (size_t)nb_pad * n_chunks * K16 * 2 * 32 * 8;  // weights per sample
    const int cin_n = wr ? 2 * fea : fea;
    unsigned short* o16 = reinterpret_cast<unsigned short*>(out);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < nel * B; idx += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(idx / nel);
        size_t r = idx % nel;
        const int t = (int)(r % 8); r /= 8;
        const int j = (int)(r % 32); r /= 32;
        const int h = (int)(r % 2); r /= 2;
        const int k16 = (int)(r % K16); r /= K16;
        const int ch = (int)(r % n_chunks); r /= n_chunks;
        const int nbi = (int)r;
        const int ci = ch * ck + 16 * k16 + 8 * h + t, co = nbi * 32 + j;
        float v = 0.f;
        if (co < co_n && ci < cin_n) {
            if (ci < fea) {
                const int hd = ci / d, ii = ci % d;
                const float* cx = ctx + (((size_t)b * (fea / d) + hd) * d + ii) * d;
                float s = 0.f;
                for (int e = 0; e < d; ++e) s = fmaf(wo[(size_t)co * fea + hd * d + e], cx[e], s);
                v = s * scale;
            } else {
                v = wr[(size_t)co * fea + (ci - fea)];
            }
        }
        unsigned q0, q1, q2;
        dd_split3(v, &q0, &q1, &q2);
        const size_t fl = (size_t)b * per + ((((size_t)nbi * n_chunks + ch) * K16 + k16) * 3) * 256 + (size_t)(h * 32 + j) * 4;  // float index, plane 0
        o16[fl * 2 + t] = (unsigned short)q0;
        o16[(fl + 256) * 2 + t] = (unsigned short)q1;
        o16[(fl + 512) * 2 + t] = (unsigned short)q2;
    }
}

// ----------------------------------------------------------------------------------------------------------------
// bottleneck self-attention on the matrix cores (the ONE place the north star asks for MFMA): exact-fp32
// v_mfma_f32_16x16x4_f32.  One wavefront = 64 queries of one (tile, head); D = 16.
//   S^T tile (16 keys x 16 queries) = K_tile (A: key rows)  x  Q_tile^T (B)      -> D layout: col = lane&15 = query,
//                                                                                    row = 4*(lane>>4)+r = key
//   so a lane's four accumulator values of a tile are P[query = lane&15][keys 4g..4g+3] -- exactly the A-operand layout
//   (A[i = lane&15][k = lane>>4]) of the second contraction O = P x V: no transpose, no LDS round trip.
//   Softmax over keys = over the rows of S^T: 16 values per lane (4 key tiles x 4 regs) + xor-shuffles 16, 32.
// Two passes over the key blocks (pass 1: running max / sum per query; pass 2: p = exp(s - m) / l, O += P V) keep the
// output accumulator free of rescaling; QK^T is recomputed (K = 16: 4 MFMAs per tile).  The contraction index d is
// permuted (step kk uses d = 4*(lane>>4) + kk) so Q / K rows are single float4 loads.
__global__ __launch_bounds__(64) void self_attn_mfma_kernel(const float* qkv, int n, int C, float scale, float* out) {
    constexpr int D = 16;
    const int hd = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
    const int j = lane & 15, g = lane >> 4;
    const int q0 = blockIdx.x * 64;
    const size_t rs = (size_t)3 * C;                                  // floats per token row of qkv
    const float* base = qkv + (size_t)b * n * rs + (size_t)hd * 3 * D;  // + token*rs + {0, D, 2D} + d
    float4 qf[4];  // B operand of S^T: Q[query = q0 + 16*qt + j][4g .. 4g+3]
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        int qi = q0 + qt * 16 + j;
        qi = qi < n ? qi : n - 1;
        qf[qt] = *reinterpret_cast<const float4*>(base + (size_t)qi * rs + 4 * g);
    }
    float m[4], l[4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        m[qt] = -INFINITY;
        l[qt] = 0.f;
    }
    f32x4 oacc[4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) oacc[qt][r] = 0.f;

    for (int pass = 0; pass < 2; ++pass) {
        for (int k0 = 0; k0 < n; k0 += 64) {
            float4 kf[4];  // A operand of S^T: K[key = k0 + 16*kt + j][4g .. 4g+3]
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                int ki = k0 + kt * 16 + j;
                ki = ki < n ? ki : n - 1;
                kf[kt] = *reinterpret_cast<const float4*>(base + (size_t)ki * rs + D + 4 * g);
            }
            f32x4 st[4][4];  // [kt][qt]: S^T tiles; value r <-> key k0 + 16*kt + 4*g + r, query q0 + 16*qt + j
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int qt = 0; qt < 4; ++qt) {
                    f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) c = DDIF_MFMA_16x16x4((&kf[kt].x)[kk], (&qf[qt].x)[kk], c);
                    st[kt][qt] = c;
                }
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool kok = k0 + kt * 16 + 4 * g + r < n;
#pragma unroll
                    for (int qt = 0; qt < 4; ++qt) st[kt][qt][r] = kok ? st[kt][qt][r] * scale : -INFINITY;
                }
            if (pass == 0) {
#pragma unroll
                for (int qt = 0; qt < 4; ++qt) {
                    float bm = -INFINITY;
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) bm = fmaxf(bm, st[kt][qt][r]);
                    bm = fmaxf(bm, __shfl_xor(bm, 16));
                    bm = fmaxf(bm, __shfl_xor(bm, 32));
                    const float mn = fmaxf(m[qt], bm);
                    float ps = 0.f;
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) ps += dd_exp(st[kt][qt][r] - mn);
                    ps += __shfl_xor(ps, 16);
                    ps += __shfl_xor(ps, 32);
                    l[qt] = l[qt] * (m[qt] == -INFINITY ? 0.f : dd_exp(m[qt] - mn)) + ps;
                    m[qt] = mn;
                }
            } else {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    float vf[4];  // B operand of P V: V[key = k0 + 16*kt + 4*g + r][channel j]
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int ki = k0 + kt * 16 + 4 * g + r;
                        ki = ki < n ? ki : n - 1;
                        vf[r] = base[(size_t)ki * rs + 2 * D + j];
                    }
#pragma unroll
                    for (int qt = 0; qt < 4; ++qt) {
                        const float inv = 1.f / l[qt];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float p = dd_exp(st[kt][qt][r] - m[qt]) * inv;  // masked keys: exp(-inf) = 0
                            oacc[qt] = DDIF_MFMA_16x16x4(p, vf[r], oacc[qt]);
                        }
                    }
                }
            }
        }
    }
    // O tile layout: col = lane&15 = channel, row = 4*g + r = query inside the tile
#pragma unroll
    for (int qt = 0; qt < 4; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qi = q0 + qt * 16 + 4 * g + r;
            if (qi < n) out[((size_t)b * n + qi) * C + hd * D + j] = oacc[qt][r];
        }
}

// ----------------------------------------------------------------------------------------------------------------
// Train mode (models/sr3_dwt.py:288-300 Block with Dropout; :534,576 DropPath on the decoder FFN).  In a train-mode plan
// the dropout sits between the GroupNorm + SiLU prologue and the conv, so the activation is materialised once:
//   y = silu(GroupNorm(x)) * mask        gn_silu_drop_kernel   (mask holds 0 or 1/(1-p); y is also what wgrad will need)
//   out = f * scale[b] + res             droppath_add_kernel   (scale[b] in {0, 1/(1-p)}; emits the GroupNorm partial of out)
//   mask generation                      train_masks_kernel    (every site in one launch; Philox keyed by (seed, site, NCHW element): split-invariant)
// grid = (chunks, B), 256 threads, float4 over C (C % 4 == 0).
__global__ __launch_bounds__(256) void gn_silu_drop_kernel(const float* x, const double* st, int np, const float* gamma, const float* beta,
                                                           const float* mask, int HW, int C, float* y) {
    const int b = blockIdx.y;
    float mean, rstd;
    gn_finalize_wave(st, np, nullptr, 0, b, (double)C * HW, &mean, &rstd);  // every wavefront for itself
    const size_t n4 = (size_t)HW * C / 4, base = (size_t)b * HW * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int c = (int)((i * 4) % C);
        const float4 v = *reinterpret_cast<const float4*>(x + base + i * 4);
        const float4 m = *reinterpret_cast<const float4*>(mask + base + i * 4);
        const float4 g = *reinterpret_cast<const float4*>(gamma + c);
        const float4 bt = *reinterpret_cast<const float4*>(beta + c);
        float4 o;
        o.x = dd_silu(fmaf(v.x, g.x * rstd, bt.x - mean * (g.x * rstd))) * m.x;
        o.y = dd_silu(fmaf(v.y, g.y * rstd, bt.y - mean * (g.y * rstd))) * m.y;
        o.z = dd_silu(fmaf(v.z, g.z * rstd, bt.z - mean * (g.z * rstd))) * m.z;
        o.w = dd_silu(fmaf(v.w, g.w * rstd, bt.w - mean * (g.w * rstd))) * m.w;
        *reinterpret_cast<float4*>(y + base + i * 4) = o;
    }
}

__global__ __launch_bounds__(256) void droppath_add_kernel(const float* f, const float* scale, const float* res, int HW, int C, float* out, double* st_out) {
    DDIF_DYN_SMEM(smem_);
    double* red = reinterpret_cast<double*>(smem_);  // [2][4]
    const int b = blockIdx.y;
    const float sc = scale[b];
    const size_t n4 = (size_t)HW * C / 4, base = (size_t)b * HW * C;
    double s1 = 0.0, s2 = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = *reinterpret_cast<const float4*>(f + base + i * 4);
        const float4 r = *reinterpret_cast<const float4*>(res + base + i * 4);
        float4 o;
        o.x = v.x * sc + r.x;
        o.y = v.y * sc + r.y;
        o.z = v.z * sc + r.z;
        o.w = v.w * sc + r.w;
        *reinterpret_cast<float4*>(out + base + i * 4) = o;
        s1 += ((double)o.x + o.y) + ((double)o.z + o.w);
        s2 += ((double)o.x * o.x + (double)o.y * o.y) + ((double)o.z * o.z + (double)o.w * o.w);
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = s1;
        red[4 + (threadIdx.x >> 6)] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        st_out[((size_t)b * gridDim.x + blockIdx.x) * 2 + 0] = (red[0] + red[1]) + (red[2] + red[3]);
        st_out[((size_t)b * gridDim.x + blockIdx.x) * 2 + 1] = (red[4] + red[5]) + (red[6] + red[7]);
    }
}

// Masks of ALL dropout / DropPath sites of a train-mode plan in one launch.  Element e of a site = its NCHW position in the (tile0 + b)-th
// tile, like the sampler noise, so a batch split over ranks or calls draws the same masks.  One Philox4x32-10 block serves the four
// elements 4q .. 4q+3 (counter = (q, site), key = seed): with 4 | H*W these are four neighbouring pixels of one channel plane.
struct MaskRec {
    float* mask;              // NHWC [B][HW][C]
    int C, HW;
    unsigned site;
    int path;                 // 1: a DropPath site (keep = keep_path)
    unsigned long long blk0;  // first workgroup of this site
};
constexpr int MASK_QPB = 1024;  // Philox blocks (4 elements each) per workgroup
__global__ __launch_bounds__(256) void train_masks_kernel(const MaskRec* recs, int n_recs, int B, unsigned long long seed, unsigned long long tile0, float keep_drop,
                                                          float keep_path) {
    int lo = 0, hi = n_recs - 1;
    while (lo < hi) {  // the last record whose first workgroup is <= blockIdx.x
        const int mid = (lo + hi + 1) >> 1;
        if (recs[mid].blk0 <= blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const MaskRec r = recs[lo];
    const float keep = r.path ? keep_path : keep_drop, inv = 1.0f / keep;
    const size_t q0 = ((size_t)blockIdx.x - r.blk0) * MASK_QPB;
    if ((r.HW & 3) == 0) {
        const int HW4 = r.HW >> 2;
        const size_t nq = (size_t)B * HW4 * r.C;
        for (int k = threadIdx.x; k < MASK_QPB; k += 256) {
            const size_t i = q0 + k;  // (b, p4, c), c fastest: neighbouring lanes write neighbouring channels
            if (i >= nq) break;
            const int c = (int)(i % r.C);
            const size_t p4 = (i / r.C) % HW4, b = i / ((size_t)r.C * HW4);
            const unsigned long long q = ((tile0 + b) * r.C + c) * HW4 + p4;  // = e >> 2 of the four elements
            unsigned o[4];
            philox4x32_10((unsigned)q, (unsigned)(q >> 32), r.site, 0xD0F0u, (unsigned)seed, (unsigned)(seed >> 32), o);
            float* dst = r.mask + ((size_t)b * r.HW + p4 * 4) * r.C + c;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float u = ((float)(o[j] >> 8) + 0.5f) * (1.0f / 16777216.0f);
                dst[(size_t)j * r.C] = u < keep ? inv : 0.f;
            }
        }
    } else {  // general shape (DropPath: one element per sample): element e takes word e & 3 of block e >> 2
        const size_t n = (size_t)B * r.HW * r.C;
        for (int k = threadIdx.x; k < MASK_QPB; k += 256) {
            const size_t i = q0 + k;
            if (i >= n) break;
            const int c = (int)(i % r.C);
            const size_t p = (i / r.C) % r.HW, b = i / ((size_t)r.C * r.HW);
            const unsigned long long e = ((tile0 + b) * r.C + c) * r.HW + p, q = e >> 2;
            unsigned o[4];
            philox4x32_10((unsigned)q, (unsigned)(q >> 32), r.site, 0xD0F0u, (unsigned)seed, (unsigned)(seed >> 32), o);
            const unsigned w = (e & 3) == 0 ? o[0] : ((e & 3) == 1 ? o[1] : ((e & 3) == 2 ? o[2] : o[3]));
            const float u = ((float)(w >> 8) + 0.5f) * (1.0f / 16777216.0f);
            r.mask[i] = u < keep ? inv : 0.f;
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------
// time embedding (PositionalEncoding + noise_level_mlp + every FeatureWiseAffine, :223-258,59-64):
// row r: pe = [sin(t*f_j), cos(t*f_j)], temb = W3 swish(W1 pe + b1) + b3, out[r][s] = Wall[s] . temb + ball[s].
// One workgroup (128 threads) per row.  inner = 32 only (engine configuration).
__global__ __launch_bounds__(128) void time_embed_kernel(const float* tvals, const float* freqs, const float* w1,
                                                         const float* b1, const float* w3, const float* b3,
                                                         const float* wall, const float* ball, int inner, int nslots,
                                                         float* out, float* aux) {
    DDIF_DYN_SMEM(smem);
    float* pe = reinterpret_cast<float*>(smem);  // [inner]
    float* h1 = pe + inner;                      // [4*inner]
    float* te = h1 + 4 * inner;                  // [inner]
    const int r = blockIdx.x, tid = threadIdx.x;
    const float t = tvals[r];
    const int half = inner / 2;
    if (tid < half) {
        const float e = t * freqs[tid];
        pe[tid] = sinf(e);
        pe[half + tid] = cosf(e);
    }
    __syncthreads();
    for (int o = tid; o < 4 * inner; o += blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < inner; ++k) s = fmaf(w1[o * inner + k], pe[k], s);
        s += b1[o];
        h1[o] = dd_silu(s);
        if (aux) {  // training: the reverse pass of the time MLP needs pe | pre-activation | hidden | temb  ([rows][.] arrays, back to back)
            const size_t rows = gridDim.x;
            aux[rows * inner + (size_t)r * 4 * inner + o] = s;
            aux[rows * inner + rows * 4 * inner + (size_t)r * 4 * inner + o] = h1[o];
        }
    }
    if (aux && tid < inner) aux[(size_t)r * inner + tid] = pe[tid];
    __syncthreads();
    for (int o = tid; o < inner; o += blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < 4 * inner; ++k) s = fmaf(w3[o * 4 * inner + k], h1[k], s);
        te[o] = s + b3[o];
        if (aux) aux[(size_t)gridDim.x * 9 * inner + (size_t)r * inner + o] = te[o];
    }
    __syncthreads();
    for (int o = tid; o < nslots; o += blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < inner; ++k) s = fmaf(wall[(size_t)o * inner + k], te[k], s);
        out[(size_t)r * nslots + o] = s + ball[o];
    }
}

// ----------------------------------------------------------------------------------------------------------------
// bilinear resize of cond channels [cbeg, cbeg+n) (F.interpolate(..., mode="bilinear"), align_corners=False, :661-663)
// NCHW in -> NHWC out.
__global__ void resize_bilinear_kernel(const float* in, int B, int CC, int H, int W, int cbeg, int n, int oh, int ow,
                                       float* out) {
    const size_t total = (size_t)B * oh * ow * n;
    const float sh = (float)H / (float)oh, sw = (float)W / (float)ow;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % n);
        const int x = (int)((i / n) % ow);
        const int y = (int)((i / ((size_t)n * ow)) % oh);
        const int b = (int)(i / ((size_t)n * ow * oh));
        float fy = sh * (y + 0.5f) - 0.5f, fx = sw * (x + 0.5f) - 0.5f;
        if (fy < 0.f) fy = 0.f;
        if (fx < 0.f) fx = 0.f;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float ly = fy - y0, lx = fx - x0;
        const float* p = in + ((size_t)b * CC + cbeg + c) * H * W;
        const float v = (1.f - ly) * ((1.f - lx) * p[(size_t)y0 * W + x0] + lx * p[(size_t)y0 * W + x1]) +
                        ly * ((1.f - lx) * p[(size_t)y1 * W + x0] + lx * p[(size_t)y1 * W + x1]);
        out[i] = v;
    }
}

// ----------------------------------------------------------------------------------------------------------------
// layout conversion at the drop-in boundary (the reference API is NCHW, the kernels are NHWC)
__global__ void nchw_to_nhwc_kernel(const float* in, int B, int C, int HW, int cbeg, int n, float* out) {
    const size_t total = (size_t)B * HW * n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % n);
        const size_t p = (i / n) % HW;
        const size_t b = i / ((size_t)n * HW);
        out[i] = in[(b * C + cbeg + c) * HW + p];
    }
}
__global__ void nhwc_to_nchw_kernel(const float* in, int B, int C, int HW, float* out) {
    const size_t total = (size_t)B * HW * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i % HW;
        const int c = (int)((i / HW) % C);
        const size_t b = i / ((size_t)C * HW);
        out[i] = in[(b * HW + p) * C + c];
    }
}

// x_T = randn (p_sample_loop :484) written in the kernels' NHWC layout; element index is the NCHW index.
__global__ void randn_nhwc_kernel(float* out, int B, int C, int HW, unsigned long long seed, unsigned draw,
                                  unsigned long long tile0) {
    const size_t total = (size_t)B * HW * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t p = (i / C) % HW;
        const size_t b = i / ((size_t)C * HW);
        out[i] = philox_normal(seed, draw, ((tile0 + b) * C + c) * HW + p);
    }
}

// ----------------------------------------------------------------------------------------------------------------
// sampler updates.  All tensors NHWC [B,HW,C] except `noise` (NCHW, as drawn by the reference) which may be null
// (-> Philox).  lms = cond[:, :C].
// Per-run sampler state lives in DEVICE memory (rewritten at the start of every sampling call) and the step index is a
// device counter advanced by step_advance_kernel: the kernels of a denoising step therefore take the same arguments
// at every step, which is what lets the whole step be replayed from a hipGraph.
struct StepArgs {
    const float* x0;   // network output
    const float* img;  // x_t
    const float* lms;
    float* out;        // x_{t-1}
    int B, C, HW;
    const SamplerRun* run;
    const int* step;
};
__global__ void step_advance_kernel(const int* cur, int* next) { *next = *cur + 1; }  // double-buffered counters: nobody reads `next` during this step

// DDPM p_sample (:418-442, 346-415, 316-325): x0c = clamp(x0 + lms) - lms; out = c1*x0c + c2*img + c3*z
// with c1/c2 = posterior_mean_coef1/2[t] (tab 0/1), c3 = [t != 0] * exp(0.5 * posterior_log_variance_clipped[t]) (tab 2).
__global__ void ddpm_step_kernel(StepArgs a) {
#pragma clang fp contract(off)
    const SamplerRun r = *a.run;
    const int k = *a.step;
    const float c1 = r.tab[0][k], c2 = r.tab[1][k], c3 = r.tab[2][k];
    const size_t total = (size_t)a.B * a.HW * a.C;
    const float* noise = r.noise ? r.noise + (size_t)k * total : nullptr;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % a.C);
        const size_t p = (i / a.C) % a.HW;
        const size_t b = i / ((size_t)a.C * a.HW);
        float x0 = a.x0[i];
        if (r.do_clamp) {
            const float l = a.lms[i];
            x0 = fminf(fmaxf(x0 + l, r.lo), r.hi) - l;
        }
        const size_t e = (b * a.C + c) * a.HW + p;
        const float z = noise ? noise[e] : philox_normal(r.seed, (unsigned)(k + 1), (r.tile0 * a.C * a.HW) + e);
        const float mean = c1 * x0 + c2 * a.img[i];
        a.out[i] = mean + c3 * z;
    }
}

// DDIM step (:594-621): eps = (sqrt_recip*img - x0)/sqrt_recipm1 ; out = sqrt(ap)*x0 + dir*eps + sigma*z
// tables: 0 sqrt_recip, 1 sqrt_recipm1, 2 sqrt(alphas_cumprod_prev), 3 sqrt(1 - ap - sigma^2), 4 [j != 0]*sigma.
__global__ void ddim_step_kernel(StepArgs a) {
#pragma clang fp contract(off)
    const SamplerRun r = *a.run;
    const int k = *a.step;
    const float sqrt_recip = r.tab[0][k], sqrt_recipm1 = r.tab[1][k], sqrt_ap = r.tab[2][k], dir_coef = r.tab[3][k], sigma = r.tab[4][k];
    const size_t total = (size_t)a.B * a.HW * a.C;
    const float* noise = r.noise ? r.noise + (size_t)k * total : nullptr;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % a.C);
        const size_t p = (i / a.C) % a.HW;
        const size_t b = i / ((size_t)a.C * a.HW);
        float x0 = a.x0[i];
        if (r.do_clamp) {
            const float l = a.lms[i];
            x0 = fminf(fmaxf(x0 + l, r.lo), r.hi) - l;
        }
        const float img = a.img[i];
        const float eps = (sqrt_recip * img - x0) / sqrt_recipm1;
        float v = x0 * sqrt_ap + dir_coef * eps;
        if (sigma != 0.f) {
            const size_t e = (b * a.C + c) * a.HW + p;
            const float z = noise ? noise[e] : philox_normal(r.seed, (unsigned)(k + 1), (r.tile0 * a.C * a.HW) + e);
            v += sigma * z;
        }
        a.out[i] = v;
    }
}

// DPM-Solver++ data prediction (solver/dpm_solver.py:298-300,441-450): the x_start -> eps -> x_start round trip of
// model_wrapper + data_prediction_fn, then the image-space clamp corrector (diffusion_engine.py:43-49).
__global__ void dpm_x0_kernel(const float* net, const float* x, const float* lms, float alpha, float sigma, float lo,
                              float hi, int do_clamp, size_t total, float* x0_out) {
#pragma clang fp contract(off)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float xv = x[i];
        const float eps = (xv - alpha * net[i]) / sigma;
        float x0 = (xv - sigma * eps) / alpha;
        if (do_clamp) {
            const float l = lms[i];
            x0 = fminf(fmaxf(x0 + l, lo), hi) - l;
        }
        x0_out[i] = x0;
    }
}
// multistep update (solver/dpm_solver.py:577-584, 828-839, 884-901) in the reference's operation order:
//   order 1: cx*x - a1*m0
//   order 2: D1 = inv_r0*(m0-m1);                          cx*x - a1*m0 - (0.5*a1)*D1
//   order 3: D10 = inv_r0*(m0-m1); D11 = inv_r1*(m1-m2); D1 = D10 + r0_frac*(D10-D11); D2 = inv_r01*(D10-D11);
//            cx*x - a1*m0 + a2*D1 - a3*D2
struct DpmUpdArgs {
    const float* x;
    const float* m0;
    const float* m1;
    const float* m2;
    float* out;
    size_t n;
    int order;
    float cx, a1, inv_r0, inv_r1, r0_frac, inv_r01, a2, a3;
};
__global__ void dpm_update_kernel(DpmUpdArgs a) {
#pragma clang fp contract(off)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x) {
        const float x = a.x[i], m0 = a.m0[i];
        float v = a.cx * x - a.a1 * m0;
        if (a.order == 2) {
            const float d1 = a.inv_r0 * (m0 - a.m1[i]);
            v = v - (0.5f * a.a1) * d1;
        } else if (a.order == 3) {
            const float m1 = a.m1[i], m2 = a.m2[i];
            const float d10 = a.inv_r0 * (m0 - m1), d11 = a.inv_r1 * (m1 - m2);
            const float d1 = d10 + a.r0_frac * (d10 - d11);
            const float d2 = a.inv_r01 * (d10 - d11);
            v = (v + a.a2 * d1) - a.a3 * d2;
        }
        a.out[i] = v;
    }
}

// q_sample (:668-681): x_t = a[b]*x0 + s[b]*noise, all NHWC
__global__ void q_sample_kernel(const float* x0, const float* noise, const float* a, const float* s, int B, size_t per,
                                float* out) {
#pragma clang fp contract(off)
    const size_t total = (size_t)B * per;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / per;
        out[i] = a[b] * x0[i] + s[b] * noise[i];
    }
}

}  // namespace ddif
