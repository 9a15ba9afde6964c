// Linear attention of FastAttnCondInjection (models/sr3_dwt.py:536-577) for the TRAINING step, NHWC:
//   q_sm = softmax over H of q_pre (per column x and channel), k_sm = softmax over W of k_pre (per row y and channel),
//   ctx[hd][a][e] = sum_n k_sm[a][n] v[e][n]            (per sample and head; d x d, heads * d = C channels)
//   o[e][n]       = sc * sum_a ctx[hd][a][e] q_sm[a][n]
// Every pass is ONE read of its operands by workgroups that each own whole softmax lines:
//   k side: a workgroup holds a few ROWS  (all W pixels x C channels)  -> the row softmax is local
//   q side: a workgroup holds a few COLUMNS (all H pixels x C channels) -> the column softmax is local
// and the only cross-workgroup quantity is the tiny context (C * d floats per sample), summed from per-workgroup partials in a fixed order
// by la_reduce_kernel.  Forward: k side -> reduce -> q side.  Backward: q side (dq, dctx partials) -> reduce -> k side (dk, dv).
// (Round 3's first form ran one 1024-thread workgroup per (sample, head): 256 workgroups walking the image in bands, 80 / 213 us per launch
// against 10-25 us of memory time.)  Deterministic: one owner thread per LDS cell and output, fixed summation orders, no atomics.
#pragma once
#include "ddif_dev.h"

namespace ddif {

constexpr int LA_THREADS = 256;
constexpr int LA_TILE_MAX = 8192;  // floats per staged tile (one line of len * C floats must fit)
__host__ __device__ inline int la_lines(int len, int C, int nlines) {
    int n = 6144 / (len * C);
    if (n < 1) n = 1;
    return n > nlines ? nlines : n;
}
__host__ __device__ inline int la_groups(int len, int C, int nlines) {
    const int nl = la_lines(len, C, nlines);
    return (nlines + nl - 1) / nl;
}

// Index arithmetic: every kernel below turns flat thread indices into (line, position, channel, head) with runtime divisors (96 channels, head
// dim 12 ...).  An integer division costs ~40 instructions on this hardware -- more than the 2 * d multiply-adds an element needs -- so the divisors
// become multiplications by a 32-bit reciprocal once per kernel: exact for n * d < 2^32 (here n < 2^20, d <= 1024).
struct FastDiv {
    unsigned d, m;
};
__device__ __forceinline__ FastDiv fd_make(int d) {
    FastDiv f;
    f.d = (unsigned)d;
    f.m = d > 1 ? 0xFFFFFFFFu / (unsigned)d + 1u : 0u;
    return f;
}
__device__ __forceinline__ int fd_div(int n, const FastDiv& f) { return f.d > 1 ? (int)(((unsigned long long)(unsigned)n * f.m) >> 32) : n; }

// Geometry of a line group: line l (0 .. nl-1) and position i (0 .. len-1) -> pixel line * ls + i * is of the H x W image
struct LaGeom {
    int len, ls, is, line0, nl;  // nl: valid lines of THIS group
    FastDiv fC4, flen, fd4;
};
__device__ __forceinline__ LaGeom la_geom(bool rows, int H, int W, int C, int d, int g) {
    LaGeom G;
    const int nlines = rows ? H : W;
    G.len = rows ? W : H;
    G.ls = rows ? W : 1;
    G.is = rows ? 1 : W;
    const int per = la_lines(G.len, C, nlines);
    G.line0 = g * per;
    G.nl = nlines - G.line0 < per ? nlines - G.line0 : per;
    G.fC4 = fd_make(C >> 2);
    G.flen = fd_make(G.len);
    G.fd4 = fd_make(d >> 2);
    return G;
}
// Everything below works on QUADS of channels (4 | d, so a quad never straddles a head): quad u of a tile -> channel quad c4, pixel-in-tile
// pi = l * len + i.  One LDS / global access and one set of index arithmetic serve four elements.
__device__ __forceinline__ void la_split(int u, int C4, const LaGeom& G, int* c4, int* pi) {
    *pi = fd_div(u, G.fC4);
    *c4 = u - *pi * C4;
}
__device__ __forceinline__ int la_pixel(int pi, const LaGeom& G) {
    const int l = fd_div(pi, G.flen), i = pi - l * G.len;
    return (G.line0 + l) * G.ls + i * G.is;
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
// two tiles of the same geometry at once (k and v, q and do): tile[(l * len + i) * C + c] = src[pixel(l, i) * ld + coff + c]; all eight loads of a
// thread are in flight before the first LDS write.  src1 == nullptr: one tile.
__device__ __forceinline__ void la_load_tiles(const float* src0, int ld0, int coff0, float* tile0, const float* src1, int ld1, int coff1, float* tile1, const LaGeom& G, int C) {
    const int C4 = C >> 2, n4 = G.nl * G.len * C4;
    for (int u0 = threadIdx.x; u0 < n4; u0 += LA_THREADS * 4) {
        float4 v0[4], v1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int u = u0 + k * LA_THREADS < n4 ? u0 + k * LA_THREADS : u0;
            int c4, pi;
            la_split(u, C4, G, &c4, &pi);
            const size_t px = (size_t)la_pixel(pi, G);
            v0[k] = ld4(src0 + px * ld0 + coff0 + c4 * 4);
            if (src1) v1[k] = ld4(src1 + px * ld1 + coff1 + c4 * 4);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int u = u0 + k * LA_THREADS;
            if (u < n4) {
                st4(tile0 + u * 4, v0[k]);
                if (src1) st4(tile1 + u * 4, v1[k]);
            }
        }
    }
}
// How many threads share one (line, channel quad): the largest power of two P <= 16 with P * (lines * C / 4) <= 256.  A thread then owns the
// positions part, part + P, ... of its line: at most LA_POS of them (tile <= 8192 floats)
constexpr int LA_POS = 16;
__device__ __forceinline__ int la_parts(int nlq) {
    int p = 1;
    while (p < 16 && 2 * p * nlq <= LA_THREADS) p *= 2;
    return p;
}
__device__ __forceinline__ float4 max4(const float4& a, const float4& b) { return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)); }
__device__ __forceinline__ float4 add4(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 fma4(float s, const float4& m, const float4& acc) {
    return make_float4(fmaf(s, m.x, acc.x), fmaf(s, m.y, acc.y), fmaf(s, m.z, acc.z), fmaf(s, m.w, acc.w));
}
// softmax along i of every (line, channel) of the tile, in place: exp(x - max) / sum (the two-pass arithmetic of torch.softmax).  The P threads of
// a line keep their positions in registers (one LDS read, one LDS write per element) and exchange max / sum through `sx` [2][256] float4, combined
// in part order.  Called by the whole workgroup (barriers inside); requires ceil(len / P) <= LA_POS (true for every tile that fits).
__device__ __forceinline__ void la_softmax_tile(float* tile, const LaGeom& G, int C, float4* sx) {
    const int C4 = C >> 2, nlq = G.nl * C4, P = la_parts(nlq), per = LA_THREADS / P;
    const int t = threadIdx.x, part = fd_div(t, fd_make(per)), lq = t - part * per;
    const bool on = lq < nlq;
    const int l = on ? fd_div(lq, G.fC4) : 0, c4 = on ? lq - l * C4 : 0;
    float* p = tile + (l * G.len) * C + c4 * 4;
    float4 v[LA_POS];
    float4 mx = make_float4(-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f);
#pragma unroll
    for (int k = 0; k < LA_POS; ++k) {
        const int i = part + k * P;
        if (on && i < G.len) {
            v[k] = ld4(p + i * C);
            mx = max4(mx, v[k]);
        }
    }
    sx[t] = mx;
    __syncthreads();
    if (on)
        for (int q = 0; q < P; ++q) mx = max4(mx, sx[q * per + lq]);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < LA_POS; ++k) {
        const int i = part + k * P;
        if (on && i < G.len) {
            v[k] = make_float4(dd_exp(v[k].x - mx.x), dd_exp(v[k].y - mx.y), dd_exp(v[k].z - mx.z), dd_exp(v[k].w - mx.w));
            s = add4(s, v[k]);
        }
    }
    sx[LA_THREADS + t] = s;
    __syncthreads();
    if (on) {
        float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int q = 0; q < P; ++q) tot = add4(tot, sx[LA_THREADS + q * per + lq]);
#pragma unroll
        for (int k = 0; k < LA_POS; ++k) {
            const int i = part + k * P;
            if (i < G.len) st4(p + i * C, make_float4(v[k].x / tot.x, v[k].y / tot.y, v[k].z / tot.z, v[k].w / tot.w));
        }
    }
    __syncthreads();
}
// sums[l * C + c] = sum over i of ta[l][i][c] * tb[l][i][c], the same way (P threads per line, combined in part order)
__device__ __forceinline__ void la_line_dots(const float* ta, const float* tb, const LaGeom& G, int C, float4* sx, float* sums) {
    const int C4 = C >> 2, nlq = G.nl * C4, P = la_parts(nlq), per = LA_THREADS / P;
    const int t = threadIdx.x, part = fd_div(t, fd_make(per)), lq = t - part * per;
    const bool on = lq < nlq;
    const int l = on ? fd_div(lq, G.fC4) : 0, c4 = on ? lq - l * C4 : 0;
    const int o = (l * G.len) * C + c4 * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < LA_POS; ++k) {
        const int i = part + k * P;
        if (on && i < G.len) {
            const float4 a = ld4(ta + o + i * C), b = ld4(tb + o + i * C);
            s = make_float4(fmaf(a.x, b.x, s.x), fmaf(a.y, b.y, s.y), fmaf(a.z, b.z, s.z), fmaf(a.w, b.w, s.w));
        }
    }
    sx[t] = s;
    __syncthreads();
    if (on && part == 0) {
        float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int q = 0; q < P; ++q) tot = add4(tot, sx[q * per + lq]);
        st4(sums + lq * 4, tot);  // = sums[l * C + c4 * 4 ..]
    }
    __syncthreads();
}
// part[(hd * d + a) * d + e] = scale * sum over the tile's pixels of A[pix][hd * d + a] * Bt[pix][hd * d + e]: a thread owns (a, quad of e) and one of
// PP interleaved pixel slices; the slices are added in slice order through `sx`.  Called by the whole workgroup.
__device__ __forceinline__ void la_contract_tile(const float* A, const float* Bt, int npix, int C, int d, const LaGeom& G, float scale, float4* sx, float* part) {
    const int d4 = d >> 2, units = C * d4;  // (ca, e4)
    const int PP = units <= 64 ? 4 : (units <= 128 ? 2 : 1);
    const FastDiv fdd = fd_make(d);
    const int t = threadIdx.x, per = LA_THREADS / PP, sl = fd_div(t, fd_make(per));
    for (int o0 = 0; o0 < units; o0 += per) {
        const int o = o0 + t - sl * per;
        const bool on = o < units;
        const int ca = on ? fd_div(o, G.fd4) : 0, e4 = on ? o - ca * d4 : 0, hd = fd_div(ca, fdd);
        const float* pa = A + ca;
        const float* pb = Bt + hd * d + e4 * 4;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (on)
            for (int n0 = sl; n0 < npix; n0 += 4 * PP) {  // four pixels' operands in registers per step
                float a[4];
                float4 b[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int n = n0 + k * PP < npix ? n0 + k * PP : n0;
                    a[k] = pa[n * C];
                    b[k] = ld4(pb + n * C);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (n0 + k * PP < npix) s = fma4(a[k], b[k], s);
            }
        if (PP > 1) {
            sx[t] = s;
            __syncthreads();
            if (on && sl == 0)
                for (int q = 1; q < PP; ++q) s = add4(s, sx[q * per + t]);
        }
        if (on && sl == 0) st4(part + ca * d + e4 * 4, make_float4(s.x * scale, s.y * scale, s.z * scale, s.w * scale));
        if (PP > 1) __syncthreads();
    }
}
// sum over t < d of M[t * d + (0..3)] * x[t]: four outputs of a head's d x d matrix (row-major, the quad of columns at M) times a pixel's d values
__device__ __forceinline__ float4 la_head_mv(const float* M, const float* x, int d) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < d; t += 4) {  // (4 | d)
        const float4 x4 = ld4(x + t);
        const float4 m0 = ld4(M + t * d), m1 = ld4(M + (t + 1) * d), m2 = ld4(M + (t + 2) * d), m3 = ld4(M + (t + 3) * d);
        s = fma4(x4.x, m0, s);
        s = fma4(x4.y, m1, s);
        s = fma4(x4.z, m2, s);
        s = fma4(x4.w, m3, s);
    }
    return s;
}
constexpr int LA_QPT = LA_TILE_MAX / 4 / LA_THREADS;  // channel quads per thread (8)

// ---- k side, forward: partial context of a group of rows.  grid = (row groups, B)
__global__ __launch_bounds__(LA_THREADS) void la_kside_fwd_kernel(const float* kv_pre, int H, int W, int C, int d, float* pctx /* [B][groups][C * d] */) {
    DDIF_DYN_SMEM(smem_);
    float4* sx = reinterpret_cast<float4*>(smem_);               // [2][256] exchange of the line reductions
    float* tk = reinterpret_cast<float*>(sx + 2 * LA_THREADS);   // k rows -> k_sm
    const LaGeom G = la_geom(true, H, W, C, d, blockIdx.x);
    float* tv = tk + la_lines(G.len, C, H) * G.len * C;
    const int b = blockIdx.y;
    const float* src = kv_pre + (size_t)b * H * W * 2 * C;
    la_load_tiles(src, 2 * C, 0, tk, src, 2 * C, C, tv, G, C);
    __syncthreads();
    la_softmax_tile(tk, G, C, sx);
    la_contract_tile(tk, tv, G.nl * G.len, C, d, G, 1.0f, sx, pctx + ((size_t)b * gridDim.x + blockIdx.x) * C * d);
}

// out[b][o] = sum over the groups of part[b][g][o], in group order (eight loads in flight)
__global__ void la_reduce_kernel(const float* part, int B, int groups, int n, float* out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)B * n; i += (size_t)gridDim.x * blockDim.x) {  // (the launch grid is capped)
        const size_t b = i / n, o = i % n;
        const float* p = part + b * groups * n + o;
        float acc = 0.f;
        for (int g0 = 0; g0 < groups; g0 += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = g0 + u < groups ? p[(size_t)(g0 + u) * n] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        out[i] = acc;
    }
}

// ---- q side, forward: o = sc * ctx^T q_sm for a group of columns.  grid = (column groups, B)
__global__ __launch_bounds__(LA_THREADS) void la_qside_fwd_kernel(const float* q_pre, const float* ctx /* [B][C * d] */, int H, int W, int C, int d, float sc, float* out,
                                                                   int ld_o) {
    DDIF_DYN_SMEM(smem_);
    float4* sx = reinterpret_cast<float4*>(smem_);
    float* tq = reinterpret_cast<float*>(sx + 2 * LA_THREADS);
    const LaGeom G = la_geom(false, H, W, C, d, blockIdx.x);
    float* cx = tq + la_lines(G.len, C, W) * G.len * C;  // [C][d]: cx[(hd * d + a) * d + e]
    const int b = blockIdx.y, C4 = C >> 2, d4 = d >> 2;
    la_load_tiles(q_pre + (size_t)b * H * W * C, C, 0, tq, nullptr, 0, 0, nullptr, G, C);
    for (int i = threadIdx.x; i < C * d4; i += LA_THREADS) st4(cx + i * 4, ld4(ctx + (size_t)b * C * d + i * 4));
    __syncthreads();
    la_softmax_tile(tq, G, C, sx);
    const int nq = G.nl * G.len * C4;
    float* ob = out + (size_t)b * H * W * ld_o;
    for (int u = threadIdx.x; u < nq; u += LA_THREADS) {
        int c4, pi;
        la_split(u, C4, G, &c4, &pi);
        const int hd = fd_div(c4, G.fd4), e4 = c4 - hd * d4;
        const float4 s = la_head_mv(cx + hd * d * d + e4 * 4, tq + pi * C + hd * d, d);  // sum_a ctx[hd][a][e..e+3] q_sm[a]
        st4(ob + (size_t)la_pixel(pi, G) * ld_o + c4 * 4, make_float4(s.x * sc, s.y * sc, s.z * sc, s.w * sc));
    }
}

// ---- q side, backward (do given): dq_pre (final) and the partial dctx of a group of columns.  grid = (column groups, B)
//   dqs[a][n] = sc * sum_e ctx[a][e] do[e][n];   dq_pre = q_sm * (dqs - sum over the column of q_sm * dqs);   dctx[a][e] = sc * sum_n q_sm[a][n] do[e][n]
__global__ __launch_bounds__(LA_THREADS) void la_qside_bwd_kernel(const float* q_pre, const float* dout, int ld_g, const float* ctx, int H, int W, int C, int d, float sc,
                                                                   float* dq_pre, float* pdctx /* [B][groups][C * d] */) {
    DDIF_DYN_SMEM(smem_);
    float4* sx = reinterpret_cast<float4*>(smem_);
    float* tq = reinterpret_cast<float*>(sx + 2 * LA_THREADS);
    const LaGeom G = la_geom(false, H, W, C, d, blockIdx.x);
    const int tile = la_lines(G.len, C, W) * G.len * C;
    float* tg = tq + tile;    // do, later dqs
    float* cxT = tg + tile;   // [C][d] transposed inside a head: cxT[(hd * d + e) * d + a] = ctx[(hd * d + a) * d + e]
    float* cs = cxT + C * d;  // [lines][C] column sums of q_sm * dqs
    const int b = blockIdx.y, C4 = C >> 2, d4 = d >> 2;
    la_load_tiles(q_pre + (size_t)b * H * W * C, C, 0, tq, dout + (size_t)b * H * W * ld_g, ld_g, 0, tg, G, C);
    for (int i4 = threadIdx.x; i4 < C * d4; i4 += LA_THREADS) {  // i4 = (hd * d + a) * d4 + e4 of the source: coalesced read, transposed LDS write
        const float4 v = ld4(ctx + (size_t)b * C * d + i4 * 4);
        const int ca = fd_div(i4, G.fd4), e4 = i4 - ca * d4, hd = fd_div(ca, fd_make(d)), a = ca - hd * d;
        float* dst = cxT + (hd * d + e4 * 4) * d + a;
        dst[0] = v.x;
        dst[d] = v.y;
        dst[2 * d] = v.z;
        dst[3 * d] = v.w;
    }
    __syncthreads();
    la_softmax_tile(tq, G, C, sx);
    la_contract_tile(tq, tg, G.nl * G.len, C, d, G, sc, sx, pdctx + ((size_t)b * gridDim.x + blockIdx.x) * C * d);
    const int nq = G.nl * G.len * C4;
    float4 r[LA_QPT];
#pragma unroll
    for (int k = 0; k < LA_QPT; ++k) {
        const int u = threadIdx.x + k * LA_THREADS;
        if (u < nq) {
            int c4, pi;
            la_split(u, C4, G, &c4, &pi);
            const int hd = fd_div(c4, G.fd4), a4 = c4 - hd * d4;
            const float4 s = la_head_mv(cxT + hd * d * d + a4 * 4, tg + pi * C + hd * d, d);  // sum_e ctx[hd][a..a+3][e] do[e]
            r[k] = make_float4(s.x * sc, s.y * sc, s.z * sc, s.w * sc);
        }
    }
    __syncthreads();  // every read of do is done
#pragma unroll
    for (int k = 0; k < LA_QPT; ++k) {
        const int u = threadIdx.x + k * LA_THREADS;
        if (u < nq) st4(tg + u * 4, r[k]);
    }
    __syncthreads();
    la_line_dots(tg, tq, G, C, sx, cs);
    float* db = dq_pre + (size_t)b * H * W * C;
#pragma unroll
    for (int k = 0; k < LA_QPT; ++k) {
        const int u = threadIdx.x + k * LA_THREADS;
        if (u < nq) {
            int c4, pi;
            la_split(u, C4, G, &c4, &pi);
            const int l = fd_div(pi, G.flen);
            const float4 q = ld4(tq + u * 4), c = ld4(cs + l * C + c4 * 4);
            st4(db + (size_t)la_pixel(pi, G) * C + c4 * 4, make_float4(q.x * (r[k].x - c.x), q.y * (r[k].y - c.y), q.z * (r[k].z - c.z), q.w * (r[k].w - c.w)));
        }
    }
}

// ---- k side, backward: dk_pre, dv (final) of a group of rows.  grid = (row groups, B)
//   dks[a][n] = sum_e dctx[a][e] v[e][n];   dk_pre = k_sm * (dks - sum over the row of k_sm * dks);   dv[e][n] = sum_a k_sm[a][n] dctx[a][e]
__global__ __launch_bounds__(LA_THREADS) void la_kside_bwd_kernel(const float* kv_pre, const float* dctx /* [B][C * d] */, int H, int W, int C, int d, float* dkv_pre) {
    DDIF_DYN_SMEM(smem_);
    float4* sx = reinterpret_cast<float4*>(smem_);
    float* tk = reinterpret_cast<float*>(sx + 2 * LA_THREADS);
    const LaGeom G = la_geom(true, H, W, C, d, blockIdx.x);
    const int tile = la_lines(G.len, C, H) * G.len * C;
    float* tv = tk + tile;     // v, later dks
    float* dc = tv + tile;     // [C][d]  dc[(hd * d + a) * d + e]
    float* dcT = dc + C * d;   // [C][d]  dcT[(hd * d + e) * d + a]
    float* rs = dcT + C * d;   // [lines][C] row sums of k_sm * dks
    const int b = blockIdx.y, C4 = C >> 2, d4 = d >> 2;
    const float* src = kv_pre + (size_t)b * H * W * 2 * C;
    la_load_tiles(src, 2 * C, 0, tk, src, 2 * C, C, tv, G, C);
    for (int i4 = threadIdx.x; i4 < C * d4; i4 += LA_THREADS) {
        const float4 v = ld4(dctx + (size_t)b * C * d + i4 * 4);
        const int ca = fd_div(i4, G.fd4), e4 = i4 - ca * d4, hd = fd_div(ca, fd_make(d)), a = ca - hd * d;
        st4(dc + i4 * 4, v);
        float* dst = dcT + (hd * d + e4 * 4) * d + a;
        dst[0] = v.x;
        dst[d] = v.y;
        dst[2 * d] = v.z;
        dst[3 * d] = v.w;
    }
    __syncthreads();
    la_softmax_tile(tk, G, C, sx);
    const int nq = G.nl * G.len * C4;
    float* ob = dkv_pre + (size_t)b * H * W * 2 * C;
    float4 r[LA_QPT];
#pragma unroll
    for (int k = 0; k < LA_QPT; ++k) {
        const int u = threadIdx.x + k * LA_THREADS;
        if (u < nq) {
            int c4, pi;
            la_split(u, C4, G, &c4, &pi);
            const int hd = fd_div(c4, G.fd4), j4 = c4 - hd * d4;
            r[k] = la_head_mv(dcT + hd * d * d + j4 * 4, tv + pi * C + hd * d, d);                 // dks[a..a+3] = sum_e dctx[a][e] v[e]
            const float4 dv = la_head_mv(dc + hd * d * d + j4 * 4, tk + pi * C + hd * d, d);      // dv[e..e+3] = sum_a dctx[a][e] k_sm[a]
            st4(ob + (size_t)la_pixel(pi, G) * 2 * C + C + c4 * 4, dv);
        }
    }
    __syncthreads();  // every read of v is done
#pragma unroll
    for (int k = 0; k < LA_QPT; ++k) {
        const int u = threadIdx.x + k * LA_THREADS;
        if (u < nq) st4(tv + u * 4, r[k]);
    }
    __syncthreads();
    la_line_dots(tv, tk, G, C, sx, rs);
#pragma unroll
    for (int k = 0; k < LA_QPT; ++k) {
        const int u = threadIdx.x + k * LA_THREADS;
        if (u < nq) {
            int c4, pi;
            la_split(u, C4, G, &c4, &pi);
            const int l = fd_div(pi, G.flen);
            const float4 kk = ld4(tk + u * 4), c = ld4(rs + l * C + c4 * 4);
            st4(ob + (size_t)la_pixel(pi, G) * 2 * C + c4 * 4, make_float4(kk.x * (r[k].x - c.x), kk.y * (r[k].y - c.y), kk.z * (r[k].z - c.z), kk.w * (r[k].w - c.w)));
        }
    }
}

// dynamic LDS of the four kernels (floats -> bytes): exchange area [2][256] float4, tiles, context copies, line sums
inline size_t la_tile_floats(bool rows, int H, int W, int C) {
    const int len = rows ? W : H, nlines = rows ? H : W;
    return (size_t)la_lines(len, C, nlines) * len * C;
}
inline size_t la_kside_fwd_smem(int H, int W, int C) { return (8 * LA_THREADS + 2 * la_tile_floats(true, H, W, C)) * sizeof(float); }
inline size_t la_qside_fwd_smem(int H, int W, int C, int d) { return (8 * LA_THREADS + la_tile_floats(false, H, W, C) + (size_t)C * d) * sizeof(float); }
inline size_t la_qside_bwd_smem(int H, int W, int C, int d) {
    return (8 * LA_THREADS + 2 * la_tile_floats(false, H, W, C) + (size_t)C * d + (size_t)la_lines(H, C, W) * C) * sizeof(float);
}
inline size_t la_kside_bwd_smem(int H, int W, int C, int d) {
    return (8 * LA_THREADS + 2 * la_tile_floats(true, H, W, C) + 2 * (size_t)C * d + (size_t)la_lines(W, C, H) * C) * sizeof(float);
}

}  // namespace ddif
