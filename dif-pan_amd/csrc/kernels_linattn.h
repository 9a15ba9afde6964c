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
constexpr int LA_TILE_MAX = 8192;  // floats per staged tile (one line of len * C floats must fit); 32 elements per thread
constexpr int LA_EPT = LA_TILE_MAX / LA_THREADS;
__host__ __device__ inline int la_lines(int len, int C, int nlines) {
    int n = 6144 / (len * C);
    if (n < 1) n = 1;
    return n > nlines ? nlines : n;
}
__host__ __device__ inline int la_groups(int len, int C, int nlines) {
    const int nl = la_lines(len, C, nlines);
    return (nlines + nl - 1) / nl;
}

// Geometry of a line group: line l (0 .. nl-1) and position i (0 .. len-1) -> pixel line * ls + i * is of the H x W image
struct LaGeom {
    int len, ls, is, line0, nl;  // nl: valid lines of THIS group
};
__device__ __forceinline__ LaGeom la_geom(bool rows, int H, int W, int C, int g) {
    LaGeom G;
    const int nlines = rows ? H : W;
    G.len = rows ? W : H;
    G.ls = rows ? W : 1;
    G.is = rows ? 1 : W;
    const int per = la_lines(G.len, C, nlines);
    G.line0 = g * per;
    G.nl = nlines - G.line0 < per ? nlines - G.line0 : per;
    return G;
}
// tile[(l * len + i) * C + c] = src[pixel(l, i) * ld + coff + c]   (16-byte loads, C % 4 == 0, ld % 4 == 0)
__device__ __forceinline__ void la_load_tile(const float* src, int ld, int coff, const LaGeom& G, int C, float* tile) {
    const int C4 = C >> 2, n4 = G.nl * G.len * C4;
    for (int u0 = threadIdx.x; u0 < n4; u0 += LA_THREADS * 4) {  // four loads in flight per thread
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int u = u0 + k * LA_THREADS < n4 ? u0 + k * LA_THREADS : u0;
            const int c4 = u % C4, pi = u / C4, i = pi % G.len, l = pi / G.len;
            const size_t p = (size_t)(G.line0 + l) * G.ls + (size_t)i * G.is;
            v[k] = *reinterpret_cast<const float4*>(src + p * ld + coff + c4 * 4);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int u = u0 + k * LA_THREADS;
            if (u < n4) *reinterpret_cast<float4*>(tile + (size_t)u * 4) = v[k];
        }
    }
}
// softmax along i of every (line, channel) of the tile, in place: exp(x - max) / sum (the two-pass arithmetic of torch.softmax)
__device__ __forceinline__ void la_softmax_tile(float* tile, const LaGeom& G, int C) {
    for (int lc = threadIdx.x; lc < G.nl * C; lc += LA_THREADS) {
        const int c = lc % C, l = lc / C;
        float* p = tile + (size_t)l * G.len * C + c;
        float mx = -3.0e38f;
        for (int i = 0; i < G.len; ++i) mx = fmaxf(mx, p[(size_t)i * C]);
        float s = 0.f;
        for (int i = 0; i < G.len; ++i) {
            const float e = dd_exp(p[(size_t)i * C] - mx);
            p[(size_t)i * C] = e;
            s += e;
        }
        for (int i = 0; i < G.len; ++i) p[(size_t)i * C] = p[(size_t)i * C] / s;
    }
}
// part[(hd * d + a) * d + e] = scale * sum over the tile's pixels of A[pix][hd * d + a] * Bt[pix][hd * d + e]   (pixels in index order)
__device__ __forceinline__ void la_contract_tile(const float* A, const float* Bt, int npix, int C, int d, float scale, float* part) {
    for (int o = threadIdx.x; o < C * d; o += LA_THREADS) {
        const int e = o % d, ca = o / d, hd = ca / d;
        const float* pa = A + ca;
        const float* pb = Bt + hd * d + e;
        float s = 0.f;
        for (int n = 0; n < npix; ++n) s = fmaf(pa[(size_t)n * C], pb[(size_t)n * C], s);
        part[o] = s * scale;
    }
}

// ---- k side, forward: partial context of a group of rows.  grid = (row groups, B)
__global__ __launch_bounds__(LA_THREADS) void la_kside_fwd_kernel(const float* kv_pre, int H, int W, int C, int d, float* pctx /* [B][groups][C * d] */) {
    DDIF_DYN_SMEM(smem_);
    float* tk = reinterpret_cast<float*>(smem_);  // k rows -> k_sm
    const LaGeom G = la_geom(true, H, W, C, blockIdx.x);
    float* tv = tk + (size_t)la_lines(G.len, C, H) * G.len * C;
    const int b = blockIdx.y;
    const float* src = kv_pre + (size_t)b * H * W * 2 * C;
    la_load_tile(src, 2 * C, 0, G, C, tk);
    la_load_tile(src, 2 * C, C, G, C, tv);
    __syncthreads();
    la_softmax_tile(tk, G, C);
    __syncthreads();
    la_contract_tile(tk, tv, G.nl * G.len, C, d, 1.0f, pctx + ((size_t)b * gridDim.x + blockIdx.x) * C * d);
}

// out[b][o] = sum over the groups of part[b][g][o], in group order (eight loads in flight)
__global__ void la_reduce_kernel(const float* part, int B, int groups, int n, float* out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * n) return;
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

// ---- q side, forward: o = sc * ctx^T q_sm for a group of columns.  grid = (column groups, B)
__global__ __launch_bounds__(LA_THREADS) void la_qside_fwd_kernel(const float* q_pre, const float* ctx /* [B][C * d] */, int H, int W, int C, int d, float sc, float* out,
                                                                   int ld_o) {
    DDIF_DYN_SMEM(smem_);
    float* tq = reinterpret_cast<float*>(smem_);
    const LaGeom G = la_geom(false, H, W, C, blockIdx.x);
    float* cx = tq + (size_t)la_lines(G.len, C, W) * G.len * C;  // [C][d]: cx[(hd * d + a) * d + e]
    const int b = blockIdx.y;
    la_load_tile(q_pre + (size_t)b * H * W * C, C, 0, G, C, tq);
    for (int i = threadIdx.x; i < C * d; i += LA_THREADS) cx[i] = ctx[(size_t)b * C * d + i];
    __syncthreads();
    la_softmax_tile(tq, G, C);
    __syncthreads();
    const int nel = G.nl * G.len * C;
    float* ob = out + (size_t)b * H * W * ld_o;
    for (int u = threadIdx.x; u < nel; u += LA_THREADS) {
        const int c = u % C, pi = u / C, i = pi % G.len, l = pi / G.len;
        const int hd = c / d, e = c % d;
        const float* pq = tq + (size_t)pi * C + hd * d;
        const float* pc = cx + (size_t)hd * d * d + e;
        float s = 0.f;
        for (int a = 0; a < d; ++a) s = fmaf(pc[a * d], pq[a], s);
        const size_t p = (size_t)(G.line0 + l) * G.ls + (size_t)i * G.is;
        ob[p * ld_o + c] = s * sc;
    }
}

// ---- q side, backward (do given): dq_pre (final) and the partial dctx of a group of columns.  grid = (column groups, B)
//   dqs[a][n] = sc * sum_e ctx[a][e] do[e][n];   dq_pre = q_sm * (dqs - sum over the column of q_sm * dqs);   dctx[a][e] = sc * sum_n q_sm[a][n] do[e][n]
__global__ __launch_bounds__(LA_THREADS) void la_qside_bwd_kernel(const float* q_pre, const float* dout, int ld_g, const float* ctx, int H, int W, int C, int d, float sc,
                                                                   float* dq_pre, float* pdctx /* [B][groups][C * d] */) {
    DDIF_DYN_SMEM(smem_);
    float* tq = reinterpret_cast<float*>(smem_);
    const LaGeom G = la_geom(false, H, W, C, blockIdx.x);
    const size_t tile = (size_t)la_lines(G.len, C, W) * G.len * C;
    float* tg = tq + tile;        // do, later dqs
    float* cxT = tg + tile;       // [C][d] transposed inside a head: cxT[(hd * d + e) * d + a] = ctx[(hd * d + a) * d + e]
    float* cs = cxT + (size_t)C * d;  // [lines][C] column sums of q_sm * dqs
    const int b = blockIdx.y;
    la_load_tile(q_pre + (size_t)b * H * W * C, C, 0, G, C, tq);
    la_load_tile(dout + (size_t)b * H * W * ld_g, ld_g, 0, G, C, tg);
    for (int i = threadIdx.x; i < C * d; i += LA_THREADS) {
        const int a = i % d, ce = i / d, hd = ce / d, e = ce % d;
        cxT[i] = ctx[(size_t)b * C * d + (size_t)(hd * d + a) * d + e];
    }
    __syncthreads();
    la_softmax_tile(tq, G, C);
    __syncthreads();
    la_contract_tile(tq, tg, G.nl * G.len, C, d, sc, pdctx + ((size_t)b * gridDim.x + blockIdx.x) * C * d);
    const int nel = G.nl * G.len * C;
    float r[LA_EPT];
#pragma unroll
    for (int k = 0; k < LA_EPT; ++k) {
        const int u = threadIdx.x + k * LA_THREADS;
        r[k] = 0.f;
        if (u < nel) {
            const int c = u % C, pi = u / C, hd = c / d, a = c % d;
            const float* pg = tg + (size_t)pi * C + hd * d;
            const float* pc = cxT + (size_t)hd * d * d + a;
            float s = 0.f;
            for (int e = 0; e < d; ++e) s = fmaf(pc[e * d], pg[e], s);
            r[k] = s * sc;
        }
    }
    __syncthreads();  // every read of do is done
#pragma unroll
    for (int k = 0; k < LA_EPT; ++k) {
        const int u = threadIdx.x + k * LA_THREADS;
        if (u < nel) tg[u] = r[k];
    }
    __syncthreads();
    for (int lc = threadIdx.x; lc < G.nl * C; lc += LA_THREADS) {
        const int c = lc % C, l = lc / C;
        const size_t o = (size_t)l * G.len * C + c;
        float s = 0.f;
        for (int i = 0; i < G.len; ++i) s = fmaf(tg[o + (size_t)i * C], tq[o + (size_t)i * C], s);
        cs[lc] = s;
    }
    __syncthreads();
    float* db = dq_pre + (size_t)b * H * W * C;
#pragma unroll
    for (int k = 0; k < LA_EPT; ++k) {
        const int u = threadIdx.x + k * LA_THREADS;
        if (u < nel) {
            const int c = u % C, pi = u / C, i = pi % G.len, l = pi / G.len;
            const size_t p = (size_t)(G.line0 + l) * G.ls + (size_t)i * G.is;
            db[p * C + c] = tq[u] * (r[k] - cs[l * C + c]);
        }
    }
}

// ---- k side, backward: dk_pre, dv (final) of a group of rows.  grid = (row groups, B)
//   dks[a][n] = sum_e dctx[a][e] v[e][n];   dk_pre = k_sm * (dks - sum over the row of k_sm * dks);   dv[e][n] = sum_a k_sm[a][n] dctx[a][e]
__global__ __launch_bounds__(LA_THREADS) void la_kside_bwd_kernel(const float* kv_pre, const float* dctx /* [B][C * d] */, int H, int W, int C, int d, float* dkv_pre) {
    DDIF_DYN_SMEM(smem_);
    float* tk = reinterpret_cast<float*>(smem_);
    const LaGeom G = la_geom(true, H, W, C, blockIdx.x);
    const size_t tile = (size_t)la_lines(G.len, C, H) * G.len * C;
    float* tv = tk + tile;             // v, later dks
    float* dc = tv + tile;             // [C][d]  dc[(hd * d + a) * d + e]
    float* dcT = dc + (size_t)C * d;   // [C][d]  dcT[(hd * d + e) * d + a]
    float* rs = dcT + (size_t)C * d;   // [lines][C] row sums of k_sm * dks
    const int b = blockIdx.y;
    const float* src = kv_pre + (size_t)b * H * W * 2 * C;
    la_load_tile(src, 2 * C, 0, G, C, tk);
    la_load_tile(src, 2 * C, C, G, C, tv);
    for (int i = threadIdx.x; i < C * d; i += LA_THREADS) {
        const float v = dctx[(size_t)b * C * d + i];
        const int e = i % d, ca = i / d, hd = ca / d, a = ca % d;
        dc[i] = v;
        dcT[(size_t)(hd * d + e) * d + a] = v;
    }
    __syncthreads();
    la_softmax_tile(tk, G, C);
    __syncthreads();
    const int nel = G.nl * G.len * C;
    float* ob = dkv_pre + (size_t)b * H * W * 2 * C;
    float r[LA_EPT];
#pragma unroll
    for (int k = 0; k < LA_EPT; ++k) {
        const int u = threadIdx.x + k * LA_THREADS;
        r[k] = 0.f;
        if (u < nel) {
            const int c = u % C, pi = u / C, i = pi % G.len, l = pi / G.len, hd = c / d, j = c % d;
            const float* pv = tv + (size_t)pi * C + hd * d;
            const float* pk = tk + (size_t)pi * C + hd * d;
            const float* p1 = dcT + (size_t)hd * d * d + j;  // dctx[hd][a = j][e] at p1[e * d]
            const float* p2 = dc + (size_t)hd * d * d + j;   // dctx[hd][a][e = j] at p2[a * d]
            float s1 = 0.f, s2 = 0.f;
            for (int t = 0; t < d; ++t) {
                s1 = fmaf(p1[t * d], pv[t], s1);
                s2 = fmaf(p2[t * d], pk[t], s2);
            }
            r[k] = s1;  // dks of channel c
            const size_t p = (size_t)(G.line0 + l) * G.ls + (size_t)i * G.is;
            ob[p * 2 * C + C + c] = s2;  // dv
        }
    }
    __syncthreads();  // every read of v is done
#pragma unroll
    for (int k = 0; k < LA_EPT; ++k) {
        const int u = threadIdx.x + k * LA_THREADS;
        if (u < nel) tv[u] = r[k];
    }
    __syncthreads();
    for (int lc = threadIdx.x; lc < G.nl * C; lc += LA_THREADS) {
        const int c = lc % C, l = lc / C;
        const size_t o = (size_t)l * G.len * C + c;
        float s = 0.f;
        for (int i = 0; i < G.len; ++i) s = fmaf(tv[o + (size_t)i * C], tk[o + (size_t)i * C], s);
        rs[lc] = s;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < LA_EPT; ++k) {
        const int u = threadIdx.x + k * LA_THREADS;
        if (u < nel) {
            const int c = u % C, pi = u / C, i = pi % G.len, l = pi / G.len;
            const size_t p = (size_t)(G.line0 + l) * G.ls + (size_t)i * G.is;
            ob[p * 2 * C + c] = tk[u] * (r[k] - rs[l * C + c]);
        }
    }
}

// dynamic LDS of the four kernels (floats -> bytes)
inline size_t la_tile_floats(bool rows, int H, int W, int C) {
    const int len = rows ? W : H, nlines = rows ? H : W;
    return (size_t)la_lines(len, C, nlines) * len * C;
}
inline size_t la_kside_fwd_smem(int H, int W, int C) { return 2 * la_tile_floats(true, H, W, C) * sizeof(float); }
inline size_t la_qside_fwd_smem(int H, int W, int C, int d) { return (la_tile_floats(false, H, W, C) + (size_t)C * d) * sizeof(float); }
inline size_t la_qside_bwd_smem(int H, int W, int C, int d) {
    return (2 * la_tile_floats(false, H, W, C) + (size_t)C * d + (size_t)la_lines(H, C, W) * C) * sizeof(float);
}
inline size_t la_kside_bwd_smem(int H, int W, int C, int d) {
    return (2 * la_tile_floats(true, H, W, C) + 2 * (size_t)C * d + (size_t)la_lines(W, C, H) * C) * sizeof(float);
}

}  // namespace ddif
