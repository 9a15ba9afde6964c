// Backward kernels of the non-convolution pieces of the denoiser (reference: autograd under loss.backward(),
// diffusion_engine.py:233).  Correctness first: plain fp32 / fp64 loops on NCHW tensors (the boundary layout), one workgroup per
// natural unit, fixed-order reductions, no atomics.  None of them is a bottleneck of the backward pass (the convolutions are).
//
//   dwconv3x3_bwd_*     depthwise 3x3 (groups = C, pad 1, no bias): FastAttnCondInjection.q[0] / kv[0]      models/sr3_dwt.py:507-520
//   film_bwd_kernel     out = xc (1 + scale) + shift: CondInjection                                        :395-396
//   selfattn_bwd_kernel SelfAttention core (scores, softmax, weighted sum) per (sample, head)              :345-358
//   linattn_bwd_kernel  FastAttnCondInjection core: softmax_H(q), softmax_W(k), ctx = k v^T, out = ctx^T q  :545-566
//   linear_bwd_*        nn.Linear: noise_level_mlp and every FeatureWiseAffine                             :59-64, 241-258
//   gn_bwd_*            GroupNorm(1 group) alone, for a normalised tensor with several consumers (prenorm_x)  :540-573
//   l1_bwd_kernel       F.l1_loss(reduction='mean') backward                                               diffusion/diffusion_ddpm_pan.py:742-749
#pragma once
#include "ddif_dev.h"

namespace ddif {

// ---------------------------------------------------------------------------------------------------------------- depthwise 3x3
__global__ void dwconv3x3_bwd_dx_kernel(const float* dy, const float* w /* (C,1,3,3) */, int B, int C, int H, int W, float* dx) {
    const size_t total = (size_t)B * C * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const int c = (int)((i / ((size_t)W * H)) % C);
        const float* plane = dy + (i / ((size_t)W * H)) * H * W;
        float s = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int oy = y + 1 - ky, ox = x + 1 - kx;  // the output pixel that read (y, x) through tap (ky, kx)
                if (oy >= 0 && oy < H && ox >= 0 && ox < W) s = fmaf(plane[oy * W + ox], w[c * 9 + ky * 3 + kx], s);
            }
        dx[i] = s;
    }
}
// one workgroup per channel: dw[c][tap] = sum over (b, y, x) of dy[b,c,y,x] * x[b,c,y+ky-1,x+kx-1]
__global__ __launch_bounds__(256) void dwconv3x3_bwd_dw_kernel(const float* x, const float* dy, int B, int C, int H, int W, float* dw) {
    DDIF_DYN_SMEM(smem_);
    double* red = reinterpret_cast<double*>(smem_);  // [9][256]
    const int c = blockIdx.x, tid = threadIdx.x;
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int HW = H * W;
    for (int i = tid; i < B * HW; i += 256) {
        const int b = i / HW, p = i % HW, y = p / W, xx = p % W;
        const float g = dy[((size_t)b * C + c) * HW + p];
        const float* plane = x + ((size_t)b * C + c) * HW;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = y + ky - 1, ix = xx + kx - 1;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) acc[ky * 3 + kx] += (double)g * (double)plane[iy * W + ix];
            }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) red[k * 256 + tid] = acc[k];
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (tid < st)
#pragma unroll
            for (int k = 0; k < 9; ++k) red[k * 256 + tid] += red[k * 256 + tid + st];
        __syncthreads();
    }
    if (tid < 9) dw[c * 9 + tid] = (float)red[tid * 256];
}

// ---------------------------------------------------------------------------------------------------------------- FiLM
// ss = (B, 2C, H, W): scale = channels [0, C), shift = [C, 2C) (y.chunk(2, dim=1)); out = xc * (1 + scale) + shift
__global__ void film_bwd_kernel(const float* xc, const float* ss, const float* dout, int B, int C, int HW, float* dxc, float* dss) {
    const size_t total = (size_t)B * C * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i % HW, c = (i / HW) % C, b = i / ((size_t)HW * C);
        const size_t is = (b * 2 * C + c) * HW + p, ih = (b * 2 * C + C + c) * HW + p;
        const float g = dout[i];
        if (dxc) dxc[i] = g * (1.f + ss[is]);
        if (dss) {
            dss[is] = g * xc[i];
            dss[ih] = g;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- SelfAttention core
// qkv = (B, heads, 3d, n) with [q | k | v] along the 3d axis (the reference's view + chunk); o[c][p] = sum_q a[p][q] v[c][q],
// a = softmax_q(q^T k * sc).  One workgroup per (sample, head); n <= 64, d <= 32 (the engine's bottleneck: n = 64, d = 16).
// LDS: q, k, v, do [d][n] + a [n][n] + ds [n][n].
__global__ __launch_bounds__(256) void selfattn_bwd_kernel(const float* qkv, const float* dout, int heads, int d, int n, float sc, float* dqkv) {
    DDIF_DYN_SMEM(smem_);
    float* qs = reinterpret_cast<float*>(smem_);
    float* ks = qs + d * n;
    float* vs = ks + d * n;
    float* gs = vs + d * n;   // do
    float* as = gs + d * n;   // [n][n]
    float* ds = as + n * n;   // [n][n]
    float* rd = ds + n * n;   // [n] row dots
    const int tid = threadIdx.x;
    const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
    const float* base = qkv + ((size_t)b * heads + hd) * 3 * d * n;
    const float* gbase = dout + ((size_t)b * heads + hd) * d * n;
    for (int i = tid; i < d * n; i += 256) {
        qs[i] = base[i];
        ks[i] = base[d * n + i];
        vs[i] = base[2 * d * n + i];
        gs[i] = gbase[i];
    }
    __syncthreads();
    for (int i = tid; i < n * n; i += 256) {  // scores and d(a)
        const int p = i / n, q = i % n;
        float s = 0.f, da = 0.f;
        for (int c = 0; c < d; ++c) {
            s = fmaf(qs[c * n + p], ks[c * n + q], s);
            da = fmaf(gs[c * n + p], vs[c * n + q], da);
        }
        as[i] = s * sc;
        ds[i] = da;
    }
    __syncthreads();
    for (int p = tid; p < n; p += 256) {  // softmax of row p, then ds = a (da - sum a da)
        float mx = -3.0e38f;
        for (int q = 0; q < n; ++q) mx = fmaxf(mx, as[p * n + q]);
        float sum = 0.f;
        for (int q = 0; q < n; ++q) {
            const float e = dd_exp(as[p * n + q] - mx);
            as[p * n + q] = e;
            sum += e;
        }
        const float inv = 1.0f / sum;
        float dot = 0.f;
        for (int q = 0; q < n; ++q) {
            as[p * n + q] *= inv;
            dot = fmaf(as[p * n + q], ds[p * n + q], dot);
        }
        rd[p] = dot;
    }
    __syncthreads();
    float* dq = dqkv + ((size_t)b * heads + hd) * 3 * d * n;
    for (int i = tid; i < d * n; i += 256) {  // dv[c][q] = sum_p a[p][q] do[c][p]
        const int c = i / n, q = i % n;
        float s = 0.f;
        for (int p = 0; p < n; ++p) s = fmaf(as[p * n + q], gs[c * n + p], s);
        dq[2 * d * n + i] = s;
    }
    __syncthreads();
    for (int i = tid; i < n * n; i += 256) ds[i] = as[i] * (ds[i] - rd[i / n]);
    __syncthreads();
    for (int i = tid; i < d * n; i += 256) {
        const int c = i / n, p = i % n;
        float s1 = 0.f, s2 = 0.f;
        for (int q = 0; q < n; ++q) {
            s1 = fmaf(ds[p * n + q], ks[c * n + q], s1);  // dq[c][p] = sc sum_q ds[p][q] k[c][q]
            s2 = fmaf(ds[q * n + p], qs[c * n + q], s2);  // dk[c][p] = sc sum_q ds[q][p] q[c][q]
        }
        dq[i] = s1 * sc;
        dq[d * n + i] = s2 * sc;
    }
}

// ---------------------------------------------------------------------------------------------------------------- linear attention core
// Per (sample, head), d = qd / heads <= 32, image H x W (W <= 64, H <= 64):
//   q = softmax over H of q_pre (per channel and column), times sc = 1/sqrt(d);  k = softmax over W of k_pre (per channel and row)
//   ctx[a][e] = sum_n k[a][n] v[e][n];   o[e][n] = sum_a ctx[a][e] q[a][n]
// Backward (do given):  dctx[a][e] = sum_n q[a][n] do[e][n];  dq = ctx do;  dk = dctx v;  dv = dctx^T k;  then the two softmax backwards.
// Pass 1 (rows): softmax statistics.  Pass 2 (rows): ctx, dctx.  Pass 3 (rows): dv (final), dk_pre (final: its softmax lives inside the
// row), dq (raw, parked in the output) and the column sums T[a][x] = sum_y dq q.  Pass 4: dq_pre = q_sm (dq - T).
// One workgroup per (sample, head); every LDS cell has one owner thread, rows are walked in order: deterministic.
__global__ __launch_bounds__(256) void linattn_bwd_kernel(const float* q_pre, const float* kv_pre, const float* dout, int heads, int d, int H, int W, float sc,
                                                          float* dq_pre, float* dkv_pre) {
    DDIF_DYN_SMEM(smem_);
    float* qmx = reinterpret_cast<float*>(smem_);  // [d][W] column max of q_pre
    float* qsm = qmx + d * W;                      // [d][W] column sum of exp
    float* T = qsm + d * W;                        // [d][W] column sums of dq * q_sm
    float* ctx = T + d * W;                        // [d][d]
    float* dctx = ctx + d * d;                     // [d][d]
    float* rk = dctx + d * d;                      // [d][W] k softmax of the current row
    float* rv = rk + d * W;                        // [d][W]
    float* rq = rv + d * W;                        // [d][W] q softmax * sc of the current row
    float* rg = rq + d * W;                        // [d][W] do of the current row
    float* rdk = rg + d * W;                       // [d][W] dk of the current row
    float* rdot = rdk + d * W;                     // [d] row dots of the k softmax backward
    float* kmx = rdot + d;                         // [d][H] row max of k_pre (softmax over W)
    float* ksm = kmx + d * H;                      // [d][H] row sum of exp
    const int tid = threadIdx.x;
    const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
    const int qd = heads * d, HW = H * W;
    const float* qb = q_pre + ((size_t)b * qd + hd * d) * HW;
    const float* kb = kv_pre + ((size_t)b * 2 * qd + hd * d) * HW;
    const float* vb = kv_pre + ((size_t)b * 2 * qd + qd + hd * d) * HW;
    const float* gb = dout + ((size_t)b * qd + hd * d) * HW;
    float* dqb = dq_pre + ((size_t)b * qd + hd * d) * HW;
    float* dkb = dkv_pre + ((size_t)b * 2 * qd + hd * d) * HW;
    float* dvb = dkv_pre + ((size_t)b * 2 * qd + qd + hd * d) * HW;
    // pass 1: column statistics of q_pre (softmax over H)
    for (int i = tid; i < d * W; i += 256) {
        const int a = i / W, x = i % W;
        float mx = -3.0e38f;
        for (int y = 0; y < H; ++y) mx = fmaxf(mx, qb[(size_t)a * HW + y * W + x]);
        float s = 0.f;
        for (int y = 0; y < H; ++y) s += dd_exp(qb[(size_t)a * HW + y * W + x] - mx);
        qmx[i] = mx;
        qsm[i] = s;
        T[i] = 0.f;
    }
    for (int i = tid; i < d * d; i += 256) {
        ctx[i] = 0.f;
        dctx[i] = 0.f;
    }
    for (int i = tid; i < d * H; i += 256) {  // row statistics of k_pre for every (channel, row): one thread per pair
        const int a = i / H, y = i % H;
        float mx = -3.0e38f;
        for (int x = 0; x < W; ++x) mx = fmaxf(mx, kb[(size_t)a * HW + y * W + x]);
        float sm = 0.f;
        for (int x = 0; x < W; ++x) sm += dd_exp(kb[(size_t)a * HW + y * W + x] - mx);
        kmx[i] = mx;
        ksm[i] = sm;
    }
    __syncthreads();
    auto load_row = [&](int y) {  // k softmax (over this row), v, q softmax * sc, do  -> LDS
        for (int i = tid; i < d * W; i += 256) {
            const int a = i / W, x = i % W;
            const size_t e = (size_t)a * HW + y * W + x;
            rk[i] = dd_exp(kb[e] - kmx[a * H + y]) / ksm[a * H + y];
            rv[i] = vb[e];
            rq[i] = dd_exp(qb[e] - qmx[i]) / qsm[i] * sc;
            rg[i] = gb[e];
        }
        __syncthreads();
    };
    // pass 2: ctx and dctx (each (a, e) cell has one owner thread; rows in order)
    for (int y = 0; y < H; ++y) {
        load_row(y);
        for (int i = tid; i < d * d; i += 256) {
            const int a = i / d, e = i % d;
            float s1 = ctx[i], s2 = dctx[i];
            for (int x = 0; x < W; ++x) {
                s1 = fmaf(rk[a * W + x], rv[e * W + x], s1);
                s2 = fmaf(rq[a * W + x], rg[e * W + x], s2);
            }
            ctx[i] = s1;
            dctx[i] = s2;
        }
        __syncthreads();
    }
    // pass 3
    for (int y = 0; y < H; ++y) {
        load_row(y);
        for (int i = tid; i < d * W; i += 256) {
            const int a = i / W, x = i % W;
            float dq = 0.f, dk = 0.f, dv = 0.f;
            for (int e = 0; e < d; ++e) {
                dq = fmaf(ctx[a * d + e], rg[e * W + x], dq);    // dq[a][n] = sum_e ctx[a][e] do[e][n]
                dk = fmaf(dctx[a * d + e], rv[e * W + x], dk);   // dk[a][n] = sum_e dctx[a][e] v[e][n]
                dv = fmaf(dctx[e * d + a], rk[e * W + x], dv);   // dv[a][n] = sum_e dctx[e][a] k[e][n]
            }
            const size_t el = (size_t)a * HW + y * W + x;
            dvb[el] = dv;
            rdk[i] = dk;
            dq *= sc;  // d(q_sm) of o = ctx^T (q_sm * sc)
            dqb[el] = dq;
            T[i] += dq * (rq[i] / sc);  // q_sm = rq / sc
        }
        __syncthreads();
        for (int a = tid; a < d; a += 256) {
            float s = 0.f;
            for (int x = 0; x < W; ++x) s = fmaf(rdk[a * W + x], rk[a * W + x], s);
            rdot[a] = s;
        }
        __syncthreads();
        for (int i = tid; i < d * W; i += 256) {
            const int a = i / W, x = i % W;
            dkb[(size_t)a * HW + y * W + x] = rk[i] * (rdk[i] - rdot[a]);
        }
        __syncthreads();
    }
    // pass 4: softmax-over-H backward of q
    for (int i = tid; i < d * W; i += 256) {
        const int a = i / W, x = i % W;
        for (int y = 0; y < H; ++y) {
            const size_t el = (size_t)a * HW + y * W + x;
            const float qs_ = dd_exp(qb[el] - qmx[i]) / qsm[i];
            dqb[el] = qs_ * (dqb[el] - T[i]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- nn.Linear
// y = x W^T + b:  dx (B, in) = dy W;  dW (out, in) = dy^T x;  db (out) = sum_b dy.  One thread per output element (tiny matrices).
__global__ void linear_bwd_kernel(const float* x, const float* w, const float* dy, int B, int nin, int nout, float* dx, float* dw, float* db) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_dx = dx ? B * nin : 0, n_dw = dw ? nout * nin : 0, n_db = db ? nout : 0;
    if (i < n_dx) {
        const int b = i / nin, k = i % nin;
        float s = 0.f;
        for (int o = 0; o < nout; ++o) s = fmaf(dy[b * nout + o], w[o * nin + k], s);
        dx[i] = s;
    } else if (i < n_dx + n_dw) {
        const int j = i - n_dx, o = j / nin, k = j % nin;
        float s = 0.f;
        for (int b = 0; b < B; ++b) s = fmaf(dy[b * nout + o], x[b * nin + k], s);
        dw[j] = s;
    } else if (i < n_dx + n_dw + n_db) {
        const int o = i - n_dx - n_dw;
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dy[b * nout + o];
        db[o] = s;
    }
}
// Swish between the two Linear layers of noise_level_mlp (:61-63): dx = dy * d/dx (x sigmoid(x))
__global__ void swish_bwd_kernel(const float* x, const float* dy, size_t n, float* dx) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float y = x[i], s = 1.0f / (1.0f + dd_exp(-y));
        dx[i] = dy[i] * (s * (1.f + y * (1.f - s)));
    }
}

// ---------------------------------------------------------------------------------------------------------------- L1 loss
// loss = mean |pred - target|: dpred = sign(pred - target) * upstream / n   (torch: sign(0) = 0)
__global__ void l1_bwd_kernel(const float* pred, const float* target, size_t n, float upstream, float* dpred) {
    const float g = upstream / (float)n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float df = pred[i] - target[i];
        dpred[i] = df > 0.f ? g : (df < 0.f ? -g : 0.f);
    }
}

// ---------------------------------------------------------------------------------------------------------------- GroupNorm alone
// GroupNorm(1 group) whose output feeds more than one consumer (FastAttnCondInjection.prenorm_x -> q[0] and attn_res, :540-573): the
// caller sums the consumers' gradients into dy.  NCHW.  ws (doubles): [B][C][2] plane sums {sum dy, sum dy x_hat} then [B][4] =
// {mean, rstd, sum_c gamma_c P0 / N, sum_c gamma_c P1 / N}.  One workgroup per sample, channels in order, fixed-order trees.
__global__ __launch_bounds__(256) void gn_bwd_sample_kernel(const float* x, const float* dy, const float* gamma, int C, int HW, double* ws, int B) {
    DDIF_DYN_SMEM(smem_);
    double* red = reinterpret_cast<double*>(smem_);  // [2][256]
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* xb = x + (size_t)b * C * HW;
    const float* gb = dy + (size_t)b * C * HW;
    double* P = ws + (size_t)b * C * 2;
    double* M = ws + (size_t)B * C * 2 + (size_t)b * 4;
    auto tree = [&](double v0, double v1, double* o0, double* o1) {
        red[tid] = v0;
        red[256 + tid] = v1;
        __syncthreads();
        for (int st = 128; st >= 1; st >>= 1) {
            if (tid < st) {
                red[tid] += red[tid + st];
                red[256 + tid] += red[256 + tid + st];
            }
            __syncthreads();
        }
        *o0 = red[0];
        *o1 = red[256];
        __syncthreads();
    };
    double s1 = 0.0, s2 = 0.0;
    for (size_t i = tid; i < (size_t)C * HW; i += 256) {
        const double v = xb[i];
        s1 += v;
        s2 += v * v;
    }
    double t1, t2;
    tree(s1, s2, &t1, &t2);
    const double n = (double)C * HW, mean = t1 / n;
    double var = t2 / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + DDIF_GN_EPS);
    const float meanf = (float)mean, rstdf = (float)rstd;
    double a1 = 0.0, a2 = 0.0;
    for (int c = 0; c < C; ++c) {
        double p0 = 0.0, p1 = 0.0;
        for (int i = tid; i < HW; i += 256) {
            const float g = gb[(size_t)c * HW + i], xh = (xb[(size_t)c * HW + i] - meanf) * rstdf;
            p0 += (double)g;
            p1 += (double)g * (double)xh;
        }
        double q0, q1;
        tree(p0, p1, &q0, &q1);
        if (tid == 0) {
            P[c * 2 + 0] = q0;
            P[c * 2 + 1] = q1;
        }
        a1 += (double)gamma[c] * q0;
        a2 += (double)gamma[c] * q1;
    }
    if (tid == 0) {
        M[0] = mean;
        M[1] = rstd;
        M[2] = a1 / n;
        M[3] = a2 / n;
    }
}
__global__ void gn_bwd_dx_kernel(const float* x, const float* dy, const float* gamma, const double* ws, int B, int C, int HW, float* dx) {
    const size_t total = (size_t)B * C * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)((i / HW) % C);
        const size_t b = i / ((size_t)HW * C);
        const double* M = ws + (size_t)B * C * 2 + b * 4;
        const float mean = (float)M[0], rstd = (float)M[1], m1 = (float)M[2], m2 = (float)M[3];
        const float xh = (x[i] - mean) * rstd;
        dx[i] = rstd * (gamma[c] * dy[i] - m1 - xh * m2);
    }
}
__global__ void gn_bwd_affine_kernel(const double* ws, int B, int C, float* dgamma, float* dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double g = 0.0, bt = 0.0;
    for (int b = 0; b < B; ++b) {
        bt += ws[((size_t)b * C + c) * 2 + 0];
        g += ws[((size_t)b * C + c) * 2 + 1];
    }
    if (dgamma) dgamma[c] = (float)g;
    if (dbeta) dbeta[c] = (float)bt;
}

}  // namespace ddif

// ================================================================================================================
// Forward counterparts used by the op-by-op training tape (tests/train_tape.py): the inference plan fuses these into its conv prologues /
// epilogues and never materialises what the backward pass needs, so the training forward runs them un-fused, NCHW, saving
// every intermediate.  Same arithmetic as the reference modules (models/sr3_dwt.py); correctness first.
namespace ddif {

__global__ void dwconv3x3_fwd_kernel(const float* x, const float* w /* (C,1,3,3) */, int B, int C, int H, int W, float* y) {
    const size_t total = (size_t)B * C * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % W), yy = (int)((i / W) % H);
        const int c = (int)((i / ((size_t)W * H)) % C);
        const float* plane = x + (i / ((size_t)W * H)) * H * W;
        float s = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = yy + ky - 1, ix = xx + kx - 1;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) s = fmaf(plane[iy * W + ix], w[c * 9 + ky * 3 + kx], s);
            }
        y[i] = s;
    }
}
// GroupNorm(1 group, eps 1e-5) [+ SiLU] [+ dropout mask], NCHW; one workgroup per sample (fp64 statistics, fixed-order tree)
__global__ __launch_bounds__(256) void gn_fwd_kernel(const float* x, const float* gamma, const float* beta, const float* mask, int C, int HW, int silu, float* y) {
    DDIF_DYN_SMEM(smem_);
    double* red = reinterpret_cast<double*>(smem_);  // [2][256]
    const int b = blockIdx.x, tid = threadIdx.x;
    const size_t n = (size_t)C * HW;
    const float* xb = x + (size_t)b * n;
    double s1 = 0.0, s2 = 0.0;
    for (size_t i = tid; i < n; i += 256) {
        const double v = xb[i];
        s1 += v;
        s2 += v * v;
    }
    red[tid] = s1;
    red[256 + tid] = s2;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (tid < st) {
            red[tid] += red[tid + st];
            red[256 + tid] += red[256 + tid + st];
        }
        __syncthreads();
    }
    const double mean = red[0] / (double)n;
    double var = red[256] / (double)n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float meanf = (float)mean, rstd = (float)(1.0 / sqrt(var + DDIF_GN_EPS));
    for (size_t i = tid; i < n; i += 256) {
        const int c = (int)(i / HW);
        float v = fmaf((xb[i] - meanf) * rstd, gamma[c], beta[c]);
        if (silu) v = dd_silu(v);
        if (mask) v *= mask[(size_t)b * n + i];
        y[(size_t)b * n + i] = v;
    }
}
// y = x * sigmoid(x) (accurate exp: the time MLP and the FFN SiLU of the training graph)
__global__ void swish_fwd_kernel(const float* x, size_t n, float* y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = x[i] / (1.0f + dd_exp(-x[i]));
}
__global__ void film_fwd_kernel(const float* xc, const float* ss, int B, int C, int HW, float* out) {
    const size_t total = (size_t)B * C * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i % HW, c = (i / HW) % C, b = i / ((size_t)HW * C);
        out[i] = xc[i] * (1.f + ss[(b * 2 * C + c) * HW + p]) + ss[(b * 2 * C + C + c) * HW + p];
    }
}
// out = a + alpha[b] * f   (residual adds; alpha = NULL: 1; the DropPath row scale otherwise)
__global__ void add_scaled_kernel(const float* a, const float* f, const float* alpha, size_t per_sample, size_t total, float* out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        out[i] = a[i] + (alpha ? alpha[i / per_sample] : 1.f) * f[i];
}
__global__ void linear_fwd_kernel(const float* x, const float* w, const float* bias, int B, int nin, int nout, float* y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * nout) return;
    const int b = i / nout, o = i % nout;
    float s = bias ? bias[o] : 0.f;
    for (int k = 0; k < nin; ++k) s = fmaf(x[b * nin + k], w[o * nin + k], s);
    y[i] = s;
}
// SelfAttention core forward (see selfattn_bwd_kernel): o[c][p] = sum_q softmax_q(q^T k sc)[p][q] v[c][q]
__global__ __launch_bounds__(256) void selfattn_fwd_kernel(const float* qkv, int heads, int d, int n, float sc, float* out) {
    DDIF_DYN_SMEM(smem_);
    float* qs = reinterpret_cast<float*>(smem_);
    float* ks = qs + d * n;
    float* vs = ks + d * n;
    float* as = vs + d * n;  // [n][n]
    const int tid = threadIdx.x;
    const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
    const float* base = qkv + ((size_t)b * heads + hd) * 3 * d * n;
    for (int i = tid; i < d * n; i += 256) {
        qs[i] = base[i];
        ks[i] = base[d * n + i];
        vs[i] = base[2 * d * n + i];
    }
    __syncthreads();
    for (int i = tid; i < n * n; i += 256) {
        const int p = i / n, q = i % n;
        float s = 0.f;
        for (int c = 0; c < d; ++c) s = fmaf(qs[c * n + p], ks[c * n + q], s);
        as[i] = s * sc;
    }
    __syncthreads();
    for (int p = tid; p < n; p += 256) {
        float mx = -3.0e38f;
        for (int q = 0; q < n; ++q) mx = fmaxf(mx, as[p * n + q]);
        float sum = 0.f;
        for (int q = 0; q < n; ++q) {
            const float e = dd_exp(as[p * n + q] - mx);
            as[p * n + q] = e;
            sum += e;
        }
        const float inv = 1.0f / sum;
        for (int q = 0; q < n; ++q) as[p * n + q] *= inv;
    }
    __syncthreads();
    float* ob = out + ((size_t)b * heads + hd) * d * n;
    for (int i = tid; i < d * n; i += 256) {
        const int c = i / n, p = i % n;
        float s = 0.f;
        for (int q = 0; q < n; ++q) s = fmaf(as[p * n + q], vs[c * n + q], s);
        ob[i] = s;
    }
}
// FastAttnCondInjection core forward (see linattn_bwd_kernel)
__global__ __launch_bounds__(256) void linattn_fwd_kernel(const float* q_pre, const float* kv_pre, int heads, int d, int H, int W, float sc, float* out) {
    DDIF_DYN_SMEM(smem_);
    float* qmx = reinterpret_cast<float*>(smem_);  // [d][W]
    float* qsm = qmx + d * W;                      // [d][W]
    float* ctx = qsm + d * W;                      // [d][d]
    float* rk = ctx + d * d;                       // [d][W]
    float* rv = rk + d * W;                        // [d][W]
    float* rmx = rv + d * W;                       // [d][H] row max of k_pre
    float* rsm = rmx + d * H;                      // [d][H] row sum of exp
    const int tid = threadIdx.x;
    const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
    const int qd = heads * d, HW = H * W;
    const float* qb = q_pre + ((size_t)b * qd + hd * d) * HW;
    const float* kb = kv_pre + ((size_t)b * 2 * qd + hd * d) * HW;
    const float* vb = kv_pre + ((size_t)b * 2 * qd + qd + hd * d) * HW;
    float* ob = out + ((size_t)b * qd + hd * d) * HW;
    for (int i = tid; i < d * W; i += 256) {
        const int a = i / W, x = i % W;
        float mx = -3.0e38f;
        for (int y = 0; y < H; ++y) mx = fmaxf(mx, qb[(size_t)a * HW + y * W + x]);
        float s = 0.f;
        for (int y = 0; y < H; ++y) s += dd_exp(qb[(size_t)a * HW + y * W + x] - mx);
        qmx[i] = mx;
        qsm[i] = s;
    }
    for (int i = tid; i < d * d; i += 256) ctx[i] = 0.f;
    for (int i = tid; i < d * H; i += 256) {  // row statistics of k_pre for every (channel, row)
        const int a = i / H, y = i % H;
        float mx = -3.0e38f;
        for (int x = 0; x < W; ++x) mx = fmaxf(mx, kb[(size_t)a * HW + y * W + x]);
        float s = 0.f;
        for (int x = 0; x < W; ++x) s += dd_exp(kb[(size_t)a * HW + y * W + x] - mx);
        rmx[i] = mx;
        rsm[i] = s;
    }
    __syncthreads();
    for (int y = 0; y < H; ++y) {
        for (int i = tid; i < d * W; i += 256) {
            const int a = i / W, x = i % W;
            rk[i] = dd_exp(kb[(size_t)a * HW + y * W + x] - rmx[a * H + y]) / rsm[a * H + y];
            rv[i] = vb[(size_t)a * HW + y * W + x];
        }
        __syncthreads();
        for (int i = tid; i < d * d; i += 256) {
            const int a = i / d, e = i % d;
            float s = ctx[i];
            for (int x = 0; x < W; ++x) s = fmaf(rk[a * W + x], rv[e * W + x], s);
            ctx[i] = s;
        }
        __syncthreads();
    }
    for (int i = tid; i < d * HW; i += 256) {  // o[e][n] = sum_a ctx[a][e] q_sm[a][n] sc
        const int e = i / HW, nn = i % HW, x = nn % W;
        float s = 0.f;
        for (int a = 0; a < d; ++a) s = fmaf(ctx[a * d + e], dd_exp(qb[(size_t)a * HW + nn] - qmx[a * W + x]) / qsm[a * W + x] * sc, s);
        ob[i] = s;
    }
}
}  // namespace ddif

namespace ddif {
// q_sample (diffusion/diffusion_ddpm_pan.py:668-681): x_t = a[b] * x0 + s[b] * noise (layout-agnostic: elementwise per sample)
__global__ void q_sample_ew_kernel(const float* x0, const float* noise, const float* a, const float* s, int B, size_t per, float* out) {
#pragma clang fp contract(off)
    const size_t total = (size_t)B * per;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / per;
        out[i] = a[b] * x0[i] + s[b] * noise[i];
    }
}
// mean |pred - target| (F.l1_loss): one workgroup, fp64 thread partials + fixed-order tree
__global__ __launch_bounds__(256) void l1_fwd_kernel(const float* pred, const float* target, size_t n, float* out) {
    DDIF_DYN_SMEM(smem_);
    double* red = reinterpret_cast<double*>(smem_);
    const int tid = threadIdx.x;
    double s = 0.0;
    for (size_t i = tid; i < n; i += 256) s += (double)fabsf(pred[i] - target[i]);
    red[tid] = s;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (tid < st) red[tid] += red[tid + st];
        __syncthreads();
    }
    if (tid == 0) out[0] = (float)(red[0] / (double)n);
}
}  // namespace ddif

namespace ddif {
// ---- wider-grid forms of the per-sample kernels above (a 32-workgroup launch leaves 7/8 of the GPU idle) -------------------------------
// GroupNorm statistics: grid (nchunk, B) -> part[b][chunk][2] (fp64 sums of x and x^2 over a flat slice of the sample)
__global__ __launch_bounds__(256) void gn_stats_partial_kernel(const float* x, size_t per_sample, int nchunk, double* part) {
    DDIF_DYN_SMEM(smem_);
    double* red = reinterpret_cast<double*>(smem_);  // [2][256]
    const int b = blockIdx.y, tid = threadIdx.x;
    const size_t per = (per_sample + nchunk - 1) / nchunk;
    const size_t i0 = (size_t)blockIdx.x * per, i1 = i0 + per < per_sample ? i0 + per : per_sample;
    double s1 = 0.0, s2 = 0.0;
    for (size_t i = i0 + tid; i < i1; i += 256) {
        const double v = x[(size_t)b * per_sample + i];
        s1 += v;
        s2 += v * v;
    }
    red[tid] = s1;
    red[256 + tid] = s2;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (tid < st) {
            red[tid] += red[tid + st];
            red[256 + tid] += red[256 + tid + st];
        }
        __syncthreads();
    }
    if (tid == 0) {
        part[((size_t)b * nchunk + blockIdx.x) * 2 + 0] = red[0];
        part[((size_t)b * nchunk + blockIdx.x) * 2 + 1] = red[256];
    }
}
__device__ __forceinline__ void gn_stats_from_partials(const double* part, int b, int nchunk, double n, float* mean, float* rstd) {
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < nchunk; ++k) {
        s1 += part[((size_t)b * nchunk + k) * 2 + 0];
        s2 += part[((size_t)b * nchunk + k) * 2 + 1];
    }
    const double m = s1 / n;
    double var = s2 / n - m * m;
    if (var < 0.0) var = 0.0;
    *mean = (float)m;
    *rstd = (float)(1.0 / sqrt(var + DDIF_GN_EPS));
}
// apply: grid (chunks, B), NCHW
__global__ __launch_bounds__(256) void gn_apply_nchw_kernel(const float* x, const double* part, int nchunk, const float* gamma, const float* beta, const float* mask,
                                                            int C, int HW, int silu, float* y) {
    const int b = blockIdx.y;
    const size_t n = (size_t)C * HW;
    float mean, rstd;
    gn_stats_from_partials(part, b, nchunk, (double)n, &mean, &rstd);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i / HW);
        float v = fmaf((x[(size_t)b * n + i] - mean) * rstd, gamma[c], beta[c]);
        if (silu) v = dd_silu(v);
        if (mask) v *= mask[(size_t)b * n + i];
        y[(size_t)b * n + i] = v;
    }
}
// GroupNorm-alone backward, plane sums: grid (C, B): P[b][c] = {sum dy, sum dy x_hat} over the plane (fp64, fixed-order tree)
__global__ __launch_bounds__(256) void gn_bwd_plane_kernel(const float* x, const float* dy, const double* part, int nchunk, int C, int HW, double* P) {
    DDIF_DYN_SMEM(smem_);
    double* red = reinterpret_cast<double*>(smem_);  // [2][256]
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    float mean, rstd;
    gn_stats_from_partials(part, b, nchunk, (double)C * HW, &mean, &rstd);
    const size_t base = ((size_t)b * C + c) * HW;
    double p0 = 0.0, p1 = 0.0;
    for (int i = tid; i < HW; i += 256) {
        const float g = dy[base + i], xh = (x[base + i] - mean) * rstd;
        p0 += (double)g;
        p1 += (double)g * (double)xh;
    }
    red[tid] = p0;
    red[256 + tid] = p1;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (tid < st) {
            red[tid] += red[tid + st];
            red[256 + tid] += red[256 + tid + st];
        }
        __syncthreads();
    }
    if (tid == 0) {
        P[((size_t)b * C + c) * 2 + 0] = red[0];
        P[((size_t)b * C + c) * 2 + 1] = red[256];
    }
}
// per sample: M[b] = {mean, rstd, sum_c gamma_c P0 / N, sum_c gamma_c P1 / N} (the layout gn_bwd_dx_kernel / gn_bwd_affine_kernel read)
__global__ void gn_bwd_sample_finalize_kernel(const double* part, int nchunk, const float* gamma, int B, int C, int HW, double* ws) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double n = (double)C * HW;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < nchunk; ++k) {
        s1 += part[((size_t)b * nchunk + k) * 2 + 0];
        s2 += part[((size_t)b * nchunk + k) * 2 + 1];
    }
    const double mean = s1 / n;
    double var = s2 / n - mean * mean;
    if (var < 0.0) var = 0.0;
    double a1 = 0.0, a2 = 0.0;
    for (int c = 0; c < C; ++c) {
        a1 += (double)gamma[c] * ws[((size_t)b * C + c) * 2 + 0];
        a2 += (double)gamma[c] * ws[((size_t)b * C + c) * 2 + 1];
    }
    double* M = ws + (size_t)B * C * 2 + (size_t)b * 4;
    M[0] = mean;
    M[1] = 1.0 / sqrt(var + DDIF_GN_EPS);
    M[2] = a1 / n;
    M[3] = a2 / n;
}
// depthwise dW: grid (C, B) -> partial[b][c][9] (fp64 tree), then a fixed-order sum over samples
__global__ __launch_bounds__(256) void dwconv3x3_bwd_dw_partial_kernel(const float* x, const float* dy, int C, int H, int W, double* partial) {
    DDIF_DYN_SMEM(smem_);
    double* red = reinterpret_cast<double*>(smem_);  // [9][256]
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int HW = H * W;
    const float* gp = dy + ((size_t)b * C + c) * HW;
    const float* plane = x + ((size_t)b * C + c) * HW;
    for (int p = tid; p < HW; p += 256) {
        const int y = p / W, xx = p % W;
        const float g = gp[p];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = y + ky - 1, ix = xx + kx - 1;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) acc[ky * 3 + kx] += (double)g * (double)plane[iy * W + ix];
            }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) red[k * 256 + tid] = acc[k];
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (tid < st)
#pragma unroll
            for (int k = 0; k < 9; ++k) red[k * 256 + tid] += red[k * 256 + tid + st];
        __syncthreads();
    }
    if (tid < 9) partial[((size_t)b * C + c) * 9 + tid] = red[tid * 256];
}
__global__ void dwconv3x3_bwd_dw_reduce_kernel(const double* partial, int B, int C, float* dw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * 9) return;
    double s = 0.0;
    for (int b = 0; b < B; ++b) s += partial[(size_t)b * C * 9 + i];
    dw[i] = (float)s;
}
}  // namespace ddif
