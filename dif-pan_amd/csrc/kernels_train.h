// NHWC kernels of the native training step: the pieces of `loss.backward()` through UNetSR3
// (reference models/sr3_dwt.py:169-219 under .train(), diffusion_engine.py:230-233) that are not convolutions.  Every activation of the
// training program is [B, H, W, C] with C innermost -- the layout the conv kernels stage from -- so nothing is converted between ops
// (round 2's op-by-op graph spent 10 % of its time in NCHW <-> NHWC transposes around every call).
// All reductions: fixed order, no atomics -> bitwise reproducible gradients.
#pragma once
#include "ddif_dev.h"

namespace ddif {

// ---------------------------------------------------------------------------------------------------------------- FiLM (CondInjection, :395-396)
// out = xc * (1 + scale) + shift with film = [scale | shift] per pixel ([B, HW, 2C]); emits the GroupNorm partial of out.
// grid = (chunks, B), 256 threads, C % 4 == 0.
__global__ __launch_bounds__(256) void film_apply_kernel(const float* xc, const float* film, int HW, int C, float* out, double* st_out) {
    DDIF_DYN_SMEM(smem_);
    double* red = reinterpret_cast<double*>(smem_);  // [2][4]
    const int b = blockIdx.y;
    const size_t n4 = (size_t)HW * C / 4, base = (size_t)b * HW * C;
    const int c4n = C / 4;
    double s1 = 0.0, s2 = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const size_t pix = i / c4n;
        const int c = (int)(i - pix * c4n) * 4;
        const float4 v = *reinterpret_cast<const float4*>(xc + base + i * 4);
        const float* f = film + ((size_t)b * HW + pix) * 2 * C;
        const float4 sc = *reinterpret_cast<const float4*>(f + c);
        const float4 sh = *reinterpret_cast<const float4*>(f + C + c);
        float4 o;
        o.x = v.x * (1.f + sc.x) + sh.x;
        o.y = v.y * (1.f + sc.y) + sh.y;
        o.z = v.z * (1.f + sc.z) + sh.z;
        o.w = v.w * (1.f + sc.w) + sh.w;
        *reinterpret_cast<float4*>(out + base + i * 4) = o;
        s1 += ((double)o.x + o.y) + ((double)o.z + o.w);
        s2 += ((double)o.x * o.x + (double)o.y * o.y) + ((double)o.z * o.z + (double)o.w * o.w);
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 63) {
        red[wave] = s1;
        red[4 + wave] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0 && st_out) {
        const size_t pi = ((size_t)b * gridDim.x + blockIdx.x) * 2;
        st_out[pi + 0] = (red[0] + red[1]) + (red[2] + red[3]);
        st_out[pi + 1] = (red[4] + red[5]) + (red[6] + red[7]);
    }
}
// dxc = dout (1 + scale);  dfilm = [dout * xc | dout]
__global__ void film_bwd_nhwc_kernel(const float* xc, const float* film, const float* dout, size_t npix, int C, float* dxc, float* dfilm) {
    const size_t total = npix * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i / C;
        const int c = (int)(i - pix * C);
        const float g = dout[i];
        dxc[i] = g * (1.f + film[pix * 2 * C + c]);
        dfilm[pix * 2 * C + c] = g * xc[i];
        dfilm[pix * 2 * C + C + c] = g;
    }
}

// ---------------------------------------------------------------------------------------------------------------- channel concat / split / pad
__global__ void concat2_kernel(const float* a, int Ca, const float* b, int Cb, size_t npix, float* out) {
    const int C = Ca + Cb;
    const size_t total = npix * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i / C;
        const int c = (int)(i - pix * C);
        out[i] = c < Ca ? a[pix * Ca + c] : b[pix * Cb + (c - Ca)];
    }
}
// out_a = in[:, :Ca] (+ add_a), out_b = in[:, Ca:] (+ add_b): add_* nullable, same shapes as the outputs
__global__ void split2_kernel(const float* in, int Ca, int Cb, size_t npix, const float* add_a, const float* add_b, float* out_a, float* out_b) {
    const int C = Ca + Cb;
    const size_t total = npix * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i / C;
        const int c = (int)(i - pix * C);
        const float v = in[i];
        if (c < Ca) {
            const size_t o = pix * Ca + c;
            out_a[o] = add_a ? v + add_a[o] : v;
        } else {
            const size_t o = pix * Cb + (c - Ca);
            out_b[o] = add_b ? v + add_b[o] : v;
        }
    }
}
// out[pix][0..Cp) = in[pix][0..C) then zeros
__global__ void pad_channels_kernel(const float* in, int C, int Cp, size_t npix, float* out) {
    const size_t total = npix * Cp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i / Cp;
        const int c = (int)(i - pix * Cp);
        out[i] = c < C ? in[pix * C + c] : 0.f;
    }
}
// weight gradient computed over zero-padded input channels (Cout, Cp, taps) -> (Cout, C, taps)
__global__ void unpad_weight_kernel(const float* dwp, int Cout, int Cp, int C, int taps, float* dw) {
    const size_t total = (size_t)Cout * C * taps;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(i % taps);
        const int ci = (int)((i / taps) % C);
        const int co = (int)(i / ((size_t)taps * C));
        dw[i] = dwp[((size_t)co * Cp + ci) * taps + t];
    }
}
// out = a + b, or a * scale[b] (per-sample row scale; nullable -> 1) -- elementwise helpers of the reverse pass
__global__ void add2_kernel(const float* a, const float* b, size_t n, float* out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = a[i] + b[i];
}
// out[pix][c] = a[pix * lda + c] + b[pix * ldb + c]   (operands that are channel slices of wider tensors)
__global__ void add2_ld_kernel(const float* a, int lda, const float* b, int ldb, int C, size_t npix, float* out) {
    const size_t total = npix * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i / C;
        const int c = (int)(i - pix * C);
        out[i] = a[pix * lda + c] + b[pix * ldb + c];
    }
}
__global__ void scale_rows_kernel(const float* a, const float* scale, size_t per_sample, size_t total, float* out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) out[i] = a[i] * scale[i / per_sample];
}

// ---------------------------------------------------------------------------------------------------------------- resampling around strided / upsampled convs
// Downsample (conv3x3 stride 2): dX = stride-1 dgrad of dY with zeros inserted.  out (B, H, W, C) from dy (B, Ho, Wo, C)
__global__ void zero_stuff_nhwc_kernel(const float* dy, int B, int C, int Ho, int Wo, int H, int W, float* out) {
    const size_t total = (size_t)B * H * W * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int x = (int)((i / C) % W);
        const int y = (int)((i / ((size_t)C * W)) % H);
        const int b = (int)(i / ((size_t)C * W * H));
        float v = 0.f;
        if (!(y & 1) && !(x & 1) && (y >> 1) < Ho && (x >> 1) < Wo) v = dy[(((size_t)b * Ho + (y >> 1)) * Wo + (x >> 1)) * C + c];
        out[i] = v;
    }
}
// Upsample (nearest x2 then conv3x3): x_up (B, 2H, 2W, C) for the weight gradient; d(x) = 2x2 sum-pool of d(x_up)
__global__ void upsample2_nhwc_kernel(const float* x, int B, int C, int H, int W, float* out) {
    const size_t total = (size_t)B * 4 * H * W * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int xx = (int)((i / C) % (2 * W));
        const int yy = (int)((i / ((size_t)C * 2 * W)) % (2 * H));
        const int b = (int)(i / ((size_t)C * 4 * W * H));
        out[i] = x[(((size_t)b * H + (yy >> 1)) * W + (xx >> 1)) * C + c];
    }
}
__global__ void sumpool2_nhwc_kernel(const float* dxu, int B, int C, int H, int W, float* dx) {
    const size_t total = (size_t)B * H * W * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int x = (int)((i / C) % W);
        const int y = (int)((i / ((size_t)C * W)) % H);
        const int b = (int)(i / ((size_t)C * W * H));
        const float* p = dxu + (((size_t)b * 2 * H + 2 * y) * 2 * W + 2 * x) * C + c;
        dx[i] = (p[0] + p[C]) + (p[(size_t)2 * W * C] + p[(size_t)2 * W * C + C]);
    }
}

// ---------------------------------------------------------------------------------------------------------------- per-sample plane sums (time-bias gradient)
// out[b * ld + c] = sum over the HW pixels of dy[b, :, c]  -- d(loss)/d(FeatureWiseAffine bias row) of sample b (:257).  One workgroup per
// sample; threads = (pixel row r, channel c) side by side, combined through LDS in row order (as bias_grad_partial_kernel).
__global__ __launch_bounds__(256) void plane_sum_nhwc_kernel(const float* dy, int HW, int C, int ld, float* out) {
    DDIF_DYN_SMEM(smem_);
    float* red = reinterpret_cast<float*>(smem_);  // [256]
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* src = dy + (size_t)b * HW * C;
    for (int c0 = 0; c0 < C; c0 += 256) {
        const int cw = C - c0 < 256 ? C - c0 : 256, rows = 256 / cw;
        const int c = tid % cw, r = tid / cw;
        float s = 0.f;
        if (r < rows)
            for (int p = r; p < HW; p += rows) s += src[(size_t)p * C + c0 + c];
        red[tid] = s;
        __syncthreads();
        if (r == 0) {
            for (int rr = 1; rr < rows; ++rr) s += red[rr * cw + c];
            out[(size_t)b * ld + c0 + c] = s;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------- depthwise conv weight gradient (NHWC)
// dw[c][k] = sum_{b,y,x} x[b, y + ky - 1, x + kx - 1, c] dy[b, y, x, c]   (q.0 / kv.0, :507-520).  x may carry padding channels (ldx >= C).
// grid = (channel blocks of 32, nsplit row bands over B*H); partial [nsplit][C][9] fp64, then a fixed-order reduce.
__global__ __launch_bounds__(256) void dw_wgrad_partial_nhwc_kernel(const float* x, int ldx, const float* dy, int ldy, int B, int C, int H, int W, int nsplit, double* partial) {
    DDIF_DYN_SMEM(smem_);
    double* red = reinterpret_cast<double*>(smem_);  // [8][32][9]
    const int tid = threadIdx.x, cl = tid & 31, r = tid >> 5;  // 8 pixel lanes x 32 channels
    const int c = blockIdx.x * 32 + cl;
    const bool cok = c < C;
    const long long rows = (long long)B * H;
    const long long r0 = rows * blockIdx.y / nsplit, r1 = rows * (blockIdx.y + 1) / nsplit;
    double acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.0;
    // this thread's pixels, in order: (row, xx) with xx = r, r + 8, ... inside a row, then the next row.  FOUR pixels' ten loads each are issued
    // before any of them is used (a load-then-accumulate loop pays one memory latency per pixel); the accumulation order is unchanged.
    long long row = r < W ? r0 : r1;  // (pixel lanes beyond a narrow row have nothing to do)
    int xx = r;
    while (row < r1) {
        float g[4], xv[4][9];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ok[u] = cok && row < r1;
            if (ok[u]) {
                const int b = (int)(row / H), y = (int)(row % H);
                g[u] = dy[(((size_t)b * H + y) * W + xx) * ldy + c];
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const int iy = y + k / 3 - 1, ix = xx + k % 3 - 1;
                    xv[u][k] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[(((size_t)b * H + iy) * W + ix) * ldx + c] : 0.f;
                }
            }
            xx += 8;
            if (xx >= W) {
                xx = r;
                ++row;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (ok[u]) {
#pragma unroll
                for (int k = 0; k < 9; ++k) acc[k] += (double)g[u] * (double)xv[u][k];  // (a tap outside the image adds +0.0)
            }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) red[(r * 32 + cl) * 9 + k] = acc[k];
    __syncthreads();
    if (r == 0 && cok) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            double s = 0.0;
            for (int rr = 0; rr < 8; ++rr) s += red[(rr * 32 + cl) * 9 + k];
            partial[((size_t)blockIdx.y * C + c) * 9 + k] = s;
        }
    }
}
// 32 lanes per output element (8 outputs per 256-thread workgroup): lane l sums splits l, l + 32, ...; the 32 partials are added in lane order
__global__ __launch_bounds__(256) void dw_wgrad_reduce_kernel(const double* partial, int nsplit, int C, float* dw /* (C,1,3,3) */) {
    DDIF_DYN_SMEM(smem_);
    double* red = reinterpret_cast<double*>(smem_);  // [256]
    const int tid = threadIdx.x, l = tid & 31, o = blockIdx.x * 8 + (tid >> 5);
    double s = 0.0;
    if (o < C * 9)
        for (int k = l; k < nsplit; k += 32) s += partial[(size_t)k * C * 9 + o];
    red[tid] = s;
    __syncthreads();
    if (l == 0 && o < C * 9) {
        double t = 0.0;
        for (int k = 0; k < 32; ++k) t += red[(tid & ~31) + k];
        dw[o] = (float)t;
    }
}

// ---------------------------------------------------------------------------------------------------------------- SelfAttention core backward (NHWC)
// qkv [B, n, 3C] with channel = head * 3d + {q: 0..d, k: d..2d, v: 2d..3d} (the reference's view(B, heads, 3d, n) + chunk, :347-348), dout / do
// [B, n, C] with channel = head * d + c.  Same math and reduction order as selfattn_bwd_kernel (kernels_bwd_ops.h); only the addressing differs.
__global__ __launch_bounds__(256) void selfattn_bwd_nhwc_kernel(const float* qkv, const float* dout, int heads, int d, int n, float sc, float* dqkv) {
    DDIF_DYN_SMEM(smem_);
    const int n1 = n + 1;      // LDS row stride: with n = 64 a stride of n puts a whole column into ONE bank (64-way conflicts in the row softmax)
    float* qs = reinterpret_cast<float*>(smem_);
    float* ks = qs + d * n1;
    float* vs = ks + d * n1;
    float* gs = vs + d * n1;   // do
    float* as = gs + d * n1;   // [n][n1]
    float* ds = as + n * n1;   // [n][n1]
    float* rd = ds + n * n1;   // [n] row dots
    const int tid = threadIdx.x;
    const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
    const int C3 = 3 * d * heads, C1 = d * heads;
    const float* base = qkv + (size_t)b * n * C3 + hd * 3 * d;
    const float* gbase = dout + (size_t)b * n * C1 + hd * d;
    for (int i = tid; i < d * n; i += 256) {
        const int c = i % d, p = i / d;  // channel fastest: contiguous reads
        qs[c * n1 + p] = base[(size_t)p * C3 + c];
        ks[c * n1 + p] = base[(size_t)p * C3 + d + c];
        vs[c * n1 + p] = base[(size_t)p * C3 + 2 * d + c];
        gs[c * n1 + p] = gbase[(size_t)p * C1 + c];
    }
    __syncthreads();
    for (int i = tid; i < n * n; i += 256) {  // scores and d(a)
        const int p = i / n, q = i % n;
        float s = 0.f, da = 0.f;
        for (int c = 0; c < d; ++c) {
            s = fmaf(qs[c * n1 + p], ks[c * n1 + q], s);
            da = fmaf(gs[c * n1 + p], vs[c * n1 + q], da);
        }
        as[p * n1 + q] = s * sc;
        ds[p * n1 + q] = da;
    }
    __syncthreads();
    for (int p = tid; p < n; p += 256) {  // softmax of row p, then ds = a (da - sum a da)
        float mx = -3.0e38f;
        for (int q = 0; q < n; ++q) mx = fmaxf(mx, as[p * n1 + q]);
        float sum = 0.f;
        for (int q = 0; q < n; ++q) {
            const float e = dd_exp(as[p * n1 + q] - mx);
            as[p * n1 + q] = e;
            sum += e;
        }
        const float inv = 1.0f / sum;
        float dot = 0.f;
        for (int q = 0; q < n; ++q) {
            as[p * n1 + q] *= inv;
            dot = fmaf(as[p * n1 + q], ds[p * n1 + q], dot);
        }
        rd[p] = dot;
    }
    __syncthreads();
    float* dq = dqkv + (size_t)b * n * C3 + hd * 3 * d;
    for (int i = tid; i < d * n; i += 256) {  // dv[c][q] = sum_p a[p][q] do[c][p]
        const int c = i % d, q = i / d;
        float s = 0.f;
        for (int p = 0; p < n; ++p) s = fmaf(as[p * n1 + q], gs[c * n1 + p], s);
        dq[(size_t)q * C3 + 2 * d + c] = s;
    }
    __syncthreads();
    for (int i = tid; i < n * n; i += 256) {
        const int p = i / n, q = i % n;
        ds[p * n1 + q] = as[p * n1 + q] * (ds[p * n1 + q] - rd[p]);
    }
    __syncthreads();
    for (int i = tid; i < d * n; i += 256) {
        const int c = i % d, p = i / d;
        float s1 = 0.f, s2 = 0.f;
        for (int q = 0; q < n; ++q) {
            s1 = fmaf(ds[p * n1 + q], ks[c * n1 + q], s1);  // dq[c][p] = sc sum_q ds[p][q] k[c][q]
            s2 = fmaf(ds[q * n1 + p], qs[c * n1 + q], s2);  // dk[c][p] = sc sum_q ds[q][p] q[c][q]
        }
        dq[(size_t)p * C3 + c] = s1 * sc;
        dq[(size_t)p * C3 + d + c] = s2 * sc;
    }
}

// (the linear attention core of the decoder blocks, :545-566, lives in kernels_linattn.h)

// ---------------------------------------------------------------------------------------------------------------- misc
// F.l1_loss(pred, target), mean: per-workgroup fp64 partials, then a fixed-order sum (the single-workgroup form of kernels_bwd_ops.h costs
// 1.7 ms on a batch-32 output)
__global__ __launch_bounds__(256) void l1_partial_kernel(const float* pred, const float* target, size_t n, double* part) {
    DDIF_DYN_SMEM(smem_);
    double* red = reinterpret_cast<double*>(smem_);
    const int tid = threadIdx.x;
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n; i += (size_t)gridDim.x * 256) s += (double)fabsf(pred[i] - target[i]);
    red[tid] = s;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (tid < st) red[tid] += red[tid + st];
        __syncthreads();
    }
    if (tid == 0) part[blockIdx.x] = red[0];
}
__global__ void l1_final_kernel(const double* part, int nblk, size_t n, float* out) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double s = 0.0;
        for (int k = 0; k < nblk; ++k) s += part[k];
        out[0] = (float)(s / (double)n);
    }
}
// time MLP, top layer: dte[b][k] = sum_o dtb[b][o] wall[o][k]  (o over all 2272 FeatureWiseAffine outputs).  One workgroup per sample; thread
// (group g = tid / inner, k = tid % inner) sums the outputs o = g, g + G, ...; the G partials are added in group order
__global__ __launch_bounds__(256) void time_dte_kernel(const float* dtb, const float* wall, int ns, int inner, float* dte) {
    DDIF_DYN_SMEM(smem_);
    float* red = reinterpret_cast<float*>(smem_);  // [256]
    const int b = blockIdx.x, tid = threadIdx.x, G = 256 / inner, g = tid / inner, k = tid % inner;
    float s = 0.f;
    if (g < G)
        for (int o = g; o < ns; o += G) s = fmaf(dtb[(size_t)b * ns + o], wall[(size_t)o * inner + k], s);
    red[tid] = s;
    __syncthreads();
    if (tid < inner) {
        float t = 0.f;
        for (int gg = 0; gg < G; ++gg) t += red[gg * inner + tid];
        dte[(size_t)b * inner + tid] = t;
    }
}
// FeatureWiseAffine gradients: rows [off, off + n) of dwall ([nslots][inner]) / dball ([nslots]) copied into each block's own (n, inner) /
// (n) gradient tensors -- one launch over a table instead of two device copies per block (64 per iteration)
struct SlotScatter { float* w; float* b; int off, n; };
__global__ void slot_scatter_kernel(const SlotScatter* tab, int nslot_blocks, const float* dwall, const float* dball, int inner) {
    const int k = blockIdx.x;
    if (k >= nslot_blocks) return;
    const SlotScatter s = tab[k];
    for (int i = threadIdx.x; i < s.n * inner; i += blockDim.x) s.w[i] = dwall[(size_t)s.off * inner + i];
    for (int i = threadIdx.x; i < s.n; i += blockDim.x) s.b[i] = dball[s.off + i];
}
// per-sample plane sums in two stages: part[b][chunk][c] over pixel chunks, then out[b * ld + c] (fixed order)
__global__ __launch_bounds__(256) void plane_sum_partial_nhwc_kernel(const float* dy, int HW, int C, int nchunk, float* part) {
    DDIF_DYN_SMEM(smem_);
    float* red = reinterpret_cast<float*>(smem_);  // [256]
    const int b = blockIdx.y, ck = blockIdx.x, tid = threadIdx.x;
    const int per = (HW + nchunk - 1) / nchunk, p0 = ck * per, p1 = p0 + per < HW ? p0 + per : HW;
    const float* src = dy + (size_t)b * HW * C;
    for (int c0 = 0; c0 < C; c0 += 256) {
        const int cw = C - c0 < 256 ? C - c0 : 256, rows = 256 / cw;
        const int c = tid % cw, r = tid / cw;
        float s = 0.f;
        if (r < rows)
            for (int p = p0 + r; p < p1; p += rows) s += src[(size_t)p * C + c0 + c];
        red[tid] = s;
        __syncthreads();
        if (r == 0) {
            for (int rr = 1; rr < rows; ++rr) s += red[rr * cw + c];
            part[((size_t)b * nchunk + ck) * C + c0 + c] = s;
        }
        __syncthreads();
    }
}
__global__ void plane_sum_final_kernel(const float* part, int B, int C, int nchunk, int ld, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i % C;
    float s = 0.f;
    for (int k = 0; k < nchunk; ++k) s += part[((size_t)b * nchunk + k) * C + c];
    out[(size_t)b * ld + c] = s;
}
}  // namespace ddif
