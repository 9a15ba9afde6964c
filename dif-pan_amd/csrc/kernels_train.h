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
    for (long long row = r0; row < r1; ++row) {
        const int b = (int)(row / H), y = (int)(row % H);
        for (int xx = r; xx < W; xx += 8) {
            if (!cok) continue;
            const float g = dy[(((size_t)b * H + y) * W + xx) * ldy + c];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int iy = y + k / 3 - 1, ix = xx + k % 3 - 1;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) acc[k] += (double)g * (double)x[(((size_t)b * H + iy) * W + ix) * ldx + c];
            }
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
    float* qs = reinterpret_cast<float*>(smem_);
    float* ks = qs + d * n;
    float* vs = ks + d * n;
    float* gs = vs + d * n;   // do
    float* as = gs + d * n;   // [n][n]
    float* ds = as + n * n;   // [n][n]
    float* rd = ds + n * n;   // [n] row dots
    const int tid = threadIdx.x;
    const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
    const int C3 = 3 * d * heads, C1 = d * heads;
    const float* base = qkv + (size_t)b * n * C3 + hd * 3 * d;
    const float* gbase = dout + (size_t)b * n * C1 + hd * d;
    for (int i = tid; i < d * n; i += 256) {
        const int c = i % d, p = i / d;  // channel fastest: contiguous reads
        qs[c * n + p] = base[(size_t)p * C3 + c];
        ks[c * n + p] = base[(size_t)p * C3 + d + c];
        vs[c * n + p] = base[(size_t)p * C3 + 2 * d + c];
        gs[c * n + p] = gbase[(size_t)p * C1 + c];
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
    float* dq = dqkv + (size_t)b * n * C3 + hd * 3 * d;
    for (int i = tid; i < d * n; i += 256) {  // dv[c][q] = sum_p a[p][q] do[c][p]
        const int c = i % d, q = i / d;
        float s = 0.f;
        for (int p = 0; p < n; ++p) s = fmaf(as[p * n + q], gs[c * n + p], s);
        dq[(size_t)q * C3 + 2 * d + c] = s;
    }
    __syncthreads();
    for (int i = tid; i < n * n; i += 256) ds[i] = as[i] * (ds[i] - rd[i / n]);
    __syncthreads();
    for (int i = tid; i < d * n; i += 256) {
        const int c = i % d, p = i / d;
        float s1 = 0.f, s2 = 0.f;
        for (int q = 0; q < n; ++q) {
            s1 = fmaf(ds[p * n + q], ks[c * n + q], s1);  // dq[c][p] = sc sum_q ds[p][q] k[c][q]
            s2 = fmaf(ds[q * n + p], qs[c * n + q], s2);  // dk[c][p] = sc sum_q ds[q][p] q[c][q]
        }
        dq[(size_t)p * C3 + c] = s1 * sc;
        dq[(size_t)p * C3 + d + c] = s2 * sc;
    }
}

// ---------------------------------------------------------------------------------------------------------------- linear attention core (NHWC), :545-566
// q_pre [B, HW, qd], kv_pre [B, HW, 2 qd] (k | v), channel = head * d + i; out / dout [B, HW, ld] (the first qd channels of each pixel).
//   q = softmax over H of q_pre (per channel and column) / sqrt(d);  k = softmax over W of k_pre (per channel and row)
//   ctx[a][e] = sum_n k[a][n] v[e][n];   o[e][n] = sum_a ctx[a][e] q[a][n]
// One workgroup per (sample, head); W, H <= 64, d <= 32.  The image is walked in BANDS of R rows (R * W <= LA_BAND pixels staged in LDS per
// step) instead of row by row: 8 barriers instead of 128 at 64x64.  Every LDS cell has one owner thread and the bands / pixels are summed
// in index order: deterministic.  (Same math as linattn_fwd / _bwd_kernel of kernels_bwd_ops.h, the NCHW forms behind the stateless C-ABI ops.)
// max / sum(exp(. - max)) of a strided line of <= 64 elements: the whole line is loaded into registers first (independent loads in flight
// together -- a load-per-iteration loop pays one memory latency per element), then the two-pass arithmetic of torch.softmax
__device__ __forceinline__ void la_line_stats(const float* p0, size_t stride, int n, float* mx_out, float* sm_out) {
    float v[64];
#pragma unroll
    for (int r = 0; r < 64; ++r) v[r] = p0[(size_t)(r < n ? r : n - 1) * stride];
    float mx = -3.0e38f;
#pragma unroll
    for (int r = 0; r < 64; ++r) mx = fmaxf(mx, v[r]);  // the clamped tail repeats the last element
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 64; ++r)
        if (r < n) s += dd_exp(v[r] - mx);
    *mx_out = mx;
    *sm_out = s;
}

constexpr int LA_NT = 1024;   // threads per workgroup: one workgroup per (sample, head) is all the parallelism there is (256 at batch 32) -> 16 waves per CU
constexpr int LA_BAND = 4096;  // floats per staged band array: rows per band R = LA_BAND / (d * W)
__host__ __device__ inline int la_rows(int d, int H, int W) {
    int r = LA_BAND / (d * W);
    if (r < 1) r = 1;
    return r > H ? H : r;
}
inline size_t linattn_fwd_smem(int d, int H, int W) { return (size_t)(2 * d * W + d * d + 2 * d * la_rows(d, H, W) * W + 2 * d * H) * sizeof(float); }
inline size_t linattn_bwd_smem(int d, int H, int W) { return (size_t)(3 * d * W + 2 * d * d + 6 * d * la_rows(d, H, W) * W + d * la_rows(d, H, W) + 2 * d * H) * sizeof(float); }

__global__ __launch_bounds__(LA_NT) void linattn_fwd_nhwc_kernel(const float* q_pre, const float* kv_pre, int heads, int d, int H, int W, float sc, float* out, int ld_o) {
    DDIF_DYN_SMEM(smem_);
    const int R = la_rows(d, H, W), RW = R * W;
    float* qmx = reinterpret_cast<float*>(smem_);  // [d][W]
    float* qsm = qmx + d * W;                      // [d][W]
    float* ctx = qsm + d * W;                      // [d][d]
    float* rk = ctx + d * d;                       // [d][RW]
    float* rv = rk + d * RW;                       // [d][RW]
    float* kmx = rv + d * RW;                      // [d][H] row max of k_pre (softmax over W)
    float* ksm = kmx + d * H;                      // [d][H]
    const int tid = threadIdx.x;
    const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
    const int qd = heads * d, HW = H * W;
    const float* qb = q_pre + (size_t)b * HW * qd + hd * d;
    const float* kb = kv_pre + (size_t)b * HW * 2 * qd + hd * d;
    const float* vb = kb + qd;
    float* ob = out + (size_t)b * HW * ld_o + hd * d;
    for (int i = tid; i < d * W; i += LA_NT) {  // column statistics of q_pre (softmax over H)
        const int a = i % d, x = i / d;
        la_line_stats(qb + (size_t)x * qd + a, (size_t)W * qd, H, &qmx[a * W + x], &qsm[a * W + x]);
    }
    for (int i = tid; i < d * H; i += LA_NT) {  // row statistics of k_pre (softmax over W)
        const int a = i % d, y = i / d;
        la_line_stats(kb + (size_t)y * W * 2 * qd + a, (size_t)2 * qd, W, &kmx[a * H + y], &ksm[a * H + y]);
    }
    for (int i = tid; i < d * d; i += LA_NT) ctx[i] = 0.f;
    __syncthreads();
    for (int y0 = 0; y0 < H; y0 += R) {  // ctx: bands in order, one owner thread per (a, e)
        const int rows = H - y0 < R ? H - y0 : R, npx = rows * W;
        const int nit = d * npx;
        for (int i0 = tid; i0 < nit; i0 += LA_NT * 4) {  // four items' loads in flight per thread
            float kv_[4], vv_[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * LA_NT < nit ? i0 + u * LA_NT : tid;
                const size_t p = (size_t)y0 * W + i / d;
                kv_[u] = kb[p * 2 * qd + i % d];
                vv_[u] = vb[p * 2 * qd + i % d];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * LA_NT;
                if (i < nit) {
                    const int a = i % d, pl = i / d, y = y0 + pl / W;
                    rk[a * RW + pl] = dd_exp(kv_[u] - kmx[a * H + y]) / ksm[a * H + y];
                    rv[a * RW + pl] = vv_[u];
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < d * d; i += LA_NT) {  // (R W and every array offset are multiples of 4 floats: 16-byte LDS reads, 4 pixels per step)
            const int a = i / d, e = i % d;
            float s1 = ctx[i];
            const float4* pk = reinterpret_cast<const float4*>(rk + a * RW);
            const float4* pv = reinterpret_cast<const float4*>(rv + e * RW);
#pragma unroll 4
            for (int q4 = 0; q4 < npx / 4; ++q4) {
                const float4 k4 = pk[q4], v4 = pv[q4];
                s1 = fmaf(k4.x, v4.x, s1);
                s1 = fmaf(k4.y, v4.y, s1);
                s1 = fmaf(k4.z, v4.z, s1);
                s1 = fmaf(k4.w, v4.w, s1);
            }
            ctx[i] = s1;
        }
        __syncthreads();
    }
    // o[e][n] = sum_a ctx[a][e] q[a][n]: the band buffer is free now -- stage q_sm * sc of a band of pixels once (every (pixel, a) value is
    // needed by all d outputs of the pixel), then one thread per output element
    for (int y0 = 0; y0 < H; y0 += R) {
        const int rows = H - y0 < R ? H - y0 : R, npx = rows * W;
        const int nit = d * npx;
        for (int i0 = tid; i0 < nit; i0 += LA_NT * 4) {
            float qv_[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * LA_NT < nit ? i0 + u * LA_NT : tid;
                qv_[u] = qb[((size_t)y0 * W + i / d) * qd + i % d];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * LA_NT;
                if (i < nit) {
                    const int a = i % d, pl = i / d, x = pl % W;
                    rk[a * RW + pl] = dd_exp(qv_[u] - qmx[a * W + x]) / qsm[a * W + x] * sc;
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < d * npx; i += LA_NT) {
            const int e = i % d, pl = i / d;
            float s = 0.f;
            for (int a = 0; a < d; ++a) s = fmaf(ctx[a * d + e], rk[a * RW + pl], s);
            ob[((size_t)y0 * W + pl) * ld_o + e] = s;
        }
        __syncthreads();
    }
}

// Backward (do given):  dctx[a][e] = sum_n q[a][n] do[e][n];  dq = ctx do;  dk = dctx v;  dv = dctx^T k;  then the two softmax backwards.
// Pass 1: softmax statistics.  Pass 2 (bands): ctx, dctx.  Pass 3 (bands): dv (final), dk_pre (final: its softmax lives inside a row), dq (raw,
// parked in the output) and the column sums T[a][x] = sum_y dq q.  Pass 4: dq_pre = q_sm (dq - T).
__global__ __launch_bounds__(LA_NT) void linattn_bwd_nhwc_kernel(const float* q_pre, const float* kv_pre, const float* dout, int ld_g, int heads, int d, int H, int W, float sc,
                                                               float* dq_pre, float* dkv_pre) {
    DDIF_DYN_SMEM(smem_);
    const int R = la_rows(d, H, W), RW = R * W;
    float* qmx = reinterpret_cast<float*>(smem_);  // [d][W] column max of q_pre
    float* qsm = qmx + d * W;                      // [d][W] column sum of exp
    float* T = qsm + d * W;                        // [d][W] column sums of dq * q_sm
    float* ctx = T + d * W;                        // [d][d]
    float* dctx = ctx + d * d;                     // [d][d]
    float* rk = dctx + d * d;                      // [d][RW] k softmax of the band
    float* rv = rk + d * RW;                       // [d][RW]
    float* rq = rv + d * RW;                       // [d][RW] q softmax * sc
    float* rg = rq + d * RW;                       // [d][RW] do
    float* rdk = rg + d * RW;                      // [d][RW] dk
    float* rdq = rdk + d * RW;                     // [d][RW] dq (raw): the column sums T read it back from here, not from memory
    float* rdot = rdq + d * RW;                    // [d][R] row dots of the k softmax backward
    float* kmx = rdot + d * R;                     // [d][H] row max of k_pre (softmax over W)
    float* ksm = kmx + d * H;                      // [d][H] row sum of exp
    const int tid = threadIdx.x;
    const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
    const int qd = heads * d, HW = H * W;
    const float* qb = q_pre + (size_t)b * HW * qd + hd * d;
    const float* kb = kv_pre + (size_t)b * HW * 2 * qd + hd * d;
    const float* vb = kb + qd;
    const float* gb = dout + (size_t)b * HW * ld_g + hd * d;
    float* dqb = dq_pre + (size_t)b * HW * qd + hd * d;
    float* dkb = dkv_pre + (size_t)b * HW * 2 * qd + hd * d;
    float* dvb = dkb + qd;
    for (int i = tid; i < d * W; i += LA_NT) {  // pass 1: column statistics of q_pre (softmax over H)
        const int a = i % d, x = i / d;
        la_line_stats(qb + (size_t)x * qd + a, (size_t)W * qd, H, &qmx[a * W + x], &qsm[a * W + x]);
        T[a * W + x] = 0.f;
    }
    for (int i = tid; i < d * d; i += LA_NT) {
        ctx[i] = 0.f;
        dctx[i] = 0.f;
    }
    for (int i = tid; i < d * H; i += LA_NT) {  // row statistics of k_pre for every (channel, row)
        const int a = i % d, y = i / d;
        la_line_stats(kb + (size_t)y * W * 2 * qd + a, (size_t)2 * qd, W, &kmx[a * H + y], &ksm[a * H + y]);
    }
    __syncthreads();
    auto load_band = [&](int y0, int npx) {  // k softmax (over its row), v, q softmax * sc, do  -> LDS
        const int nit = d * npx;
        for (int i0 = tid; i0 < nit; i0 += LA_NT * 4) {  // four items' loads (16 in all) in flight per thread
            float kv_[4], vv_[4], qv_[4], gv_[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * LA_NT < nit ? i0 + u * LA_NT : tid;
                const size_t p = (size_t)y0 * W + i / d;
                const int a = i % d;
                kv_[u] = kb[p * 2 * qd + a];
                vv_[u] = vb[p * 2 * qd + a];
                qv_[u] = qb[p * qd + a];
                gv_[u] = gb[p * ld_g + a];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * LA_NT;
                if (i < nit) {
                    const int a = i % d, pl = i / d, y = y0 + pl / W, x = pl % W, l = a * RW + pl;
                    rk[l] = dd_exp(kv_[u] - kmx[a * H + y]) / ksm[a * H + y];
                    rv[l] = vv_[u];
                    rq[l] = dd_exp(qv_[u] - qmx[a * W + x]) / qsm[a * W + x] * sc;
                    rg[l] = gv_[u];
                }
            }
        }
        __syncthreads();
    };
    for (int y0 = 0; y0 < H; y0 += R) {  // pass 2: ctx and dctx
        const int rows = H - y0 < R ? H - y0 : R, npx = rows * W;
        load_band(y0, npx);
        for (int i = tid; i < d * d; i += LA_NT) {
            const int a = i / d, e = i % d;
            float s1 = ctx[i], s2 = dctx[i];
            const float4* pk = reinterpret_cast<const float4*>(rk + a * RW);
            const float4* pv = reinterpret_cast<const float4*>(rv + e * RW);
            const float4* pq = reinterpret_cast<const float4*>(rq + a * RW);
            const float4* pg = reinterpret_cast<const float4*>(rg + e * RW);
#pragma unroll 2
            for (int q4 = 0; q4 < npx / 4; ++q4) {  // 16-byte LDS reads, 4 pixels per step (same summation order)
                const float4 k4 = pk[q4], v4 = pv[q4], q4v = pq[q4], g4 = pg[q4];
                s1 = fmaf(k4.x, v4.x, s1);
                s2 = fmaf(q4v.x, g4.x, s2);
                s1 = fmaf(k4.y, v4.y, s1);
                s2 = fmaf(q4v.y, g4.y, s2);
                s1 = fmaf(k4.z, v4.z, s1);
                s2 = fmaf(q4v.z, g4.z, s2);
                s1 = fmaf(k4.w, v4.w, s1);
                s2 = fmaf(q4v.w, g4.w, s2);
            }
            ctx[i] = s1;
            dctx[i] = s2;
        }
        __syncthreads();
    }
    for (int y0 = 0; y0 < H; y0 += R) {  // pass 3
        const int rows = H - y0 < R ? H - y0 : R, npx = rows * W;
        load_band(y0, npx);
        for (int i = tid; i < d * npx; i += LA_NT) {
            const int a = i % d, pl = i / d, l = a * RW + pl;
            float dq = 0.f, dk = 0.f, dv = 0.f;
            for (int e = 0; e < d; ++e) {
                dq = fmaf(ctx[a * d + e], rg[e * RW + pl], dq);    // dq[a][n] = sum_e ctx[a][e] do[e][n]
                dk = fmaf(dctx[a * d + e], rv[e * RW + pl], dk);   // dk[a][n] = sum_e dctx[a][e] v[e][n]
                dv = fmaf(dctx[e * d + a], rk[e * RW + pl], dv);   // dv[a][n] = sum_e dctx[e][a] k[e][n]
            }
            const size_t p = (size_t)y0 * W + pl;
            dvb[p * 2 * qd + a] = dv;
            rdk[l] = dk;
            rdq[l] = dq * sc;
            dqb[p * qd + a] = dq * sc;  // d(q_sm) of o = ctx^T (q_sm * sc)
        }
        __syncthreads();
        for (int i = tid; i < d * rows; i += LA_NT) {  // row dots of the k softmax backward
            const int a = i % d, r = i / d;
            float s = 0.f;
            for (int x = 0; x < W; ++x) s = fmaf(rdk[a * RW + r * W + x], rk[a * RW + r * W + x], s);
            rdot[a * R + r] = s;
        }
        for (int i = tid; i < d * W; i += LA_NT) {  // column sums T += dq * q_sm over the band's rows, in row order (one owner per (a, x))
            const int a = i % d, x = i / d;
            float t = T[a * W + x];
            for (int r = 0; r < rows; ++r) {
                const int pl = r * W + x;
                t += rdq[a * RW + pl] * (rq[a * RW + pl] / sc);  // q_sm = rq / sc
            }
            T[a * W + x] = t;
        }
        __syncthreads();
        for (int i = tid; i < d * npx; i += LA_NT) {
            const int a = i % d, pl = i / d, l = a * RW + pl;
            dkb[((size_t)y0 * W + pl) * 2 * qd + a] = rk[l] * (rdk[l] - rdot[a * R + pl / W]);
        }
        __syncthreads();
    }
    const int NEL = d * HW;
    for (int i0 = tid; i0 < NEL; i0 += LA_NT * 4) {  // pass 4: softmax-over-H backward of q (four elements per thread in flight)
        float qv[4], gv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * LA_NT < NEL ? i0 + u * LA_NT : tid;
            const size_t el = (size_t)(i / d) * qd + i % d;
            qv[u] = qb[el];
            gv[u] = dqb[el];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * LA_NT;
            if (i < NEL) {
                const int a = i % d, p = i / d, x = p % W;
                const float qs_ = dd_exp(qv[u] - qmx[a * W + x]) / qsm[a * W + x];
                dqb[(size_t)p * qd + a] = qs_ * (gv[u] - T[a * W + x]);
            }
        }
    }
}

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
// flipped depthwise taps: w [9][C] (tap-major, the forward layout) -> [9][C] with tap k <- 8 - k: dX of a depthwise conv is the same
// depthwise conv of dY with the taps reversed
__global__ void flip_dw_taps_kernel(const float* w, int C, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 9 * C) return;
    const int k = i / C, c = i % C;
    out[i] = w[(8 - k) * C + c];
}

}  // namespace ddif
