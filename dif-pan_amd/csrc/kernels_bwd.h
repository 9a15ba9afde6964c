// Backward kernels of the 3x3 convolution (stride 1, pad 1, NHWC fp32) -- the first pieces of the training step
// (reference: loss.backward() through nn.Conv2d, diffusion_engine.py:233).
//
//   dgrad  dX = conv3x3(dY, W')  with W'[ci][co][ky][kx] = W[co][ci][2-ky][2-kx]: the FORWARD implicit-GEMM kernel
//          (kernels_conv.h, exact-fp32 MFMA instantiation) on weights re-packed on the device by
//          pack_dgrad_weights_kernel -- no new contraction kernel;
//   wgrad  dW[co][ci][tap] = sum over (b, y, x) of dY[b,y,x,co] * X[b,y+ky-1,x+kx-1,ci]: a GEMM with K = B*H*W.
//          conv3x3_wgrad_kernel: a workgroup owns a (32 co x 32 ci) block and walks row bands (sample, 4 rows); the band of
//          dY and the haloed band of X sit in LDS as fp32; the four waves split the band's pixel pairs and run, per pair,
//          one exact-fp32 v_mfma_f32_32x32x2_f32 per tap (A = dY^T fragment, B = shifted X fragment, 9 accumulators of
//          32x32) -- K split over workgroups (grid.y) AND waves, combined deterministically: waves through LDS in
//          fixed order, workgroups through a partials buffer reduced by wgrad_reduce_kernel in index order.  No atomics.
//   dbias  db[co] = sum dY: bias_grad_kernel (two-level fixed-order reduction).
#pragma once
#include "ddif_dev.h"

namespace ddif {

// packed fp32 fragment order of kernels_conv.h (pack_conv in ddif_net.cpp) for the dgrad weights, 16-channel chunks:
//   [n-block of 32 "couts" = ci][chunk of 16 "cins" = co][tap][k8][half h][j][4]   with cin' = chunk*16 + k8*8 + 4h + i
__global__ void pack_dgrad_weights_kernel(const float* w /* (Cout, Cin, 3, 3) */, int Cout, int Cin, int n_chunks, int nb_pad, float* out) {
    const size_t total = (size_t)nb_pad * n_chunks * 9 * 2 * 256;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        size_t r = idx;
        const int i = (int)(r % 4); r /= 4;
        const int j = (int)(r % 32); r /= 32;
        const int h = (int)(r % 2); r /= 2;
        const int k8 = (int)(r % 2); r /= 2;
        const int tap = (int)(r % 9); r /= 9;
        const int ch = (int)(r % n_chunks); r /= n_chunks;
        const int nbi = (int)r;
        const int co = ch * 16 + k8 * 8 + 4 * h + i;  // contraction index of the dgrad conv = output channel of the forward conv
        const int ci = nbi * 32 + j;                  // output channel of the dgrad conv = input channel of the forward conv
        float v = 0.f;
        if (co < Cout && ci < Cin) v = w[((size_t)co * Cin + ci) * 9 + (8 - tap)];  // flipped tap: (2-ky)*3 + (2-kx) = 8 - tap
        out[idx] = v;
    }
}

struct WgradArgs {
    const float* x;    // [B, H, W, Cin]  NHWC
    const float* dy;   // [B, H, W, Cout] NHWC
    int B, H, W, Cin, Cout;
    int n_ci;          // ci blocks of 32
    int rb;            // rows per band (4, 2 or 1: the largest whose tiles fit LDS)
    int bands_y;       // ceil(H / rb)
    float* partial;    // [gridDim.y][n_co * n_ci][9][32 co][32 ci]
    int centre_only;   // 1x1 convs riding this kernel: only tap 4 (the centre) is contracted and written
    int wshift;        // log2(W) when W is a power of two, else -1
    float* bpartial;   // nullable: [gridDim.y][n_co * 32] per-split column sums of dY (the BIAS gradient's partials), written by the ci-block-0 workgroups
};

// grid = (n_co * n_ci, NSPLIT); block 256.  LDS: dY band [rb*W][32] + X band [(rb+2)*(W+2)][32] + reduction scratch [4][1024].
// PF = 1 (bands of <= WG_PF * 256 float4 items, chosen by the host): the NEXT band is fetched into registers while the current one is in
// the MFMA loop -- a band's loads are then hidden instead of serialised in front of its MFMAs.  PF = 0: any band size, loads in batches
// of eight before the LDS writes.
// CENTRE = 1 (1x1 convs, only with PF): the X band is the SAME pixel set as the dY band -- no halo rows / columns are fetched or staged (the
// 3x3 form loaded three rows of X to use one: the 1x1 weight gradients ran at 3-9 % of their matrix time, profiles/r03/k_wgrad_by_shape.txt)
constexpr int WG_PF = 9, WG_PF_CENTRE = 13;  // prefetch registers (float4 per thread): 3x3 keeps 144 accumulators and must stay at 2 waves / SIMD
template <int PF, int CENTRE = 0>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_kernel(WgradArgs a) {
    dd_touch_kernargs<sizeof(WgradArgs)>();  // (ddif_dev.h: the argument block in one round trip)
    DDIF_DYN_SMEM(smem);
    constexpr int HALO = CENTRE ? 0 : 1;
    constexpr int NPF = CENTRE ? WG_PF_CENTRE : WG_PF;
    const int W = a.W, IW = W + 2 * HALO;
    const int RB = a.rb;
    float* Ys = reinterpret_cast<float*>(smem);  // [RB*W][32]
    float* Xs = Ys + RB * W * 32;                 // [(RB+2)*IW][32]  (CENTRE: [RB*W][32])
    float* Rs = Xs + (RB + 2 * HALO) * IW * 32;   // [4 waves][16 regs][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, j = lane & 31;
    const int wg_x = blockIdx.x, wg_y = blockIdx.y;  // (an XCD-aware renumbering -- the block pairs of one K split on one XCD, so that their re-reads of the
    // bands hit that L2 -- measured nothing: 41.8 / 20.1 us per launch against 42.0 / 19.7, profiles/r03/z9; removed)
    const int cob = wg_x / a.n_ci, cib = wg_x % a.n_ci;
    const int nbands = a.B * a.bands_y;
    const int NY = RB * W * 8, NX = (RB + 2 * HALO) * IW * 8, NTOT = NY + NX;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bacc = 0.f;  // bias gradient: this thread's (pixel lane tid / 32, channel tid % 32) column sum over the workgroup's bands

    // item i of a band (float4): i < NY -> dY row y0 + p / W, 32 couts of this block; else X rows y0 - 1 .. y0 + RB with a one-pixel zero border
    auto load_item = [&](int b, int y0, int i) -> float4 {
        const bool isy = i < NY;
        const int k = isy ? i : i - NY;
        const int c4 = k & 7, p = k >> 3;
        const int rowlen = isy ? W : IW;
        const int pr_ = p / rowlen, pc_ = p - pr_ * rowlen;
        const int y = isy ? y0 + pr_ : y0 - HALO + pr_, x = isy ? pc_ : pc_ - HALO;
        const int cc = (isy ? cob : cib) * 32 + c4 * 4, Cc = isy ? a.Cout : a.Cin;
        const bool ok = (i < NTOT) & (y >= 0) & (y < a.H) & (x >= 0) & (x < W) & (cc < Cc);
        const float* src = isy ? a.dy : a.x;
        const size_t off = ok ? (((size_t)b * a.H + y) * W + x) * Cc + cc : 0;
        const float4 ld = *reinterpret_cast<const float4*>(src + off);  // (offset 0 is always readable)
        return ok ? ld : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    [[maybe_unused]] float4 pf[PF ? NPF : 1];
    // PF: the geometry of this thread's items is the same for every band -- only (b, y0) move.  Precomputed once: element offset relative to
    // pixel (b, y0, 0) of the item's tensor, its row relative to y0, and whether it can ever be valid (column / channel inside).  Per band an
    // item then costs an add, two compares and a select instead of two integer divisions (the exact-fp32 MFMA shares the vector datapath:
    // this address arithmetic ran INSTEAD of MFMAs, ~1.3 us per one-row band against 1.9 us of matrix work)
    [[maybe_unused]] int it_off[PF ? NPF : 1], it_row[PF ? NPF : 1];
    [[maybe_unused]] unsigned it_ok = 0, it_isy = 0;
    if constexpr (PF) {
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int i = u * 256 + tid;
            const bool isy = i < NY;
            const int k = isy ? i : i - NY;
            const int c4 = k & 7, p = k >> 3;
            const int rowlen = isy ? W : IW;
            const int pr_ = p / rowlen, pc_ = p - pr_ * rowlen;
            const int x = isy ? pc_ : pc_ - HALO;
            const int cc = (isy ? cob : cib) * 32 + c4 * 4, Cc = isy ? a.Cout : a.Cin;
            const bool ok = (i < NTOT) & (x >= 0) & (x < W) & (cc < Cc);
            it_row[u] = isy ? pr_ : pr_ - HALO;
            it_off[u] = ok ? (it_row[u] * W + x) * Cc + cc : 0;
            it_ok |= (ok ? 1u : 0u) << u;
            it_isy |= (isy ? 1u : 0u) << u;
        }
    }
    auto fetch_band = [&](int b, int y0) {
        const size_t by = ((size_t)b * a.H + y0) * W * a.Cout, bx = ((size_t)b * a.H + y0) * W * a.Cin;
#pragma unroll
        for (int u = 0; u < (PF ? NPF : 1); ++u) {
            const int y = y0 + it_row[u];
            const bool isy = (it_isy >> u) & 1u;
            const bool ok = ((it_ok >> u) & 1u) & (y >= 0) & (y < a.H);
            const float* src = isy ? a.dy + by : a.x + bx;
            const float4 ld = *reinterpret_cast<const float4*>(ok ? src + it_off[u] : a.dy);  // (a.dy itself is always readable)
            pf[u] = ok ? ld : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if constexpr (PF) {
        if (wg_y < nbands) fetch_band(wg_y / a.bands_y, (wg_y % a.bands_y) * RB);
    }
    for (int band = wg_y; band < nbands; band += gridDim.y) {
        const int b = band / a.bands_y, y0 = (band % a.bands_y) * RB;
        __syncthreads();  // previous band fully consumed
        if constexpr (PF) {
#pragma unroll
            for (int u = 0; u < NPF; ++u) {
                const int i = u * 256 + tid;
                if (i < NTOT) *reinterpret_cast<float4*>(&Ys[i * 4]) = pf[u];  // Xs follows Ys: item i lives at float 4 i of the joint tile
            }
            __syncthreads();
            const int nb = band + gridDim.y;
            if (nb < nbands) fetch_band(nb / a.bands_y, (nb % a.bands_y) * RB);  // workgroup-uniform: the next band's loads fly during this band's MFMAs
        } else {
            for (int base = 0; base < NTOT; base += 256 * 8) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = load_item(b, y0, base + u * 256 + tid);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = base + u * 256 + tid;
                    if (i < NTOT) *reinterpret_cast<float4*>(&Ys[i * 4]) = v[u];
                }
            }
            __syncthreads();
        }
        const int npix = RB * W, npairs = (npix + 1) / 2;
        if (a.bpartial && cib == 0) {  // workgroup-uniform: bias gradient = column sums of the dY band already sitting in LDS (rows past the image are zero)
            const int c = tid & 31;
            float sb = 0.f;
            for (int p = tid >> 5; p < npix; p += 8) sb += Ys[p * 32 + c];
            bacc += sb;
        }
        // pixel pairs of the band: pair q -> pixels (2q, 2q+1) in row-major order of the RB x W band; wave w takes q = w, w+4, ...
        for (int q = wave; q < npairs; q += 4) {
            const int pr = 2 * q + h;         // this lane half's pixel (k index of the MFMA)
            const bool pv = pr < npix;        // odd pixel count: the last pair's second pixel does not exist
            const int p = pv ? pr : npix - 1;
            const int py = a.wshift >= 0 ? (p >> a.wshift) : p / W, px = p - py * W;  // W is a power of two at every level of the engine network
            const float av = pv ? Ys[p * 32 + j] : 0.f;  // A[i = co j][k = h]
            if constexpr (CENTRE) {
                acc[4] = DDIF_MFMA_32x32x2(av, pv ? Xs[p * 32 + j] : 0.f, acc[4]);
                continue;
            }
            if (a.centre_only) {
                acc[4] = DDIF_MFMA_32x32x2(av, Xs[((py + 1) * IW + px + 1) * 32 + j], acc[4]);
                continue;
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float bv = Xs[((py + t / 3) * IW + px + t % 3) * 32 + j];  // B[k = h][j = ci]: X at (y + ky - 1, x + kx - 1)
                acc[t] = DDIF_MFMA_32x32x2(av, bv, acc[t]);
            }
        }
    }
    if (a.bpartial && cib == 0) {  // the eight pixel lanes of every channel, in lane order
        __syncthreads();
        Rs[tid] = bacc;
        __syncthreads();
        if (tid < 32) {
            float sb = 0.f;
            for (int r = 0; r < 8; ++r) sb += Rs[r * 32 + tid];
            a.bpartial[(size_t)wg_y * (gridDim.x / a.n_ci) * 32 + cob * 32 + tid] = sb;
        }
    }
    // combine the four waves tap by tap in fixed order, write this workgroup's partial block
    float* outp = a.partial + ((size_t)wg_y * gridDim.x + wg_x) * 9 * 1024;
    for (int t = 0; t < 9; ++t) {
        if (a.centre_only && t != 4) continue;  // workgroup-uniform
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) Rs[(wave * 16 + r) * 64 + lane] = acc[t][r];
        __syncthreads();
        // 1024 outputs / 256 threads: element e = (r, lane): row (co) = (r&3) + 8*(r>>2) + 4*(lane>>5), col (ci) = lane & 31
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = tid + k * 256;
            const int r = e >> 6, l = e & 63;
            const float s = (Rs[(0 * 16 + r) * 64 + l] + Rs[(1 * 16 + r) * 64 + l]) + (Rs[(2 * 16 + r) * 64 + l] + Rs[(3 * 16 + r) * 64 + l]);
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
            outp[t * 1024 + row * 32 + col] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------------------------------
// The weight gradient on the 16-bit matrix pipe (round 5).  conv3x3_wgrad_kernel above contracts TWO pixels per v_mfma_f32_32x32x2_f32 (64 cycles, and the
// fp32 matrix op shares the vector datapath with the address / LDS instructions around it): 15-29 % of its own matrix bound at the 3x3 shapes, 3-9 % at the
// 1x1 ones (profiles/r03/k_wgrad_by_shape.txt).  Here K = 16 pixels per v_mfma_f32_32x32x16_bf16 with BOTH operands split three ways (bf16x3, six exact
// products, small terms first, fp32 accumulate -- the arithmetic of the forward / dgrad convs of the training step): 192 instead of 512 matrix cycles per
// 16 pixels and tap, on a pipe of its own.
//   * the contraction index is the PIXEL, so the operands need pixels contiguous per channel: rows are staged TRANSPOSED, [plane][channel][pixel] as bf16,
//     by items of (8 consecutive pixels x 4 channels): eight float4 loads, split, twelve 16-byte LDS writes.  Channel stride = an odd number of 16-byte
//     slots (fragment reads, lane = channel, conflict-free); items are numbered pixel-group-fastest (the eight lanes of a write phase hit consecutive slots);
//   * a 16-pixel group = two 8-pixel segments (lane half h = segment); the A fragment of lane (j, h) is dY^T[co = j][8 pixels], the B fragment of tap
//     (ky, kx) is X[ci = j][the same 8 pixels shifted by (ky - 1, kx - 1)]: rows shift by whole rows of the staged tile, columns by ONE bf16 = 2 bytes --
//     the aligned chunk plus one dword of each neighbour chunk, funnel-shifted (VALU work beside the matrix pipe).  The X rows carry 8 zero pixels on either
//     side (written once per launch), so the neighbours of the first / last segment exist;
//   * a workgroup owns a (32 co x 32 ci) block and a CONTIGUOUS run of bands (sample, RB rows) of the K split; the X rows live in a ring of RB + 2 row slots
//     (row y in slot (y + 1) mod (RB + 2)), so a band stages its RB rows of dY and only the RB NEW rows of X -- the two rows it shares with the previous band
//     stay where they are (a band-per-item form staged RB + 2 rows of X per RB rows of dY: three for one at 64 x 64).  At the start of a run and of every
//     sample the two leading rows (y0 - 1, y0) are staged by a step of their own;
//   * the next band's loads fly during the current band's MFMAs, four waves split the pixel groups and meet in LDS in fixed order, partial blocks
//     [split][block][tap][32][32] reduced by wgrad_reduce_kernel in index order.  No atomics; bit-reproducible.  Needs 8 | W and W <= 128 (the host falls back
//     to the fp32 kernel otherwise).
struct WgradX3Geom {
    int rb, xw, ys, xs;  // rows per band; staged X row length (W + 16, or W for the halo-free 1x1 form); channel strides of the two tiles in bf16 elements
};
// ABL (tools/mbench_wgrad.cpp only): 1 = no global loads, 2 = no MFMAs, 4 = no staging (split + LDS writes), 8 = no partial stores, 16 = no fragment reads
template <int CENTRE, int ABL = 0>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_x3_kernel(WgradArgs a, WgradX3Geom gm) {
    dd_touch_kernargs<sizeof(WgradArgs) + sizeof(WgradX3Geom)>();
    DDIF_DYN_SMEM(smem);
    constexpr int HALO = CENTRE ? 0 : 1, LPAD = CENTRE ? 0 : 8;
    const int W = a.W, RB = gm.rb, XR = RB + 2 * HALO, XW = gm.xw, YS = gm.ys, XS = gm.xs, SEGS = W >> 3;
    unsigned short* Yt = reinterpret_cast<unsigned short*>(smem);  // [3 planes][32 co][YS]      pixel p = r * W + x of the band
    unsigned short* Xt = Yt + 3 * 32 * YS;                         // [3 planes][32 ci][XS]      row slot s, pixel x at s * XW + LPAD + x
    float* Rs = reinterpret_cast<float*>(smem);                    // epilogue: [4 waves][16 regs][64 lanes], aliases the tiles (barrier in between)
    const int tid = threadIdx.x, lane = tid & 63;
#ifdef DDIF_EMU
    const int wave = tid >> 6;
#else
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const int h = lane >> 5, j = lane & 31;
    const int wg_x = blockIdx.x, wg_y = blockIdx.y;
    const int cob = wg_x / a.n_ci, cib = wg_x % a.n_ci;
    const int nbands = a.B * a.bands_y;
    const int band0 = (int)((long long)nbands * wg_y / gridDim.y), band1 = (int)((long long)nbands * (wg_y + 1) / gridDim.y);  // this split's run of bands
    const int npg = RB * SEGS, NYI = npg * 8, NIT = 2 * NYI;  // items of a band: (row, segment, channel quad) of dY, then of the new X rows (host: NIT <= 256)
    f32x16 acc[CENTRE ? 1 : 9];
#pragma unroll
    for (int t = 0; t < (CENTRE ? 1 : 9); ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bacc[4] = {0.f, 0.f, 0.f, 0.f};  // bias gradient: this thread's dY item summed over its 8 pixels, per channel of the quad, over the bands
    const bool want_bias = a.bpartial && cib == 0;

    if constexpr (!CENTRE) {  // the zero borders of the X row slots: 8 pixels left and right of every (plane, channel, slot)
        for (int i = tid; i < 3 * 32 * XR * 2; i += 256) {
            const int side = i & 1, r = (i >> 1) % XR, pc = (i >> 1) / XR;
            *reinterpret_cast<uint4*>(&Xt[(size_t)pc * XS + r * XW + (side ? LPAD + W : 0)]) = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    // this thread's item (the same for every band): tensor, row within the band, element offset relative to pixel (b, y0, 0), LDS element (row slot 0 for X)
    const bool it_in = tid < NIT, it_isy = tid < NYI;
    int it_r, it_off, it_dst;
    bool it_cok;
    {
        const int k = it_isy ? tid : tid - NYI;
        const int c4 = (k / npg) & 7, pg = k % npg;  // pixel group fastest, channel quad slowest
        const int sg = pg % SEGS, r = pg / SEGS;
        const int cc = (it_isy ? cob : cib) * 32 + c4 * 4, Cc = it_isy ? a.Cout : a.Cin;
        it_cok = it_in & (cc < Cc);
        it_r = r;
        it_off = it_cok ? sg * 8 * Cc + cc : 0;  // + row * W * Cc per band
        it_dst = it_isy ? (c4 * 4) * YS + r * W + sg * 8 : 3 * 32 * YS + (c4 * 4) * XS + LPAD + sg * 8;
    }
    // the two leading rows of a run / of a sample (X rows y0 - 1 and y0): items (row of the pair, segment, channel quad)
    const bool ld_in = !CENTRE && tid < 2 * SEGS * 8;
    int ld_r = 0, ld_off = 0, ld_dst = 0;
    bool ld_cok = false;
    if constexpr (!CENTRE) {
        const int c4 = (tid / (2 * SEGS)) & 7, pg = tid % (2 * SEGS);
        const int sg = pg % SEGS, cc = cib * 32 + c4 * 4;
        ld_r = pg / SEGS;
        ld_cok = ld_in & (cc < a.Cin);
        ld_off = ld_cok ? sg * 8 * a.Cin + cc : 0;
        ld_dst = 3 * 32 * YS + (c4 * 4) * XS + LPAD + sg * 8;
    }
    auto load8 = [&](const float* src, int ps, bool ok, float4* v) {  // eight consecutive pixels of one channel quad
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (ABL & 1) {
                v[e] = make_float4(0.25f, -0.5f, 0.125f, (float)e);
                continue;
            }
            const float4 ld = *reinterpret_cast<const float4*>(src + (size_t)e * ps);
            v[e] = ok ? ld : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto split8 = [&](const float4* v, unsigned short* d, int cs, int ps) {  // -> three planes of [4 channels][8 pixels]
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            unsigned hh[4], mm[4], ll[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) dd_split3_pair((&v[2 * q].x)[c], (&v[2 * q + 1].x)[c], &hh[q], &mm[q], &ll[q]);
            *reinterpret_cast<uint4*>(d + c * cs) = make_uint4(hh[0], hh[1], hh[2], hh[3]);
            *reinterpret_cast<uint4*>(d + c * cs + ps) = make_uint4(mm[0], mm[1], mm[2], mm[3]);
            *reinterpret_cast<uint4*>(d + c * cs + 2 * ps) = make_uint4(ll[0], ll[1], ll[2], ll[3]);
        }
    };
    auto slot_of = [&](int s) { return s >= XR ? s - XR : s; };  // (s < 2 XR)
    float4 pf[8];
    auto fetch_band = [&](int b, int y0) {  // dY rows y0 .. y0 + RB - 1 and the NEW X rows (3x3: y0 + 1 .. y0 + RB; 1x1: the same rows as dY)
        if (!it_in) return;
        const int y = y0 + it_r + (it_isy ? 0 : HALO);
        const bool ok = it_cok & (y < a.H);
        const int Cc = it_isy ? a.Cout : a.Cin;
        const float* src = ok ? (it_isy ? a.dy : a.x) + ((size_t)b * a.H + y) * W * Cc + it_off : a.dy;  // (a.dy itself is always readable)
        load8(src, ok ? Cc : 0, ok, pf);
    };
    auto stage_band = [&](int y0) {
        if (!it_in) return;
        if (ABL & 4) {
            bacc[0] += pf[0].x + pf[7].w;
            return;
        }
        const int cs = it_isy ? YS : XS;
        const int slot = (it_isy || CENTRE) ? 0 : slot_of((y0 + 2) % XR + it_r);  // X row y0 + 1 + r lives in slot (y0 + 2 + r) mod XR
        split8(pf, Yt + it_dst + ((CENTRE && !it_isy) ? it_r * XW : slot * XW), cs, 32 * cs);
        if (it_isy) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                bacc[c] += (((&pf[0].x)[c] + (&pf[1].x)[c]) + ((&pf[2].x)[c] + (&pf[3].x)[c])) + (((&pf[4].x)[c] + (&pf[5].x)[c]) + ((&pf[6].x)[c] + (&pf[7].x)[c]));
        }
    };
    auto stage_lead = [&](int b, int y0) {  // X rows y0 - 1 and y0 (row y0 - 1 of the first band of a sample is the zero border: masked)
        if (!ld_in) return;
        const int y = y0 - 1 + ld_r;
        const bool ok = ld_cok & (y >= 0) & (y < a.H);
        const float* src = ok ? a.x + ((size_t)b * a.H + y) * W * a.Cin + ld_off : a.x;
        float4 v[8];
        load8(src, ok ? a.Cin : 0, ok, v);
        if (ABL & 4) {
            bacc[1] += v[0].x + v[7].w;
            return;
        }
        split8(v, Yt + ld_dst + slot_of(y0 % XR + ld_r) * XW, XS, 32 * XS);
    };
    auto frag = [&](const unsigned short* p) -> float4 {
        if (ABL & 16) return make_float4(1e-3f * (float)lane, 2e-3f, 3e-3f, 4e-3f);
        return *reinterpret_cast<const float4*>(p);
    };
    auto x3 = [&](f32x16 c, const float4* A, const float4* Bv) -> f32x16 {  // planes: 0 hi, 1 mid, 2 lo
        if (ABL & 2) {
            c[0] += A[0].x * Bv[0].x + A[1].y * Bv[1].y + A[2].z * Bv[2].z + A[0].w * Bv[2].w;
            return c;
        }
        c = DDIF_MFMA_32x32x16_BF16(A[2], Bv[0], c);
        c = DDIF_MFMA_32x32x16_BF16(A[0], Bv[2], c);
        c = DDIF_MFMA_32x32x16_BF16(A[1], Bv[1], c);
        c = DDIF_MFMA_32x32x16_BF16(A[1], Bv[0], c);
        c = DDIF_MFMA_32x32x16_BF16(A[0], Bv[1], c);
        c = DDIF_MFMA_32x32x16_BF16(A[0], Bv[0], c);
        return c;
    };
    auto funnel = [&](unsigned hi, unsigned lo) -> unsigned {  // ({hi, lo} >> 16)[31:0]
#ifdef DDIF_EMU
        return (lo >> 16) | (hi << 16);
#else
        return __builtin_amdgcn_alignbit(hi, lo, 16);
#endif
    };

    if (band0 < band1) fetch_band(band0 / a.bands_y, (band0 % a.bands_y) * RB);
    const int ngroups = (npg + 1) >> 1;  // 16-pixel groups of a band (npg half-groups of 8 pixels)
    for (int band = band0; band < band1; ++band) {
        const int b = band / a.bands_y, by = band - b * a.bands_y, y0 = by * RB;
        __syncthreads();  // previous band fully consumed (first band: the zero borders are written)
        if (!CENTRE && (band == band0 || by == 0)) stage_lead(b, y0);  // (workgroup-uniform)
        stage_band(y0);
        __syncthreads();
        if (band + 1 < band1) fetch_band((band + 1) / a.bands_y, ((band + 1) % a.bands_y) * RB);  // in flight during this band's MFMAs
        const int sb = CENTRE ? 0 : y0 % XR;  // X row y0 - 1 lives in slot y0 mod XR
        for (int g = wave; g < ngroups; g += 4) {
            const int hg = 2 * g + h;
            const bool pv = hg < npg;  // odd number of segments in the band: the last group's second half does not exist
            const int hgc = pv ? hg : npg - 1;
            const int r = hgc / SEGS, sg = hgc - r * SEGS;
            float4 A[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                A[pl] = frag(Yt + (size_t)(pl * 32 + j) * YS + r * W + sg * 8);
                if (!pv) A[pl] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if constexpr (CENTRE) {
                float4 Bv[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) Bv[pl] = frag(Xt + (size_t)(pl * 32 + j) * XS + r * XW + sg * 8);
                acc[0] = x3(acc[0], A, Bv);
            } else {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    float4 B0[3], B1[3], B2[3];
                    const int slot = slot_of(sb + r + ky);  // X row y0 + r + ky - 1
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const unsigned short* row = Xt + (size_t)(pl * 32 + j) * XS + slot * XW + LPAD + sg * 8;
                        const uint4 cur = (ABL & 16) ? make_uint4(lane, 2u, 3u, (unsigned)ky) : *reinterpret_cast<const uint4*>(row);
                        const unsigned prev = (ABL & 16) ? 5u : *reinterpret_cast<const unsigned*>(row - 2), next = (ABL & 16) ? 7u : *reinterpret_cast<const unsigned*>(row + 8);
                        B1[pl] = __builtin_bit_cast(float4, cur);
                        B0[pl] = __builtin_bit_cast(float4, make_uint4(funnel(cur.x, prev), funnel(cur.y, cur.x), funnel(cur.z, cur.y), funnel(cur.w, cur.z)));   // x - 1
                        B2[pl] = __builtin_bit_cast(float4, make_uint4(funnel(cur.y, cur.x), funnel(cur.z, cur.y), funnel(cur.w, cur.z), funnel(next, cur.w)));   // x + 1
                    }
                    acc[3 * ky + 0] = x3(acc[3 * ky + 0], A, B0);
                    acc[3 * ky + 1] = x3(acc[3 * ky + 1], A, B1);
                    acc[3 * ky + 2] = x3(acc[3 * ky + 2], A, B2);
                }
            }
        }
    }
    __syncthreads();  // the tiles are dead: their LDS becomes the reduction scratch
    if (want_bias) {  // the dY items of one channel quad are consecutive item indices: summed in index order
        float* Bs = Rs;  // [NYI][4]
        if (tid < NYI) *reinterpret_cast<float4*>(&Bs[tid * 4]) = make_float4(bacc[0], bacc[1], bacc[2], bacc[3]);
        __syncthreads();
        if (tid < 32) {
            const int c4 = tid >> 2, c = tid & 3;
            float sb2 = 0.f;
            for (int t = c4 * npg; t < (c4 + 1) * npg; ++t) sb2 += Bs[t * 4 + c];
            a.bpartial[(size_t)wg_y * (gridDim.x / a.n_ci) * 32 + cob * 32 + tid] = sb2;
        }
        __syncthreads();
    }
    float* outp = a.partial + ((size_t)wg_y * gridDim.x + wg_x) * 9 * 1024;
#pragma unroll
    for (int t = 0; t < (CENTRE ? 1 : 9); ++t) {
        if (t > 0) __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) Rs[(wave * 16 + r) * 64 + lane] = acc[t][r];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = tid + k * 256;
            const int r = e >> 6, l = e & 63;
            const float s = (Rs[(0 * 16 + r) * 64 + l] + Rs[(1 * 16 + r) * 64 + l]) + (Rs[(2 * 16 + r) * 64 + l] + Rs[(3 * 16 + r) * 64 + l]);
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
            if (!(ABL & 8) || s == 12345.678f) outp[(CENTRE ? 4 : t) * 1024 + row * 32 + col] = s;
        }
    }
}

// Fixed-order sum over the K splits, shared by the eight 32-thread SLICES of a 256-thread workgroup: thread (slice sl = tid >> 5, lane c = tid & 31)
// adds the splits sl, sl + 8, ... of element c with eight loads in flight (a load-per-iteration loop pays one memory latency per split; one
// thread per element pays nsplit / 8 of them -- 64 at 512 splits -- on a grid of a few dozen workgroups), the slices are combined through LDS
// in slice order.  Every thread of the workgroup must call it (two barriers); slice 0 returns the sum.
__device__ __forceinline__ float wgrad_sum_splits8(const float* p, bool valid, size_t stride, int nsplit, float* red /* [256] */) {
    const int tid = threadIdx.x, sl = tid >> 5;
    float acc = 0.f;
    if (valid)
        for (int k0 = sl; k0 < nsplit; k0 += 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = k0 + 8 * u < nsplit ? p[(size_t)(k0 + 8 * u) * stride] : 0.f;
            acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
    red[tid] = acc;
    __syncthreads();
    float tot = 0.f;
    if (sl == 0) {
        const int c = tid;
        tot = ((red[c] + red[c + 32]) + (red[c + 64] + red[c + 96])) + ((red[c + 128] + red[c + 160]) + (red[c + 192] + red[c + 224]));
    }
    __syncthreads();
    return tot;
}
// (both reduce kernels also finish the bias gradient when bpartial / db are given: db[c] = fixed-order sum over the splits)
// (the launch carries ONE extra workgroup -- the last -- for the bias; returns true for that workgroup)
__device__ __forceinline__ bool wgrad_reduce_bias(const float* bpartial, int nsplit, int n_co, int Cout, float* db, float* red) {
    if (!bpartial || !db || blockIdx.x != gridDim.x - 1) return false;
    for (int c0 = 0; c0 < Cout; c0 += 32) {
        const int c = c0 + (threadIdx.x & 31);
        const float v = wgrad_sum_splits8(bpartial + c, c < Cout, (size_t)n_co * 32, nsplit, red);
        if (threadIdx.x < 32 && c < Cout) db[c] = v;
    }
    return true;
}
// 1x1 form: dW (Cout, Cin) from the centre-tap blocks only.  A workgroup takes rows of 32 input channels of the PARTIAL layout
// ([block][tap][co % 32][ci % 32]) at a time.
__global__ __launch_bounds__(256) void wgrad_reduce_centre_kernel(const float* partial, int nsplit, int nblk, int n_ci, int Cout, int Cin, float* dw, const float* bpartial,
                                                                  float* db) {
    DDIF_DYN_SMEM(smem_);
    float* red = reinterpret_cast<float*>(smem_);  // [256]
    if (wgrad_reduce_bias(bpartial, nsplit, nblk / n_ci, Cout, db, red)) return;
    const int nb_main = (bpartial && db) ? gridDim.x - 1 : gridDim.x;
    const int nrows = nblk * 32;  // (block, co % 32)
    for (int row = blockIdx.x; row < nrows; row += nb_main) {
        const int blk = row >> 5, r = row & 31, c = threadIdx.x & 31;
        const int co = (blk / n_ci) * 32 + r, ci = (blk % n_ci) * 32 + c;
        const bool ok = co < Cout && ci < Cin;
        const float v = wgrad_sum_splits8(partial + ((size_t)blk * 9 + 4) * 1024 + (size_t)r * 32 + c, ok, (size_t)nblk * 9 * 1024, nsplit, red);
        if (threadIdx.x < 32 && ok) dw[(size_t)co * Cin + ci] = v;
    }
}
// dW (OIHW) = fixed-order sum of the partial blocks, the same way: a workgroup per row (block, tap, co % 32) of 32 input channels
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* partial, int nsplit, int nblk, int n_ci, int Cout, int Cin, float* dw, const float* bpartial, float* db) {
    DDIF_DYN_SMEM(smem_);
    float* red = reinterpret_cast<float*>(smem_);  // [256]
    if (wgrad_reduce_bias(bpartial, nsplit, nblk / n_ci, Cout, db, red)) return;
    const int nb_main = (bpartial && db) ? gridDim.x - 1 : gridDim.x;
    const int nrows = nblk * 9 * 32;
    const size_t total = (size_t)nrows * 32;
    for (int row = blockIdx.x; row < nrows; row += nb_main) {
        const int c = threadIdx.x & 31, r = row & 31;
        const int t = (row >> 5) % 9, blk = row / (9 * 32);
        const int co = (blk / n_ci) * 32 + r, ci = (blk % n_ci) * 32 + c;
        const bool ok = co < Cout && ci < Cin;
        const float v = wgrad_sum_splits8(partial + (size_t)row * 32 + c, ok, total, nsplit, red);
        if (threadIdx.x < 32 && ok) dw[((size_t)co * Cin + ci) * 9 + t] = v;
    }
}

// db[co] = sum over pixels of dY[., co]  (NHWC): grid.x = chunks of pixels -> partial [chunk][Cout]; then fixed-order sum.
// thread = (pixel row r, channel c): rows = 256 / Cout threads walk the chunk's pixels side by side, combined through LDS in row order
__global__ __launch_bounds__(256) void bias_grad_partial_kernel(const float* dy, size_t npix, int Cout, int nchunk, float* partial) {
    DDIF_DYN_SMEM(smem_);
    float* red = reinterpret_cast<float*>(smem_);  // [256]
    const int tid = threadIdx.x;
    const size_t per = (npix + nchunk - 1) / nchunk;
    const size_t p0 = (size_t)blockIdx.x * per, p1 = p0 + per < npix ? p0 + per : npix;
    for (int c0 = 0; c0 < Cout; c0 += 256) {
        const int cw = Cout - c0 < 256 ? Cout - c0 : 256, rows = 256 / cw;
        const int c = tid % cw, r = tid / cw;
        float s = 0.f;
        if (r < rows)
            for (size_t p = p0 + r; p < p1; p += rows) s += dy[p * Cout + c0 + c];
        red[tid] = s;
        __syncthreads();
        if (r == 0) {
            for (int rr = 1; rr < rows; ++rr) s += red[rr * cw + c];
            partial[(size_t)blockIdx.x * Cout + c0 + c] = s;
        }
        __syncthreads();
    }
}
// one 64-thread workgroup per channel: thread-strided partial sums, then a fixed-order tree
__global__ __launch_bounds__(64) void bias_grad_reduce_kernel(const float* partial, int nchunk, int Cout, float* db) {
    DDIF_DYN_SMEM(smem_);
    float* red = reinterpret_cast<float*>(smem_);  // [64]
    const int c = blockIdx.x, tid = threadIdx.x;
    float s = 0.f;
    for (int k = tid; k < nchunk; k += 64) s += partial[(size_t)k * Cout + c];
    red[tid] = s;
    __syncthreads();
    for (int st = 32; st >= 1; st >>= 1) {
        if (tid < st) red[tid] += red[tid + st];
        __syncthreads();
    }
    if (tid == 0) db[c] = red[0];
}

}  // namespace ddif

namespace ddif {
// layout conversion at this op's NCHW boundary (the reference's layout): a batched matrix transpose out[b][s][r] = in[b][r][s] through a
// 32 x 33 LDS tile, so that both the reads (along s) and the writes (along r) are coalesced.  256 threads = 32 x 8; dynamic LDS = TR_SMEM.
constexpr size_t TR_SMEM = 32 * 33 * sizeof(float);
__device__ __forceinline__ void transpose_batched(const float* in, int B, int R, int S, float* out) {
    DDIF_DYN_SMEM(smem_);
    float* tile = reinterpret_cast<float*>(smem_);  // [32][33]
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int tr = (R + 31) / 32, ts = (S + 31) / 32;
    const long long ntiles = (long long)B * tr * ts;
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int b = (int)(t / (tr * ts)), rem = (int)(t % (tr * ts));
        const int r0 = (rem / ts) * 32, s0 = (rem % ts) * 32;
        const float* ib = in + (size_t)b * R * S;
        float* ob = out + (size_t)b * R * S;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + ty + 8 * k, sidx = s0 + tx;
            if (r < R && sidx < S) tile[(ty + 8 * k) * 33 + tx] = ib[(size_t)r * S + sidx];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int sidx = s0 + ty + 8 * k, r = r0 + tx;
            if (r < R && sidx < S) ob[(size_t)sidx * R + r] = tile[tx * 33 + ty + 8 * k];
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void bwd_nchw_to_nhwc_kernel(const float* in, int B, int C, int HW, float* out) { transpose_batched(in, B, C, HW, out); }
__global__ __launch_bounds__(256) void bwd_nhwc_to_nchw_kernel(const float* in, int B, int C, int HW, float* out) { transpose_batched(in, B, HW, C, out); }
}  // namespace ddif

namespace ddif {
// ----------------------------------------------------------------------------------------------------------------
// Backward of the part of `Block` in front of its convolution (models/sr3_dwt.py:288-300):
//     a = Dropout(SiLU(GroupNorm_1group(x)))       x_hat = (x - mean_b) * rstd_b,  y = gamma_c x_hat + beta_c,  a = silu(y) * m
// (m = 0 or 1/(1-p): the train-mode plan's dropout-site mask; NULL = eval).  Given dA (from the conv's dgrad):
//     dy = dA * m * silu'(y)            silu'(y) = s (1 + y (1 - s)),  s = sigmoid(y)
//     dgamma_c = sum_{b,p} dy x_hat     dbeta_c = sum_{b,p} dy
//     dx = rstd_b (gamma_c dy - S1_b / N - x_hat S2_b / N),   S1_b = sum_{c,p} gamma_c dy,  S2_b = sum_{c,p} gamma_c dy x_hat,  N = C H W
// Everything NHWC, C % 4 == 0.  Reductions: fp64, fixed order (thread-strided partials -> LDS tree in index order ->
// per-chunk partials in memory -> serial sums), no atomics: bitwise reproducible.
__global__ __launch_bounds__(256) void gnb_stats_kernel(const float* x, size_t per_sample, int nchunk, double* part /* [B][nchunk][2] */) {
    DDIF_DYN_SMEM(smem_);
    double* red = reinterpret_cast<double*>(smem_);  // [2][256]
    const int b = blockIdx.y, tid = threadIdx.x;
    const size_t per = (per_sample / 4 + nchunk - 1) / nchunk;
    const size_t i0 = (size_t)blockIdx.x * per, i1 = i0 + per < per_sample / 4 ? i0 + per : per_sample / 4;
    double s1 = 0.0, s2 = 0.0;
    for (size_t i = i0 + tid; i < i1; i += 256) {
        const float4 v = *reinterpret_cast<const float4*>(x + (size_t)b * per_sample + i * 4);
        s1 += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
        s2 += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
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
// The consumers below take the fp64 partials {sum, sum of squares} [B][np][2] themselves -- the ones gnb_stats_kernel wrote or, in the training
// step, the ones the FORWARD producer of x left behind (Tensor::st) -- and every wavefront finalises mean / rstd for itself
// (gn_finalize_wave, the same code the forward prologues run): no statistics launch of its own, no second read of x.
// a = silu(gamma x_hat + beta) * mask     grid = (chunks, B)
__global__ __launch_bounds__(256) void gnb_act_kernel(const float* x, const double* st, int np, const float* gamma, const float* beta, const float* mask, int HW,
                                                      int C, int silu, float* out) {
    const int b = blockIdx.y;
    float mean, rstd;
    gn_finalize_wave(st, np, nullptr, 0, b, (double)C * HW, &mean, &rstd);
    const size_t n4 = (size_t)HW * C / 4, base = (size_t)b * HW * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int c = (int)((i * 4) % C);
        const float4 v = *reinterpret_cast<const float4*>(x + base + i * 4);
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float y = fmaf(((&v.x)[k] - mean) * rstd, gamma[c + k], beta[c + k]);
            o[k] = silu ? dd_silu(y) : y;
            if (mask) o[k] *= mask[base + i * 4 + k];
        }
        *reinterpret_cast<float4*>(out + base + i * 4) = make_float4(o[0], o[1], o[2], o[3]);
    }
}
__device__ __forceinline__ float gnb_dy(float xh, float g, float bt, float da, float m, int silu) {
    if (!silu) return da * m;  // GroupNorm alone (SelfAttention.norm, the attention prenorms)
    const float y = fmaf(xh, g, bt);
    const float s = dd_sigmoid(y);
    return da * m * (s * (1.f + y * (1.f - s)));
}
// per (sample, pixel chunk, channel): {sum dy, sum dy x_hat}.  grid = (nchunk, B); thread = (pixel row r, channel quad q)
// TWO = 1: x is the channel concatenation of two tensors (x: c0 channels, x1: C - c0; 4 | c0) that is never materialised -- FastAttnCondInjection's
// prenorm_x over cat[h, skip] -- with the statistics from both producers' partials.
template <int TWO>
__global__ __launch_bounds__(256) void gnb_bwd_partial_kernel(const float* x, const float* x1, int c0, const float* da, const float* mask, const double* st, int np,
                                                              const double* st1, int np1, const float* gamma, const float* beta, int HW, int C, int nchunk, int silu,
                                                              double* cpart /* [B][nchunk][C][2] */, double* gpart /* nullable: [B][nchunk][2] */) {
    // gpart (the training step, round 4): this chunk's gamma-weighted channel sums {sum_c gamma_c sum dy, sum_c gamma_c sum dy x_hat} -- what the dx launch needs
    // of the partials (nchunk pairs per sample instead of nchunk x C): it then forms S_b itself and the reduce launch leaves the gradient chain
    DDIF_DYN_SMEM(smem_);
    double* red = reinterpret_cast<double*>(smem_);  // [256][8]
    const int b = blockIdx.y, tid = threadIdx.x;
    double g0 = 0.0, g1 = 0.0;  // thread 0: accumulated over the channel passes
    const int C4 = C / 4, rows = 256 / C4 > 0 ? 256 / C4 : 1;
    float mean, rstd;
    gn_finalize_wave(st, np, TWO ? st1 : nullptr, TWO ? np1 : 0, b, (double)C * HW, &mean, &rstd);
    const int per = (HW + nchunk - 1) / nchunk;
    const int p0 = blockIdx.x * per, p1 = p0 + per < HW ? p0 + per : HW;
    for (int q0 = 0; q0 < C4; q0 += 256) {  // C > 1024: several passes over channel quads
        const int q = q0 + tid % (C4 < 256 ? C4 : 256), r = tid / (C4 < 256 ? C4 : 256);
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (q < C4 && r < rows) {
            for (int p = p0 + r; p < p1; p += rows) {
                const size_t e = ((size_t)b * HW + p) * C + q * 4;
                const size_t px = (size_t)b * HW + p;
                const float4 xv = *reinterpret_cast<const float4*>(!TWO ? x + e : (q * 4 < c0 ? x + px * c0 + q * 4 : x1 + px * (C - c0) + (q * 4 - c0)));
                const float4 dv = *reinterpret_cast<const float4*>(da + e);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float xh = ((&xv.x)[k] - mean) * rstd;
                    const float dy = gnb_dy(xh, gamma[q * 4 + k], beta[q * 4 + k], (&dv.x)[k], mask ? mask[e + k] : 1.f, silu);
                    acc[2 * k] += (double)dy;
                    acc[2 * k + 1] += (double)dy * (double)xh;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) red[tid * 8 + k] = acc[k];
        __syncthreads();
        double t0 = 0.0, t1 = 0.0;
        if (q < C4 && r == 0) {  // fixed order over the pixel rows
            const int stride = C4 < 256 ? C4 : 256;
            for (int k = 0; k < 8; ++k) {
                double s = 0.0;
                for (int rr = 0; rr < rows; ++rr) s += red[(rr * stride + tid) * 8 + k];
                cpart[(((size_t)b * nchunk + blockIdx.x) * C + q * 4 + k / 2) * 2 + (k & 1)] = s;
                if (k & 1) t1 += (double)gamma[q * 4 + k / 2] * s;
                else t0 += (double)gamma[q * 4 + k / 2] * s;
            }
        }
        __syncthreads();
        if (gpart) {  // workgroup-uniform: fixed-order tree over the channel-quad threads (threads past them hold zeros)
            red[tid] = t0;
            red[256 + tid] = t1;
            __syncthreads();
            for (int stp = 128; stp >= 1; stp >>= 1) {
                if (tid < stp) {
                    red[tid] += red[tid + stp];
                    red[256 + tid] += red[256 + tid + stp];
                }
                __syncthreads();
            }
            if (tid == 0) {
                g0 += red[0];
                g1 += red[256];
            }
            __syncthreads();
        }
    }
    if (gpart && tid == 0) {
        gpart[((size_t)b * nchunk + blockIdx.x) * 2 + 0] = g0;
        gpart[((size_t)b * nchunk + blockIdx.x) * 2 + 1] = g1;
    }
}
// dgamma / dbeta of MANY GroupNorms in one launch (the training step, round 4): every GroupNorm backward of the iteration leaves its per-chunk partials in a
// buffer of its own and this launch, at the end of the reverse program, turns all of them into parameter gradients -- one launch instead of one per GroupNorm
// on the gradient chain.  Workgroup -> (record, 32 channels) through the records' first-block prefix; the body is gnb_bwd_reduce_kernel's channel branch.
// (GnRedRec: ddif_dev.h -- {partials, dgamma, dbeta, C, nchunk, first workgroup})
__global__ __launch_bounds__(1024) void gnb_bwd_reduce_all_kernel(const GnRedRec* recs, int nrec, int B) {
    DDIF_DYN_SMEM(smem_);
    double(*red)[1024] = reinterpret_cast<double(*)[1024]>(smem_);  // [2][1024]
    const int tid = threadIdx.x;
    int ri = 0;
    for (int k = 1; k < nrec; ++k)
        if ((int)blockIdx.x >= recs[k].blk0) ri = k;
    const GnRedRec rc = recs[ri];
    const int C = rc.C, nchunk = rc.nchunk;
    const int c = ((int)blockIdx.x - rc.blk0) * 32 + (tid & 31), sl = tid >> 5;  // slice sl takes the pairs (b, k) = sl, sl + 32, ...
    const int npair = B * nchunk;
    const int cc = c < C ? c : C - 1;
    double s0 = 0.0, s1 = 0.0;
    for (int p0 = sl; p0 < npair; p0 += 32 * 8) {
        double v0[8], v1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int pr = p0 + 32 * u < npair ? p0 + 32 * u : sl;
            v0[u] = rc.cpart[((size_t)pr * C + cc) * 2 + 0];
            v1[u] = rc.cpart[((size_t)pr * C + cc) * 2 + 1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (p0 + 32 * u < npair) {
                s0 += v0[u];
                s1 += v1[u];
            }
    }
    red[0][tid] = s0;
    red[1][tid] = s1;
    __syncthreads();
    for (int st = 16; st >= 1; st >>= 1) {
        if (sl < st) {
            red[0][tid] += red[0][tid + st * 32];
            red[1][tid] += red[1][tid + st * 32];
        }
        __syncthreads();
    }
    if (sl == 0 && c < C) {
        if (rc.dbeta) rc.dbeta[c] = (float)red[0][tid];
        if (rc.dgamma) rc.dgamma[c] = (float)red[1][tid];
    }
}
// Everything that follows the per-chunk partials, in ONE launch of 1024-thread workgroups (it replaces a plane-sum launch and a finalize launch):
//   workgroups [0, ceil(C / 32)):  dgamma[c] = sum_{b,k} cpart[b][k][c][1], dbeta[c] = sum_{b,k} cpart[b][k][c][0]  for 32 channels each
//                                  (thread = (channel, one of 32 interleaved (b, k) slices), eight loads in flight, fixed-order LDS tree over the slices)
//   workgroups [nC, nC + B):       S[b] = {sum_{c,k} gamma[c] cpart[b][k][c][0], .. [1]} / C   (divided by HW in the consumer: N = C * HW)
// fp64, fixed order: bitwise reproducible.
constexpr int GNB_RED_NT = 1024;
__global__ __launch_bounds__(GNB_RED_NT) void gnb_bwd_reduce_kernel(const double* cpart, const float* gamma, int B, int nchunk, int C, float* dgamma, float* dbeta,
                                                                     float* S /* [B][2] */) {
    DDIF_DYN_SMEM(smem_);
    double(*red)[GNB_RED_NT] = reinterpret_cast<double(*)[GNB_RED_NT]>(smem_);  // [2][GNB_RED_NT]
    const int tid = threadIdx.x;
    const int nC = (C + 31) / 32;
    double s0 = 0.0, s1 = 0.0;
    if ((int)blockIdx.x < nC) {
        const int c = blockIdx.x * 32 + (tid & 31), sl = tid >> 5;  // slice sl takes the pairs (b, k) = sl, sl + 32, ...
        const int npair = B * nchunk;
        const int cc = c < C ? c : C - 1;
        for (int p0 = sl; p0 < npair; p0 += 32 * 8) {
            double v0[8], v1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int pr = p0 + 32 * u < npair ? p0 + 32 * u : sl;
                v0[u] = cpart[((size_t)pr * C + cc) * 2 + 0];
                v1[u] = cpart[((size_t)pr * C + cc) * 2 + 1];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (p0 + 32 * u < npair) {
                    s0 += v0[u];
                    s1 += v1[u];
                }
        }
        red[0][tid] = s0;
        red[1][tid] = s1;
        __syncthreads();
        for (int st = 16; st >= 1; st >>= 1) {  // over the 32 slices (tid >> 5), channel kept in the low 5 bits
            if (sl < st) {
                red[0][tid] += red[0][tid + st * 32];
                red[1][tid] += red[1][tid + st * 32];
            }
            __syncthreads();
        }
        if (sl == 0 && c < C) {
            if (dbeta) dbeta[c] = (float)red[0][tid];
            if (dgamma) dgamma[c] = (float)red[1][tid];
        }
    } else {
        const int b = blockIdx.x - nC;
        const int n = nchunk * C;  // entries (k, c) of sample b, c fastest
        const double* base = cpart + (size_t)b * n * 2;
        for (int i0 = tid; i0 < n; i0 += GNB_RED_NT * 4) {
            double v0[4], v1[4], gm[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + GNB_RED_NT * u < n ? i0 + GNB_RED_NT * u : tid % n;
                gm[u] = (double)gamma[i % C];
                v0[u] = base[(size_t)i * 2 + 0];
                v1[u] = base[(size_t)i * 2 + 1];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + GNB_RED_NT * u < n) {
                    s0 += gm[u] * v0[u];
                    s1 += gm[u] * v1[u];
                }
        }
        red[0][tid] = s0;
        red[1][tid] = s1;
        __syncthreads();
        for (int st = GNB_RED_NT / 2; st >= 1; st >>= 1) {
            if (tid < st) {
                red[0][tid] += red[0][tid + st];
                red[1][tid] += red[1][tid + st];
            }
            __syncthreads();
        }
        if (tid == 0) {
            S[b * 2 + 0] = (float)(red[0][0] / ((double)C));
            S[b * 2 + 1] = (float)(red[1][0] / ((double)C));
        }
    }
}
// dx = rstd (gamma dy - m1 - x_hat m2)  (+ res: the gradient arriving over the residual path of a ResnetBlock / SelfAttention, added here instead
// of by a launch of its own)
// TWO = 1: the two sources as above, and the gradient goes straight to the two tensors' own gradients (dx: c0 channels, dx1: C - c0) -- no cat / split launches
// gpart != nullptr: S_b from the partial launch's gamma-weighted chunk sums (nchunk <= 32 pairs, summed in chunk order by every thread) instead of `S`
template <int TWO>
__global__ __launch_bounds__(256) void gnb_bwd_dx_kernel(const float* x, const float* x1, int c0, const float* da, const float* mask, const double* st, int np,
                                                         const double* st1, int np1, const float* gamma, const float* beta, const float* S, const float* res, int HW, int C,
                                                         int silu, float* dx, float* dx1, const double* gpart, int nchunk) {
    const int b = blockIdx.y;
    float mean, rstd;
    gn_finalize_wave(st, np, TWO ? st1 : nullptr, TWO ? np1 : 0, b, (double)C * HW, &mean, &rstd);
    float S0, S1;
    if (gpart) {
        const double* gp = gpart + (size_t)b * nchunk * 2;
        double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
        for (int k = 0; k < nchunk; ++k) {
            s0 += gp[2 * k];
            s1 += gp[2 * k + 1];
        }
        S0 = (float)(s0 / (double)C);
        S1 = (float)(s1 / (double)C);
    } else {
        S0 = S[b * 2];
        S1 = S[b * 2 + 1];
    }
    const float m1 = S0 / (float)HW, m2 = S1 / (float)HW;
    const size_t n4 = (size_t)HW * C / 4, base = (size_t)b * HW * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const size_t pl = (i * 4) / C;  // pixel of the sample
        const int c = (int)(i * 4 - pl * C);
        const size_t px = (size_t)b * HW + pl;
        const float4 xv = *reinterpret_cast<const float4*>(!TWO ? x + base + i * 4 : (c < c0 ? x + px * c0 + c : x1 + px * (C - c0) + (c - c0)));
        const float4 dv = *reinterpret_cast<const float4*>(da + base + i * 4);
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh = ((&xv.x)[k] - mean) * rstd;
            const float dy = gnb_dy(xh, gamma[c + k], beta[c + k], (&dv.x)[k], mask ? mask[base + i * 4 + k] : 1.f, silu);
            o[k] = rstd * (gamma[c + k] * dy - m1 - xh * m2);
        }
        if (res) {
            const float4 rv = *reinterpret_cast<const float4*>(res + base + i * 4);
            o[0] += rv.x;
            o[1] += rv.y;
            o[2] += rv.z;
            o[3] += rv.w;
        }
        float* dst = !TWO ? dx + base + i * 4 : (c < c0 ? dx + px * c0 + c : dx1 + px * (C - c0) + (c - c0));
        *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
    }
}
// out[b * C + c] = sum over pixels of in[b, c, :] (NCHW planes; fixed-order tree): the FeatureWiseAffine gradient d(noise_func output)
// of a ResnetBlock (models/sr3_dwt.py:241-258, 322) is this row sum of block1's output gradient
__global__ __launch_bounds__(256) void plane_sum_nchw_kernel(const float* in, int HW, float* out) {
    DDIF_DYN_SMEM(smem_);
    double* red = reinterpret_cast<double*>(smem_);  // [256]
    const int tid = threadIdx.x;
    const float* p = in + (size_t)blockIdx.x * HW;
    double s = 0.0;
    for (int i = tid; i < HW; i += 256) s += (double)p[i];
    red[tid] = s;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (tid < st) red[tid] += red[tid + st];
        __syncthreads();
    }
    if (tid == 0) out[blockIdx.x] = (float)red[0];
}
// 1x1 convolutions ride the 3x3 backward kernels for now (correctness first): the (Cout, Cin) weights become the centre tap of a
// zero 3x3 kernel -- its dgrad IS the 1x1 dgrad; the wgrad kernel contracts the centre tap only (WgradArgs::centre_only)
__global__ void embed_1x1_kernel(const float* w1, size_t n /* Cout * Cin */, float* w3) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * 9; i += (size_t)gridDim.x * blockDim.x) w3[i] = (i % 9 == 4) ? w1[i / 9] : 0.f;
}
// ---- SiLU alone in front of a conv (FastAttnCondInjection.ffn: conv3x3 -> SiLU -> conv3x3, models/sr3_dwt.py:528-533)
__global__ void silu_fwd_kernel(const float* x, size_t n, float* a) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = dd_silu(x[i]);
}
__global__ void silu_bwd_kernel(const float* x, const float* da, size_t n, float* dx) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float y = x[i], s = dd_sigmoid(y);
        dx[i] = da[i] * (s * (1.f + y * (1.f - s)));
    }
}
// ---- Downsample = conv3x3 stride 2 pad 1 (:276-282): its backward is the stride-1 backward on dY with zeros inserted
//      (dYu[2oy][2ox] = dY[oy][ox]): dX[i] = sum_k dYu[i + 1 - k] w[k] and dW[k] = sum_i dYu[i] x[i + k - 1] are exactly the strided sums
__global__ void zero_stuff_nchw_to_nhwc_kernel(const float* dy, int B, int C, int Ho, int Wo, int H, int W, float* out /* (B,H,W,C) */) {
    const size_t total = (size_t)B * H * W * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int x = (int)((i / C) % W), y = (int)((i / ((size_t)C * W)) % H);
        const size_t b = i / ((size_t)C * W * H);
        float v = 0.f;
        if (!(x & 1) && !(y & 1) && (y >> 1) < Ho && (x >> 1) < Wo) v = dy[((b * C + c) * Ho + (y >> 1)) * Wo + (x >> 1)];
        out[i] = v;
    }
}
// ---- Upsample = nearest x2 then conv3x3 (:266-273): the conv sees x_up; dX = 2x2 sum pooling of d(x_up)
__global__ void upsample2_nchw_to_nhwc_kernel(const float* x, int B, int C, int H, int W, float* out /* (B,2H,2W,C) */) {
    const size_t total = (size_t)B * 4 * H * W * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int xx = (int)((i / C) % (2 * W)), yy = (int)((i / ((size_t)C * 2 * W)) % (2 * H));
        const size_t b = i / ((size_t)C * 4 * W * H);
        out[i] = x[((b * C + c) * H + (yy >> 1)) * W + (xx >> 1)];
    }
}
__global__ void sumpool2_nhwc_to_nchw_kernel(const float* dxu /* (B,2H,2W,C) */, int B, int C, int H, int W, float* dx /* (B,C,H,W) */) {
    const size_t total = (size_t)B * C * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const int c = (int)((i / ((size_t)W * H)) % C);
        const size_t b = i / ((size_t)W * H * C);
        const float* p = dxu + ((b * 2 * H + 2 * y) * 2 * W + 2 * x) * C + c;
        dx[i] = (p[0] + p[C]) + (p[(size_t)2 * W * C] + p[(size_t)2 * W * C + C]);
    }
}

// packed fp32 fragment order of kernels_conv.h for FORWARD weights, 16-channel chunks (cf. pack_dgrad_weights_kernel):
//   [n-block of 32 couts][chunk of 16 cins][tap][k8][half h][j][4]  with cin = chunk*16 + k8*8 + 4h + i, cout = nb*32 + j.
// ks = 1: the (Cout, Cin) weights become the centre tap of a zero 3x3 kernel.
__global__ void pack_fwd_weights_kernel(const float* w, int Cout, int Cin, int ks, int n_chunks, int nb_pad, float* out) {
    const size_t total = (size_t)nb_pad * n_chunks * 9 * 2 * 256;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        size_t r = idx;
        const int i = (int)(r % 4); r /= 4;
        const int j = (int)(r % 32); r /= 32;
        const int h = (int)(r % 2); r /= 2;
        const int k8 = (int)(r % 2); r /= 2;
        const int tap = (int)(r % 9); r /= 9;
        const int ch = (int)(r % n_chunks); r /= n_chunks;
        const int nbi = (int)r;
        const int ci = ch * 16 + k8 * 8 + 4 * h + i, co = nbi * 32 + j;
        float v = 0.f;
        if (co < Cout && ci < Cin) {
            if (ks == 3) v = w[((size_t)co * Cin + ci) * 9 + tap];
            else if (tap == 4) v = w[(size_t)co * Cin + ci];
        }
        out[idx] = v;
    }
}
}  // namespace ddif

namespace ddif {
// The same weights as three bf16 planes for the bf16x3 conv kernels (host counterpart: pack_conv_x3 in ddif_net.cpp), 16-channel chunks:
//   [n-block of 32][chunk of 16][tap][plane hi|mid|lo][lane = h*32 + j][8 bf16: cin = chunk*16 + 8h + t];  `flip` = dgrad form (taps mirrored,
//   in / out channels swapped: value = w[co = cin'][ci = cout'][8 - tap]), ks = 1 = centre tap of a zero 3x3 kernel.
__global__ void pack_weights_x3_kernel(const float* w, int Cout, int Cin, int ks, int flip, int n_chunks, int nb_pad, float* out) {
    // forward: rows (n-blocks) = conv couts, contraction = conv cins;  dgrad: rows = conv cins, contraction = conv couts
    const int Crow = flip ? Cin : Cout, Ccon = flip ? Cout : Cin;
    unsigned short* o = reinterpret_cast<unsigned short*>(out);
    const size_t total = (size_t)nb_pad * n_chunks * 9 * 2 * 32 * 8;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        size_t r = idx;
        const int t = (int)(r % 8); r /= 8;
        const int j = (int)(r % 32); r /= 32;
        const int h = (int)(r % 2); r /= 2;
        const int tap = (int)(r % 9); r /= 9;
        const int ch = (int)(r % n_chunks); r /= n_chunks;
        const int nbi = (int)r;
        const int kc = ch * 16 + 8 * h + t, row = nbi * 32 + j;
        float val = 0.f;
        if (row < Crow && kc < Ccon) {
            const int co = flip ? kc : row, ci = flip ? row : kc;
            const int tp = flip ? 8 - tap : tap;
            if (ks == 3) val = w[((size_t)co * Cin + ci) * 9 + tp];
            else if (tap == 4) val = w[(size_t)co * Cin + ci];
        }
        unsigned hi, mid, lo;
        dd_split3(val, &hi, &mid, &lo);
        const size_t fl = ((((size_t)nbi * n_chunks + ch) * 9 + tap) * 3) * 256 + (size_t)(h * 32 + j) * 4;  // float index of the lane's 16 bytes, plane 0
        o[fl * 2 + t] = (unsigned short)hi;
        o[(fl + 256) * 2 + t] = (unsigned short)mid;
        o[(fl + 512) * 2 + t] = (unsigned short)lo;
    }
}
}  // namespace ddif
