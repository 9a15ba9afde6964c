// Fused feed-forward half of a decoder block at the high-resolution levels (round 4), gfx950.  OPT-IN (DDIF_FFNFUSE=1): correct (emulator + MI355X parity,
// bit-equal across batch sizes) but NOT faster than the two launches it replaces -- 98 us (first form) / 106 us (deeper weight ring, W1 and the next input
// tile prefetched) against 46.5 + 47.7 us at 64 x 64, B = 64 (profiles/r04_t_ffn_fused_ab.txt).  The arithmetic says why: the halo recompute makes it 1.19 x
// the matrix work, one 32 x 32 accumulator tile per wave makes both phases LDS-read-bound (4 KiB of fragments per MFMA triple), and it runs at the same ~30 %
// of its matrix floor as the kernels it replaces; what it saves -- the intermediate's round trip and one launch boundary -- is less than what it adds.
//
// Reference: FastAttnCondInjection.ffn + residual (models/sr3_dwt.py:528-533, 576), eval mode:
//     out = a + ffn[3](ffn[2](SiLU(ffn[0](a))))      ffn[0]: 3x3, C -> 2C, no bias;  ffn[3] o ffn[2]: ONE 3x3 conv 2C -> C + bias (merged at commit, ".ffn.23")
// As two launches of conv_mfma_kernel the 2C-channel intermediate makes a round trip through memory (134 MB written + read per block at 64 x 64, B = 64 -- more
// than the block's input and output together) and each launch stages its input once per 32-cout tile.  Here a workgroup owns a 16 x 16 output tile and keeps the
// intermediate ON CHIP:
//   (x)  the 20 x 20 input halo tile (all C channels), pre-scaled and split into two fp16 planes, to LDS            -- one staging per work item
//   (A)  y = SiLU(conv3x3(x, W0)) on the 18 x 18 halo of the output tile (eleven 32-pixel blocks x 2C/32 cout blocks over eight waves; W0 fragments straight
//        from L2 into the operand registers, one tap ahead), zero outside the image (the second conv pads y, not x), split into two fp16 planes, to LDS
//   (B)  out = conv3x3(y, W1) + bias + a on the 16 x 16 tile (one 32-pixel block per wave; W1 in 16-channel chunks through LDS, double-buffered in the
//        region the input tile no longer needs), GroupNorm partial of the output, 16-byte NHWC stores.
// Arithmetic: the f16x2 split products of kernels_conv.h MATH = 3 (operands x 2^4 / x 2^10, hi * hi + hi * lo + lo * hi on v_mfma_f32_32x32x16_f16, fp32
// accumulate), the same expressions for SiLU, bias and residual as the two-launch form; the K order differs (fp32-class, not bitwise the same sums).  The halo of y
// is computed by every tile that needs it (18^2 / 16^2 = 1.27 x the first conv's matrix work): each value of y depends on x only, so a tile of a batch is the bits
// of the same tile run alone.
#pragma once
#include "kernels_conv.h"

namespace ddif {

struct FfnFuseArgs {
    const float* x;      // [B, H, W, C]  the block's input a (also the residual)
    const float* w0;     // ffn[0] in f16x2 pack order (ddif_net.cpp pack_conv_f16, ck = 16)
    const float* w1;     // merged ffn[3] o ffn[2], f16x2 pack
    const float* bias;   // [C] of the merged conv
    float* out;          // [B, H, W, C]
    double* st_out;      // GroupNorm partials of `out`: [B][tiles][2]
    int B, b0, H, W;
    int tiles_x, tiles_y;
};

template <int C, int CM>
struct FfnFuseGeom {
    static constexpr int TH = 16, TW = 16, XH = TH + 4, XW = TW + 4, YH = TH + 2, YW = TW + 2;
    static constexpr int SX = C / 16, SY = CM / 16;                 // 16-channel slabs
    static constexpr int LDX = SX * 16 + 4, LDY = SY * 16 + 4;      // floats per staged pixel: slabs x (hi 8 | lo 8 floats) + 16 B pad
    static constexpr int RPX = 56, RPY = 56;                        // row pads: conflict-free ds_read_b128 fragments (bank enumeration)
    static constexpr int LRX = XW * LDX + RPX, LRY = YW * LDY + RPY;
    static constexpr int XFL = XH * LRX, YFL = YH * LRY;
    static constexpr int WCH = 9 * 2 * 256;                         // floats of one 16-channel chunk of W1 for one 32-cout block (18 KB)
    static_assert(2 * WCH * (C / 32) <= XFL, "the W1 double buffer lives in the input tile's region");
    static constexpr size_t smem = (size_t)(XFL + YFL + 64) * sizeof(float) + 32 * sizeof(double);
    static constexpr int NYPIX = YH * YW, NMB = (NYPIX + 31) / 32;  // 324 halo pixels of y = 11 blocks of 32
};

template <int C, int CM>
__global__ __launch_bounds__(512) void ffn_fused_kernel(FfnFuseArgs a) {
    using G = FfnFuseGeom<C, CM>;
    static_assert(C == 32 && CM == 64, "instantiated for the 64-pixel-tile level of the engine network (inner_channel 32)");
    constexpr int TH = G::TH, TW = G::TW, XW = G::XW, XH = G::XH, YW = G::YW;
    constexpr int LDX = G::LDX, LDY = G::LDY, LRX = G::LRX, LRY = G::LRY, NYPIX = G::NYPIX, NMB = G::NMB, WCH = G::WCH;
    constexpr int NBA = CM / 32, NBB = C / 32;   // cout blocks of the two convs
    constexpr int NCH0 = C / 16, NCH1 = CM / 16; // 16-channel chunks (K) of the two convs
    static_assert(NBB == 1 && NBA == 2, "wave layout below");
    constexpr int XIT = (XH * XW * (C / 4) + 511) / 512;  // float4 staging items per thread

    DDIF_DYN_SMEM(smem);
    float* Xs = reinterpret_cast<float*>(smem);
    float* Ys = Xs + G::XFL;
    float* Bs = Ys + G::YFL;                                                            // [32] bias
    double* red = reinterpret_cast<double*>(smem + (size_t)(G::XFL + G::YFL + 64) * sizeof(float));  // [8 waves][2]
    float* W1s = Xs;                                                                    // phase B: [2][WCH]

    const int tid = threadIdx.x, lane = tid & 63;
#ifdef DDIF_EMU
    const int wave = tid >> 6;
#else
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const int h = lane >> 5, j = lane & 31;
    const int tiles = a.tiles_x * a.tiles_y;
    const int nwork = a.B * tiles;
    int w0, w1;
    wg_work_range(nwork, &w0, &w1);
    if (w0 >= w1) return;
    if (tid < 32) Bs[tid] = a.bias[tid];

    // phase A geometry of this wave: cout block nbA, pixel blocks mbA0 + 4k of the y halo (linear index m = row * 18 + col)
    const int nbA = wave & 1, mbA0 = wave >> 1;
    int xoffA[3];  // float offset of this lane's y pixel in the x tile (tap (0,0)), + 4h
    bool mokA[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int mb = mbA0 + 4 * k;
        const int mr = mb * 32 + j;
        const int m = mr < NYPIX ? mr : NYPIX - 1;
        mokA[k] = mb < NMB && mr < NYPIX;
        xoffA[k] = (m / YW) * LRX + (m % YW) * LDX + 4 * h;
    }
    const int nkA = (mbA0 + 8 < NMB) ? 3 : 2;  // wave-uniform: pixel blocks this wave really owns
    // phase B geometry: pixel block `wave` of the 16 x 16 tile (two rows of 16)
    const int pB = wave * 32 + j, pyB = pB / TW, pxB = pB % TW;
    const int yoffB = pyB * LRY + pxB * LDY + 4 * h;

    const char* w0base = reinterpret_cast<const char*>(a.w0) + (size_t)nbA * NCH0 * 9 * 2 * 1024 + (size_t)lane * 16;

    // the input tile of a work item: requested one item ahead (during the previous item's phase B), validity as a mask
    float4 raw[XIT];
    unsigned rok = 0;
    auto load_x = [&](int work) {
        const int bl = work / tiles, b = a.b0 + bl, t = work - bl * tiles;
        const int ty0 = (t / a.tiles_x) * TH, tx0 = (t % a.tiles_x) * TW;
        rok = 0;
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int i = tid + it * 512, pix = i / (C / 4), c4 = i % (C / 4);
            const int py = pix / XW, px = pix % XW;
            const int iy = ty0 - 2 + py, ix = tx0 - 2 + px;
            const bool ok = pix < XH * XW && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            rok |= (ok ? 1u : 0u) << it;
            const size_t off = ok ? (((size_t)b * a.H + iy) * a.W + ix) * C + c4 * 4 : 0;
            raw[it] = *reinterpret_cast<const float4*>(a.x + off);
        }
    };
    // W0 fragments of this wave: a ring of RING steps (step = chunk * 9 + tap: 2 KiB, hi | lo), filled ahead of the MFMAs by that many steps
    constexpr int NSTEP0 = NCH0 * 9, RING = 6;
    float4 wr[RING][2];
    auto load_w0 = [&](int slot, int step) {
        const char* p = w0base + (size_t)step * 2048;
        wr[slot][0] = *reinterpret_cast<const float4*>(p);
        wr[slot][1] = *reinterpret_cast<const float4*>(p + 1024);
    };
    // W1: chunks of 18 KiB (1152 float4 / 512 threads) travel registers -> LDS two chunks ahead of their MFMAs: chunks 0 and 1 are requested at the head of the
    // item, chunk c + 2 when chunk c has been written to LDS (register slot = chunk parity)
    float4 wst[2][3];
    auto load_w1 = [&](int ch) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int i = tid + u * 512;
            wst[ch & 1][u] = *reinterpret_cast<const float4*>(a.w1 + (size_t)ch * WCH + (size_t)(i < WCH / 4 ? i : 0) * 4);
        }
    };
    auto store_w1 = [&](int buf, int ch) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int i = tid + u * 512;
            if (i < WCH / 4) *reinterpret_cast<float4*>(&W1s[buf * WCH + i * 4]) = wst[ch & 1][u];
        }
    };
    load_x(w0);

    for (int work = w0; work < w1; ++work) {
        const int bl = work / tiles, b = a.b0 + bl, t = work - bl * tiles;
        const int ty0 = (t / a.tiles_x) * TH, tx0 = (t % a.tiles_x) * TW;
#pragma unroll
        for (int sidx = 0; sidx < RING; ++sidx) load_w0(sidx, sidx);
        load_w1(0);
        load_w1(1);
        __syncthreads();  // the previous item's phase B is done with W1s (= Xs) and Ys
        // ---- (x) input halo tile -> two fp16 planes in LDS
        {
#pragma unroll
            for (int it = 0; it < XIT; ++it) {
                const int i = tid + it * 512, pix = i / (C / 4), c4 = i % (C / 4);
                if (pix < XH * XW) {
                    const int py = pix / XW, px = pix % XW;
                    const bool okx = (rok >> it) & 1u;
                    float v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = okx ? (&raw[it].x)[q] * DDIF_F16_ASCALE : 0.f;
                    unsigned h01, l01, h23, l23;
                    dd_split2_pair(v[0], v[1], &h01, &l01);
                    dd_split2_pair(v[2], v[3], &h23, &l23);
                    float* d = &Xs[py * LRX + px * LDX + (c4 >> 2) * 16 + (c4 & 3) * 2];
                    *reinterpret_cast<uint2*>(d) = make_uint2(h01, h23);
                    *reinterpret_cast<uint2*>(d + 8) = make_uint2(l01, l23);
                }
            }
        }
        __syncthreads();

        // ---- (A) y = SiLU(conv3x3(x, W0)) on the halo: this wave's cout block, up to three pixel blocks; steps = (chunk, tap), weights one step ahead
        {
            f32x16 acc[3];
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
#pragma unroll
            for (int step = 0; step < NSTEP0; ++step) {
                const int ch = step / 9, tap = step % 9;
                const int toff = (tap / 3) * LRX + (tap % 3) * LDX + ch * 16;
                float4 xa[3][2];
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    if (k < nkA) {
                        xa[k][0] = *reinterpret_cast<const float4*>(&Xs[xoffA[k] + toff]);
                        xa[k][1] = *reinterpret_cast<const float4*>(&Xs[xoffA[k] + toff + 8]);
                    }
                DDIF_SCHED_FENCE();
                const float4 whi = wr[step % RING][0], wlo = wr[step % RING][1];
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    if (k < nkA) {
                        f32x16 c = acc[k];
                        c = DDIF_MFMA_32x32x16_F16(wlo, xa[k][0], c);  // lo * hi
                        c = DDIF_MFMA_32x32x16_F16(whi, xa[k][1], c);  // hi * lo
                        c = DDIF_MFMA_32x32x16_F16(whi, xa[k][0], c);  // hi * hi
                        acc[k] = c;
                    }
                if (step + RING < NSTEP0) load_w0(step % RING, step + RING);  // the step RING ahead takes the slot just consumed
                DDIF_SCHED_FENCE();
            }
            // epilogue A: lane (j, h) owns y pixel m and, per quad g, the channels nbA * 32 + 8 g + 4 h .. + 3  -> slab nbA * 2 + g / 2, halves (g & 1) * 8 + 4 h ..
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (k < nkA) {
                    const int mr = (mbA0 + 4 * k) * 32 + j;
                    const int m = mr < NYPIX ? mr : NYPIX - 1;
                    const int yr = m / YW, yc = m % YW;
                    const int iy = ty0 - 1 + yr, ix = tx0 - 1 + yc;
                    const bool in = mokA[k] && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;  // zero padding of the second conv
                    if (mokA[k]) {
                        float* d = &Ys[yr * LRY + yc * LDY + 2 * h];
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            float v[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[q] = in ? dd_silu_scaled(acc[k][4 * g + q] * DDIF_F16_OSCALE, 1.0f / DDIF_F16_ASCALE) : 0.f;
                            unsigned h01, l01, h23, l23;
                            dd_split2_pair(v[0], v[1], &h01, &l01);
                            dd_split2_pair(v[2], v[3], &h23, &l23);
                            float* dd = d + (nbA * 2 + (g >> 1)) * 16 + (g & 1) * 4;
                            *reinterpret_cast<uint2*>(dd) = make_uint2(h01, h23);
                            *reinterpret_cast<uint2*>(dd + 8) = make_uint2(l01, l23);
                        }
                    }
                }
        }
        // ---- (B) out = conv3x3(y, W1) + bias + a: W1 chunk by chunk through LDS (the input tile's region)
        // epilogue operands of this lane's pixel: the residual (exact fp32, from memory)
        const int oy = ty0 + pyB, ox = tx0 + pxB;
        const bool pok = oy < a.H && ox < a.W;
        const size_t opix = pok ? ((size_t)b * a.H + oy) * a.W + ox : (size_t)b * a.H * a.W;
        float4 er[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) er[g] = *reinterpret_cast<const float4*>(a.x + opix * C + 8 * g + 4 * h);
        __syncthreads();  // y complete; every wave is done reading the input tile
        store_w1(0, 0);
        if (2 < NCH1) load_w1(2);
        load_x(work + 1 < w1 ? work + 1 : work);  // the next item's input tile flies during phase B (the last item re-reads its own: never consumed)
        __syncthreads();
        f32x16 accb;
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[r] = 0.f;
#pragma unroll
        for (int ch = 0; ch < NCH1; ++ch) {
            const float* Wc = W1s + (ch & 1) * WCH + lane * 4;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int toff = (tap / 3) * LRY + (tap % 3) * LDY + ch * 16;
                const float4 x0 = *reinterpret_cast<const float4*>(&Ys[yoffB + toff]);
                const float4 x1 = *reinterpret_cast<const float4*>(&Ys[yoffB + toff + 8]);
                const float4 whi = *reinterpret_cast<const float4*>(&Wc[(tap * 2 + 0) * 256]);
                const float4 wlo = *reinterpret_cast<const float4*>(&Wc[(tap * 2 + 1) * 256]);
                accb = DDIF_MFMA_32x32x16_F16(wlo, x0, accb);
                accb = DDIF_MFMA_32x32x16_F16(whi, x1, accb);
                accb = DDIF_MFMA_32x32x16_F16(whi, x0, accb);
            }
            if (ch + 1 < NCH1) {
                store_w1((ch + 1) & 1, ch + 1);  // (the buffer read two chunks ago: every wave passed the barrier below since)
                if (ch + 3 < NCH1) load_w1(ch + 3);
                __syncthreads();
            }
        }
        // epilogue B
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bt = *reinterpret_cast<const float4*>(&Bs[8 * g + 4 * h]);
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = fmaf(accb[4 * g + q], DDIF_F16_OSCALE, (&bt.x)[q]) + (&er[g].x)[q];
            if (pok) {
                *reinterpret_cast<float4*>(a.out + opix * C + 8 * g + 4 * h) = make_float4(v[0], v[1], v[2], v[3]);
                s1 += (v[0] + v[1]) + (v[2] + v[3]);
                s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            }
        }
        if (a.st_out) {
            const float t1 = wave_sum_fast(s1), t2 = wave_sum_fast(s2);  // total in lane 63
            if (lane == 63) {
                red[wave * 2 + 0] = (double)t1;
                red[wave * 2 + 1] = (double)t2;
            }
            __syncthreads();
            if (tid == 0) {
                const size_t pi = ((size_t)b * tiles + t) * 2;
                a.st_out[pi + 0] = ((red[0] + red[2]) + (red[4] + red[6])) + ((red[8] + red[10]) + (red[12] + red[14]));
                a.st_out[pi + 1] = ((red[1] + red[3]) + (red[5] + red[7])) + ((red[9] + red[11]) + (red[13] + red[15]));
            }
        }
    }
}

}  // namespace ddif
