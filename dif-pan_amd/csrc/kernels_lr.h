// Low-resolution convolution (3x3 / 1x1, NHWC fp32, bf16x3 split products) for the 8x8 / 16x16 levels of the UNet.
//
// Same reference ops as kernels_conv.h (models/sr3_dwt.py:288-327 Block / ResnetBlock, :338-339 attention 1x1s,
// :376-396 CondInjection.x_conv, :507-577 FastAttnCondInjection q / attn_out / ffn) -- a different decomposition for
// the regime where a sample has <= 256 pixels and 64-512 channels:
//
//   kernels_conv.h at these levels is bound by its WEIGHT stream, not by MFMAs or activations: M = 64 pixels per sample,
//   so a 64-cout tile stages 55 KB of weights through LDS per 16-channel chunk to feed 54 MFMAs per wave (measured
//   30 us per 128->128 3x3 conv at B = 64 where the matrix work is 3 us), and it only has 128 work items for 256 CUs.
//
//   Here a workgroup (4 wavefronts) owns (sample, 8x8 or 8x16 pixel tile, 32 couts) -- 256 workgroups at B = 64 -- and
//   the waves split K: wave w contracts the 16-channel slabs w, w+4, w+8, ... for ALL pixels of the tile (MB = 2 or 4
//   accumulator blocks of 32 pixels x 32 couts).  Consequences:
//     * weights are not shared between waves, so they never touch LDS: each wave reads its own B-fragment stream
//       (pre-packed per-lane order, 1 KiB per (tap, plane), perfectly coalesced) straight from L2 into the MFMA operand
//       registers through a small register ring that runs 1-3 steps ahead (counted vmcnt, in-order retirement);
//     * the activation tile (whole sample at 8x8) is staged ONCE per <=128-channel phase with the producer-side
//       GroupNorm / SiLU / column-softmax / depthwise prologue applied, as three bf16 planes;
//     * the four K-partials meet in LDS (fixed order: (w0 + w1) + (w2 + w3)) and wave g runs the epilogue of accumulator
//       quad g (bias + time bias, FiLM, SiLU, residual, float4 NHWC stores, GroupNorm partial of the output).
//   Arithmetic is the bf16x3 scheme of kernels_conv.h (operands split hi/mid/lo, six exact bf16 products per fp32
//   product on v_mfma_f32_32x32x16_bf16, small terms first, fp32 accumulate); only the K summation order differs.
#pragma once
#include "kernels_conv.h"

namespace ddif {

template <int KS, int MB, int PRO>
struct LrGeom {
    static constexpr int TH = 8, TW = MB == 2 ? 8 : 16;
    static constexpr bool DWM = PRO == PRO_GN_DW;
    static constexpr int PAD = KS / 2, LP = DWM ? 1 : PAD;
    static constexpr int LH = TH + 2 * LP, LW = TW + 2 * LP;   // loaded tile (halo of the 3x3 / depthwise taps)
    static constexpr int IH = TH + 2 * PAD, IW = TW + 2 * PAD; // tile the MFMAs read
    static constexpr int PC = (MB == 4 && (KS == 3 || DWM)) ? 64 : 128;  // channels staged per phase
    static constexpr int SP = PC / 16;                         // 16-channel slabs per phase
    static constexpr int APIX = SP * 24 + 4;                   // floats per staged pixel: SP x (3 planes x 32 B) + 16 B pad
    static constexpr int HPIX = PC + 4;                        // DWM: floats per pixel of the fp32 scratch tile
    static constexpr int AFL = IH * IW * APIX;
    static constexpr int RFL = 4 * MB * 4 * 64 * 4;            // K-partials: [wave][mb][quad g][lane] float4
    static constexpr int HFL = DWM ? LH * LW * HPIX : 0;
    static constexpr size_t smem = (size_t)((AFL > RFL ? AFL : RFL) + HFL + 16) * sizeof(float);
    static constexpr int SPW = SP / 4;                         // slabs per wave and phase
    static constexpr int TAPS = KS * KS;
    static constexpr int U = (SPW * TAPS) % 3 == 0 ? 3 : ((SPW * TAPS) % 2 == 0 ? 2 : 1);  // weight ring depth (steps)
};

template <int KS, int MB, int PRO, int EPI>
__global__ __launch_bounds__(256) void conv_lr_kernel(ConvArgs a) {
    using G = LrGeom<KS, MB, PRO>;
    constexpr int TH = G::TH, TW = G::TW, LP = G::LP, LH = G::LH, LW = G::LW, IW = G::IW, PC = G::PC, SP = G::SP;
    constexpr int APIX = G::APIX, HPIX = G::HPIX, TAPS = G::TAPS, U = G::U, SPW = G::SPW;
    constexpr bool DWM = G::DWM;
    constexpr bool GNP = (PRO == PRO_GN || PRO == PRO_GN_SILU || PRO == PRO_GN_DW);
    constexpr bool FILM = (EPI & EPI_FILM) != 0, RES = (EPI & EPI_RES) != 0, SILU = (EPI & EPI_SILU) != 0;
    constexpr int C4 = PC / 4;                          // float4 channel groups per staged pixel
    constexpr int PSTEP = 256 / C4;                     // pixels covered by one pass of the 256 threads
    constexpr int NIT = (LH * LW + PSTEP - 1) / PSTEP;  // staging items per thread and phase
    constexpr int DIT = (TH * TW) / PSTEP;              // depthwise output items per thread and phase
    static_assert(256 % C4 == 0 && (TH * TW) % PSTEP == 0, "staging geometry");
    static_assert((SPW * TAPS) % U == 0, "weight ring depth must divide the steps of a phase");

    DDIF_DYN_SMEM(smem);
    float* As = reinterpret_cast<float*>(smem);
    float* Red = As;  // reused after the last phase (barrier in between)
    float* Hs = As + (G::AFL > G::RFL ? G::AFL : G::RFL);
    float* Sst = Hs + G::HFL;  // [4 waves][2] statistics partials

    const int tid = threadIdx.x, lane = tid & 63;
#ifdef DDIF_EMU
    const int wave = tid >> 6;
#else
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const int h = lane >> 5, j = lane & 31;
    const int tiles = a.tiles_x * a.tiles_y;
    const int nwork = a.B * tiles * a.n_ct;
    const int w0 = (int)((long long)blockIdx.x * nwork / gridDim.x);
    const int w1 = (int)((long long)(blockIdx.x + 1) * nwork / gridDim.x);
    const int Ctot = a.c0 + a.c1;
    // K extent in 16-channel slabs, rounded up so that every wave runs whole groups of U steps in every phase (1x1 convs
    // with a 2-step ring: a multiple of 8).  Slabs past the data stage ZERO activations; their weight reads are clamped to
    // a real slab (finite x 0), so the padding only costs the few MFMAs of a ragged tail.
    constexpr int SQ = 4 * (TAPS == 1 ? U : 1);
    const int NS = (((Ctot + 15) / 16) + SQ - 1) / SQ * SQ;
    const int NSW = a.n_chunks * (KS == 3 ? 1 : 2);  // slabs in the packed weights
    const int NP = (NS + SP - 1) / SP;               // phases
    const int c4 = tid % C4, p0 = tid / C4;
    const float* tbrow = a.tbias + (a.step_ptr ? (size_t)(*a.step_ptr) * a.tb_rowstride : 0);

    // A-fragment base of this lane for accumulator block mb: pixel m = mb*32 + j of the tile (+ tap offset later)
    int abase[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int m = mb * 32 + j;
        abase[mb] = ((m / TW) * IW + (m % TW)) * APIX + 4 * h;
    }

    int gn_b = -1;
    float mean = 0.f, rstd = 1.f;

    for (int work = w0; work < w1; ++work) {
        const int pt = work / a.n_ct, ct = work - pt * a.n_ct;  // cout tile fastest: neighbours re-read the same input from L2
        const int b = pt / tiles, t = pt - b * tiles;
        const int oy0 = (t / a.tiles_x) * TH, ox0 = (t % a.tiles_x) * TW;
        if (GNP && b != gn_b) {  // workgroup-uniform; every wavefront reduces the producer's partials itself
            gn_finalize_wave(a.st0, a.np0, a.st1, a.np1, b, (double)Ctot * a.Hin * a.Win, &mean, &rstd);
            gn_b = b;
        }
        // ---- weight stream of this wave: slabs wave, wave+4, ... ; steps = (slab, tap); ring of U steps in registers
        const char* wbase = reinterpret_cast<const char*>(a.w + (size_t)b * a.w_bstride) + ((size_t)ct * NSW * TAPS) * 3072 + (size_t)lane * 16;
        const int nslab_w = NS > wave ? (NS - wave + 3) / 4 : 0;
        const int nstep_w = nslab_w * TAPS;
        int pf_slab = wave, pf_tap = 0, pf_n = 0;  // prefetch cursor
        float4 wr[U][3];
        auto ring_load = [&](int slot) {
            // past the end the cursor re-reads the last real step (never consumed): a constant number of loads in flight
            const int se = pf_n < nstep_w ? pf_slab : wave;
            const int s = se < NSW ? se : 0;
            const int tp = pf_n < nstep_w ? pf_tap : 0;
            const char* p = wbase + ((size_t)s * TAPS + tp) * 3072;
#pragma unroll
            for (int q = 0; q < 3; ++q) wr[slot][q] = *reinterpret_cast<const float4*>(p + q * 1024);
            ++pf_n;
            if (++pf_tap == TAPS) {
                pf_tap = 0;
                pf_slab += 4;
            }
        };
#pragma unroll
        for (int u = 0; u < U; ++u) ring_load(u);

        f32x16 acc[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;

        int cs_slab = wave;  // consume cursor (slab of the next step to run)
        for (int ph = 0; ph < NP; ++ph) {
            const int cb = ph * PC;
            // ---- stage PC channels of the (haloed) tile: loads first, then prologue + LDS writes
            const int c = cb + c4 * 4;            // first of this thread's 4 channels
            const bool cok = c < Ctot;            // channels past the end stage zeros
            const bool s0 = !cok || c < a.c0;    // (padding channels read source 0, channel 0: always valid)
            const float* src = s0 ? a.in0 + (cok ? c : 0) : a.in1 + (c - a.c0);
            const int cs = s0 ? a.c0 : a.c1;
            float4 sv[NIT];
            [[maybe_unused]] float4 mxv[PRO == PRO_COLSM ? NIT : 1], smv[PRO == PRO_COLSM ? NIT : 1];
            unsigned okm = 0;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int pix = p0 + it * PSTEP;
                const int py = pix / LW, px = pix - py * LW;
                const int iy = oy0 - LP + py, ix = ox0 - LP + px;
                const bool ok = (pix < LH * LW) & (iy >= 0) & (iy < a.Hin) & (ix >= 0) & (ix < a.Win) & cok;
                okm |= (ok ? 1u : 0u) << it;
                const int iyc = iy < 0 ? 0 : (iy >= a.Hin ? a.Hin - 1 : iy), ixc = ix < 0 ? 0 : (ix >= a.Win ? a.Win - 1 : ix);
                const size_t sp = ((size_t)b * a.Hin + iyc) * a.Win + ixc;
                sv[it] = *reinterpret_cast<const float4*>(src + sp * cs);
                if constexpr (PRO == PRO_COLSM) {
                    const size_t so = ((size_t)b * a.Win + ixc) * a.c0 + ((s0 && cok) ? c : 0);
                    mxv[it] = *reinterpret_cast<const float4*>(a.cs_mx + so);
                    smv[it] = *reinterpret_cast<const float4*>(a.cs_sm + so);
                }
            }
            float ga[4] = {1.f, 1.f, 1.f, 1.f}, gb[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (GNP) {
                const float4 gq = *reinterpret_cast<const float4*>(a.gamma + (cok ? c : 0));
                const float4 bq = *reinterpret_cast<const float4*>(a.beta + (cok ? c : 0));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ga[i] = (&gq.x)[i] * rstd;
                    gb[i] = (&bq.x)[i] - mean * ga[i];
                }
            }
            [[maybe_unused]] float4 dwv[DWM ? 9 : 1];
            if constexpr (DWM) {
#pragma unroll
                for (int k = 0; k < 9; ++k) dwv[k] = *reinterpret_cast<const float4*>(a.dw_w + (size_t)k * Ctot + (cok ? c : 0));
            }
            if (ph > 0) __syncthreads();  // every wave is done reading the previous phase's tile
            const bool colsm = (PRO == PRO_COLSM) && s0 && cok;
            const int slab_l = (c4 * 4) / 16, cin_slab = (c4 * 4) % 16;  // slab within the phase, channel within the slab
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int pix = p0 + it * PSTEP;
                const bool ok = (okm >> it) & 1u;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float x = (&sv[it].x)[i];
                    if constexpr (GNP) {
                        x = fmaf(x, ga[i], gb[i]);
                        if constexpr (PRO == PRO_GN_SILU) x = dd_silu(x);
                    }
                    if constexpr (PRO == PRO_COLSM) {
                        if (colsm) x = dd_exp2_fast((x - (&mxv[it].x)[i]) * 1.4426950408889634f) * dd_rcp_fast((&smv[it].x)[i]);
                    }
                    v[i] = ok ? x : 0.f;  // zero padding comes AFTER the activation (and channels past the end are zero)
                }
                if (pix < LH * LW) {
                    if constexpr (DWM) {
                        *reinterpret_cast<float4*>(&Hs[pix * HPIX + c4 * 4]) = make_float4(v[0], v[1], v[2], v[3]);
                        const int py = pix / LW, px = pix - py * LW;
                        // centre pixels inside the image: this IS xn = GroupNorm(cat[h, skip]) (attn_res input); one cout tile writes it
                        if (a.out_xn && ok && ct == 0 && py >= 1 && py <= TH && px >= 1 && px <= TW)
                            *reinterpret_cast<float4*>(a.out_xn + (((size_t)b * a.Hin + oy0 + py - 1) * a.Win + ox0 + px - 1) * Ctot + c) =
                                make_float4(v[0], v[1], v[2], v[3]);
                    } else {
                        unsigned h01, m01, l01, h23, m23, l23;
                        dd_split3_pair(v[0], v[1], &h01, &m01, &l01);
                        dd_split3_pair(v[2], v[3], &h23, &m23, &l23);
                        float* d = &As[pix * APIX + slab_l * 24 + cin_slab / 2];
                        *reinterpret_cast<uint2*>(d) = make_uint2(h01, h23);
                        *reinterpret_cast<uint2*>(d + 8) = make_uint2(m01, m23);
                        *reinterpret_cast<uint2*>(d + 16) = make_uint2(l01, l23);
                    }
                }
            }
            if constexpr (DWM) {
                __syncthreads();  // normalised halo tile complete in Hs
#pragma unroll
                for (int it = 0; it < DIT; ++it) {
                    const int m = p0 + it * PSTEP;
                    const int ty = m / TW, tx = m - ty * TW;
                    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < 9; ++k) {
                        const float4 hv = *reinterpret_cast<const float4*>(&Hs[((ty + k / 3) * LW + tx + k % 3) * HPIX + c4 * 4]);
                        s[0] = fmaf(hv.x, dwv[k].x, s[0]);
                        s[1] = fmaf(hv.y, dwv[k].y, s[1]);
                        s[2] = fmaf(hv.z, dwv[k].z, s[2]);
                        s[3] = fmaf(hv.w, dwv[k].w, s[3]);
                    }
                    if (!cok) s[0] = s[1] = s[2] = s[3] = 0.f;
                    unsigned h01, m01, l01, h23, m23, l23;
                    dd_split3_pair(s[0], s[1], &h01, &m01, &l01);
                    dd_split3_pair(s[2], s[3], &h23, &m23, &l23);
                    float* d = &As[m * APIX + slab_l * 24 + cin_slab / 2];
                    *reinterpret_cast<uint2*>(d) = make_uint2(h01, h23);
                    *reinterpret_cast<uint2*>(d + 8) = make_uint2(m01, m23);
                    *reinterpret_cast<uint2*>(d + 16) = make_uint2(l01, l23);
                }
            }
            __syncthreads();
            // ---- this wave's K steps of the phase: slabs cb/16 + wave, + 4, ...
            const int s_end = (ph + 1) * SP < NS ? (ph + 1) * SP : NS;
            const int ngrp = cs_slab < s_end ? ((s_end - cs_slab + 3) / 4) * TAPS / U : 0;
            int st_tap = 0;
            for (int gI = 0; gI < ngrp; ++gI) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int sl = cs_slab - ph * SP;  // slab within the phase
                    const int aoff = ((st_tap / KS) * IW + (st_tap % KS)) * APIX + sl * 24;
                    float4 xa[MB][3];
#pragma unroll
                    for (int q = 0; q < 3; ++q)
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) xa[mb][q] = *reinterpret_cast<const float4*>(&As[abase[mb] + aoff + q * 8]);
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) {
                        f32x16 cacc = acc[mb];
                        cacc = DDIF_MFMA_32x32x16_BF16(wr[u][2], xa[mb][0], cacc);  // lo * hi
                        cacc = DDIF_MFMA_32x32x16_BF16(wr[u][0], xa[mb][2], cacc);  // hi * lo
                        cacc = DDIF_MFMA_32x32x16_BF16(wr[u][1], xa[mb][1], cacc);  // mid * mid
                        cacc = DDIF_MFMA_32x32x16_BF16(wr[u][1], xa[mb][0], cacc);  // mid * hi
                        cacc = DDIF_MFMA_32x32x16_BF16(wr[u][0], xa[mb][1], cacc);  // hi * mid
                        cacc = DDIF_MFMA_32x32x16_BF16(wr[u][0], xa[mb][0], cacc);  // hi * hi
                        acc[mb] = cacc;
                    }
                    ring_load(u);  // refill the slot just consumed with the step U ahead
                    if (++st_tap == TAPS) {
                        st_tap = 0;
                        cs_slab += 4;
                    }
                }
            }
        }
        // ---- epilogue operands of this wave's quad g = wave: couts ct*32 + 8*wave + 4h .. +3 of pixel (mb, j)
        const int co = ct * 32 + 8 * wave + 4 * h;
        const bool cok_o = co < a.Cout;  // Cout is a multiple of 4
        const int coc = cok_o ? co : 0;
        const float4 bq = *reinterpret_cast<const float4*>(a.bias + coc);
        const float4 tq = *reinterpret_cast<const float4*>(tbrow + (size_t)b * a.tbias_stride + coc);
        [[maybe_unused]] float4 e_res[RES ? MB : 1], e_fs[FILM ? MB : 1], e_fh[FILM ? MB : 1];
        bool pok[MB];
        size_t opix[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int m = mb * 32 + j;
            const int oy = oy0 + m / TW, ox = ox0 + m % TW;
            pok[mb] = (oy < a.Hout) & (ox < a.Wout) & cok_o;
            opix[mb] = pok[mb] ? ((size_t)b * a.Hout + oy) * a.Wout + ox : 0;
            if constexpr (RES) e_res[mb] = *reinterpret_cast<const float4*>(a.res + opix[mb] * a.Cout + coc);
            if constexpr (FILM) {
                e_fs[mb] = *reinterpret_cast<const float4*>(a.film + opix[mb] * 2 * a.Cout + coc);
                e_fh[mb] = *reinterpret_cast<const float4*>(a.film + opix[mb] * 2 * a.Cout + a.Cout + coc);
            }
        }
        __syncthreads();  // the activation tile is dead: its LDS becomes the K-partial exchange
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(&Red[(((wave * MB + mb) * 4 + g) * 64 + lane) * 4]) =
                    make_float4(acc[mb][4 * g + 0], acc[mb][4 * g + 1], acc[mb][4 * g + 2], acc[mb][4 * g + 3]);
        __syncthreads();
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            float4 pp[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) pp[w] = *reinterpret_cast<const float4*>(&Red[(((w * MB + mb) * 4 + wave) * 64 + lane) * 4]);
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float x = ((&pp[0].x)[i] + (&pp[1].x)[i]) + ((&pp[2].x)[i] + (&pp[3].x)[i]);
                x += (&bq.x)[i] + (&tq.x)[i];
                if constexpr (FILM) x = x * (1.f + (&e_fs[mb].x)[i]) + (&e_fh[mb].x)[i];
                if constexpr (SILU) x = dd_silu(x);
                if constexpr (RES) x += (&e_res[mb].x)[i];
                v[i] = x;
            }
            if (pok[mb]) {
                *reinterpret_cast<float4*>(a.out + opix[mb] * a.Cout + co) = make_float4(v[0], v[1], v[2], v[3]);
                s1 += (v[0] + v[1]) + (v[2] + v[3]);
                s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            }
        }
        if (a.st_out) {
            const float t1 = wave_sum_fast(s1), t2 = wave_sum_fast(s2);  // total in lane 63
            if (lane == 63) {
                Sst[wave * 2 + 0] = t1;
                Sst[wave * 2 + 1] = t2;
            }
        }
        __syncthreads();  // also: Red is free again for the next item's tile
        if (a.st_out && tid == 0) {
            const size_t pi = ((size_t)b * (tiles * a.n_ct) + (size_t)t * a.n_ct + ct) * 2;
            a.st_out[pi + 0] = ((double)Sst[0] + (double)Sst[2]) + ((double)Sst[4] + (double)Sst[6]);
            a.st_out[pi + 1] = ((double)Sst[1] + (double)Sst[3]) + ((double)Sst[5] + (double)Sst[7]);
        }
    }
}

}  // namespace ddif
