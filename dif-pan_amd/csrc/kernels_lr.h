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
//       registers.  The ring holds one slab of a 3x3 conv (9 taps x 3 planes = 108 VGPRs) and is filled BEFORE the
//       activation tile is staged, so the weight stream's L2 latency hides behind the staging; each slot is refilled
//       with the step 9 ahead (1.4 us of MFMAs) as soon as its own MFMAs have issued (counted vmcnt, in-order
//       retirement).  A 3-step ring measured 19.8 us per 128->128 conv at 8x8 with 3 us of matrix work (six exposed L2
//       round trips); a whole-phase ring (216 VGPRs) spills;
//     * every other load of a work item (GroupNorm partials, bias / time-bias / residual / FiLM operands of the epilogue,
//       the first phase's activation tile) is issued in the same burst, before anything is waited on: one memory
//       latency per work item instead of four; the next phase's tile is fetched before the current phase's MFMAs;
//     * the activation tile (whole sample at 8x8) is staged ONCE per <=128-channel phase with the producer-side
//       GroupNorm / SiLU / column-softmax prologue applied, as three bf16 planes (the depthwise branch of the q conv is
//       a kernel of its own at these levels, gn_dw3x3_small_kernel: fused here it would be recomputed by every cout tile);
//     * the four K-partials meet in LDS (fixed order: (w0 + w1) + (w2 + w3)) and wave g runs the epilogue of accumulator
//       quad g (bias + time bias, FiLM, SiLU, residual, float4 NHWC stores, GroupNorm partial of the output).
//   Arithmetic is the bf16x3 scheme of kernels_conv.h (operands split hi/mid/lo, six exact bf16 products per fp32
//   product on v_mfma_f32_32x32x16_bf16, small terms first, fp32 accumulate); only the K summation order differs.
#pragma once
#include "kernels_conv.h"

namespace ddif {

// TALL: 16x8 instead of 8x16 pixels for MB = 4 -- whole image columns inside one tile, which is what the column-softmax
// statistics epilogue (EPI_COLST) needs
#ifndef LR_ROWPAD
#define LR_ROWPAD 1
#endif
// (kernels_lr.h K loop) keeps the prefetched LDS reads of the next step in front of the current step's MFMAs: a compiler-level memory fence, no instruction
#if defined(DDIF_EMU) || !defined(__HIP_DEVICE_COMPILE__)
#define LR_PREFETCH_FENCE() ((void)0)
#elif defined(LR_NO_FENCE)
#define LR_PREFETCH_FENCE() ((void)0)
#else
#define LR_PREFETCH_FENCE() asm volatile("" ::: "memory")
#endif
template <int KS, int MB, int PRO, bool TALL = false, bool F16 = false, bool B1 = false>
struct LrGeom {
    static constexpr int NPL = F16 ? 2 : (B1 ? 1 : 3);         // operand planes: f16x2 (hi, lo), bf16x3 (hi, mid, lo) or bf16x1 (the throughput variant)
    static constexpr int TH = (TALL && MB == 4) ? 16 : 8, TW = (MB == 2 || TALL) ? 8 : 16;
    static constexpr int PAD = KS / 2;
    static constexpr int IH = TH + 2 * PAD, IW = TW + 2 * PAD; // staged tile (halo of the 3x3 taps)
    static constexpr int PC = (MB == 4 && KS == 3) ? 64 : 128; // channels staged per phase
    static constexpr int SP = PC / 16;                         // 16-channel slabs per phase
    static constexpr int APIX = SP * NPL * 8 + 4;              // floats per staged pixel: SP x (NPL planes x 32 B) + 16 B pad
    // row pad (round 5): the A-fragment ds_read_b128 of a 32-pixel block is conflict-free when the pixel stride is odd in 16-byte slots (APIX / 4 is: 17, 25, 33,
    // 49) AND the row stride is 8 mod 16 slots for the 8-wide tiles (lanes 0-7 / 8-15 / ... are rows) or 0 mod 16 for the 16-wide ones (bank enumeration
    // over the documented lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}: DESIGN section 3); without it the rows collide two ways
    static constexpr int RSL = (IW * APIX / 4) % 16;
    static constexpr int RPAD = LR_ROWPAD ? (((TW == 8 ? 8 : 0) - RSL + 16) % 16) * 4 : 0;
    static constexpr int AROW = IW * APIX + RPAD;              // floats per staged row
    static constexpr int AFL = IH * AROW;
    static constexpr int RFL = 4 * MB * 4 * 64 * 4;            // K-partials: [wave][mb][quad g][lane] float4
    static constexpr size_t smem = (size_t)((AFL > RFL ? AFL : RFL) + 16) * sizeof(float);
    static constexpr int SPW = SP / 4;                         // slabs per wave and phase
    static constexpr int TAPS = KS * KS;
    static constexpr int U = KS == 3 ? TAPS : SPW;             // weight ring (steps): one slab of a 3x3 conv (9 taps), one phase of a 1x1
};

// ABL (tools/mbench_lr.cpp only): 1 = no weight loads, 2 = no MFMAs, 4 = no activation loads, 8 = no output stores
// F16: the f16x2 split (ddif_dev.h; kernels_conv.h MATH = 3) instead of bf16x3 -- operands pre-scaled by 2^4 / 2^10 and split into two
// half planes, three products per slab and tap, 2/3 of the weight stream and of the staged tile; the accumulator is scaled back in the epilogue
// B1: the throughput variant (kernels_conv.h MATH = 4): one bf16 plane per operand, one product
// ROWS (round 6, 3x3 only): the tile spans the whole image width (Win == TW, one tile column) -- what the 8x8 / 16x16 levels of a 64x64 tile are.  The staging item is
// then a halo ROW and the thread's image column is fixed (PSTEP == TW): no per-item pixel decomposition, clamps or 64-bit index products (the general path spends
// ~25 instructions on each of its 12-13 items before the first load goes out: profiles/r06/lr_stamps.txt), rows outside the image are skipped by a wave-uniform
// branch instead of being loaded, normalised and masked (36 % of the 10 x 10 halo tile of an 8 x 8 sample is padding), and the two halo columns are zeroed once per
// work item.  Same LDS image, same contraction: bit-identical results.
template <int KS, int MB, int PRO, int EPI, int ABL = 0, bool F16 = false, bool B1 = false, bool ROWS = false>
__global__ __launch_bounds__(256) void conv_lr_kernel(ConvArgs a) {
    constexpr bool COLST = (EPI & EPI_COLST) != 0;
    static_assert(!COLST || KS == 1, "column statistics epilogue: 1x1 convs");
    static_assert(!F16 || PRO != PRO_COLSM, "f16x2: the column-softmax prologue stays on bf16x3");
    static_assert(!(F16 && B1), "one operand format");
    using G = LrGeom<KS, MB, PRO, COLST, F16, B1>;
    constexpr int NPL = G::NPL, SLF = NPL * 8;  // floats of one 16-channel slab of a staged pixel
    constexpr int WSTEP = NPL * 1024;           // bytes of one (slab, tap) step of the packed weights
    constexpr int TH = G::TH, TW = G::TW, LP = G::PAD, LH = G::IH, LW = G::IW, IW = G::IW, PC = G::PC, SP = G::SP;
    constexpr int APIX = G::APIX, AROW = G::AROW, TAPS = G::TAPS, U = G::U, SPW = G::SPW;
    constexpr bool GNP = (PRO == PRO_GN || PRO == PRO_GN_SILU);
    static_assert(PRO == PRO_NONE || PRO == PRO_GN || PRO == PRO_GN_SILU || PRO == PRO_COLSM, "prologues of the low-resolution kernel");
    constexpr bool FILM = (EPI & EPI_FILM) != 0, RES = (EPI & EPI_RES) != 0, SILU = (EPI & EPI_SILU) != 0;
    constexpr int C4 = PC / 4;                          // float4 channel groups per staged pixel
    constexpr int PSTEP = 256 / C4;                     // pixels covered by one pass of the 256 threads
    constexpr int NIT = ROWS ? LH : (LH * LW + PSTEP - 1) / PSTEP;  // staging items per thread and phase
    static_assert(256 % C4 == 0, "staging geometry");
    static_assert(!ROWS || (KS == 3 && PSTEP == TW && PRO != PRO_COLSM), "row staging: 3x3 convs, one pass of the threads = one image row");

    dd_touch_kernargs<sizeof(ConvArgs)>();  // every line of the argument block in ONE round trip (ddif_dev.h)
    DDIF_DYN_SMEM(smem);
    float* As = reinterpret_cast<float*>(smem);
    float* Red = As;  // reused after the last phase (barrier in between)
    float* Sst = As + (G::AFL > G::RFL ? G::AFL : G::RFL);  // [4 waves][2] statistics partials

    const int tid = threadIdx.x, lane = tid & 63;
#ifdef DDIF_EMU
    const int wave = tid >> 6;
#else
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const int h = lane >> 5, j = lane & 31;
    const int tiles = a.tiles_x * a.tiles_y;
    const int nwork = a.B * tiles * a.n_ct;
    int w0, w1;
    wg_work_range(nwork, &w0, &w1, a.xcd);
    const int Ctot = a.c0 + a.c1;
    // (scalar copies: with the fields themselves hipcc turned the per-thread select `s0 ? a.c0 : a.c1` below into a select of kernel-argument ADDRESSES and a vector load
    // from the argument segment -- one more serial round trip in front of the first staging load)
#ifdef DDIF_EMU
    const int ac0 = a.c0, ac1 = a.c1;
#else
    const int ac0 = __builtin_amdgcn_readfirstlane(a.c0), ac1 = __builtin_amdgcn_readfirstlane(a.c1);
#endif
    const int NS = (Ctot + 15) / 16;                 // 16-channel slabs (channels past the end of the last one stage zeros)
    const int NSW = a.n_chunks * (KS == 3 ? 1 : 2);  // slabs in the packed weights (>= NS)
    const int NP = (NS + SP - 1) / SP;               // phases
    const int c4 = tid % C4, p0 = tid / C4;
    const int stepv = a.step_ptr ? *a.step_ptr : 0;  // used late (dd_late at the time-bias load below)

    // A-fragment base of this lane for accumulator block mb: pixel m = mb*32 + j of the tile (+ tap offset later)
    int abase[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int m = mb * 32 + j;
        abase[mb] = (m / TW) * AROW + (m % TW) * APIX + 4 * h;
    }

    int gn_b = -1;
    float mean = 0.f, rstd = 1.f;
    constexpr bool RANGE = F16 && !GNP && PRO != PRO_COLSM;  // raw activations x 2^4 into halves: watch the range (ConvArgs::range_flag, kernels_conv.h)
    [[maybe_unused]] float r_max = 0.f;

    for (int work = w0; work < w1; ++work) {
        const int pt = work / a.n_ct, ct = work - pt * a.n_ct;  // cout tile fastest: neighbours re-read the same input from L2
        const int bl = pt / tiles, t = pt - bl * tiles;
        const int b = a.b0 + bl;
        const int oy0 = (t / a.tiles_x) * TH, ox0 = (t % a.tiles_x) * TW;

        // ================= (1) every load of the work item that does not depend on another one, in one burst ==========
        [[maybe_unused]] GnPartials gp;
        const bool new_b = GNP && b != gn_b;  // workgroup-uniform
        if (new_b) gn_load_partials(a.st0, a.np0, a.st1, a.np1, b, &gp);

        // epilogue operands of this wave's quad g = wave: couts ct*32 + 8*wave + 4h .. +3 of pixel (mb, j)
        const int co = ct * 32 + 8 * wave + 4 * h;
        const bool cok_o = co < a.Cout;  // Cout is a multiple of 4
        const int coc = cok_o ? co : 0;
        const float4 bq = *reinterpret_cast<const float4*>(a.bias + coc);
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

        // weight stream of this wave: slabs wave, wave+4, ... ; steps = (slab, tap) in order; ring slot = step within the phase
        const char* wbase = reinterpret_cast<const char*>(a.w + (size_t)b * a.w_bstride) + ((size_t)ct * NSW * TAPS) * WSTEP + (size_t)lane * 16;
        int pf_slab = wave, pf_tap = 0;  // prefetch cursor
        float4 wr[U][NPL];
        auto ring_load = [&](int slot) {
            const int s = pf_slab < NS ? pf_slab : 0;  // (the initial fill of a wave without slabs re-reads slab 0; never consumed)
            const char* p = wbase + ((size_t)s * TAPS + pf_tap) * WSTEP;
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                if (ABL & 1) wr[slot][q] = make_float4(1e-3f * (float)q, 2e-3f, 3e-3f, 4e-3f);
                else wr[slot][q] = *reinterpret_cast<const float4*>(p + q * 1024);
            }
            if (++pf_tap == TAPS) {
                pf_tap = 0;
                pf_slab += 4;
            }
        };
#pragma unroll
        for (int u = 0; u < U; ++u) ring_load(u);

        // activation tile of one phase: loads (stage_load) and prologue + LDS writes (stage_write) are separate so that the
        // next phase's loads fly during the current phase's MFMAs
        float4 sv[NIT];
        [[maybe_unused]] float4 mxv[PRO == PRO_COLSM ? NIT : 1], smv[PRO == PRO_COLSM ? NIT : 1];
        [[maybe_unused]] float4 gq4, bq4;
        unsigned okm = 0;
        auto stage_load = [&](int ph) {
            const int c = ph * PC + c4 * 4;      // first of this thread's 4 channels
            const bool cok = c < Ctot;           // channels past the end stage zeros
            const bool s0 = !cok || c < a.c0;    // (padding channels read source 0, channel 0: always valid)
            const float* src = s0 ? a.in0 + (cok ? c : 0) : a.in1 + (c - a.c0);
            const int cs = s0 ? ac0 : ac1;
            okm = 0;
            if constexpr (ROWS) {
                const float* rp = src + (((size_t)b * a.Hin + oy0) * a.Win + p0) * cs;  // this thread's column of image row oy0
                const ptrdiff_t rs = (ptrdiff_t)a.Win * cs;
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int iy = oy0 - LP + it;
                    if ((iy >= 0) & (iy < a.Hin)) {  // workgroup-uniform
                        okm |= 1u << it;
                        if (ABL & 4) sv[it] = make_float4(0.5f, 0.25f, -0.5f, 0.125f);
                        else sv[it] = *reinterpret_cast<const float4*>(rp + (ptrdiff_t)(it - LP) * rs);
                    }
                }
            } else {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int pix = p0 + it * PSTEP;
                const int py = pix / LW, px = pix - py * LW;
                const int iy = oy0 - LP + py, ix = ox0 - LP + px;
                const bool ok = (pix < LH * LW) & (iy >= 0) & (iy < a.Hin) & (ix >= 0) & (ix < a.Win) & cok;
                okm |= (ok ? 1u : 0u) << it;
                const int iyc = iy < 0 ? 0 : (iy >= a.Hin ? a.Hin - 1 : iy), ixc = ix < 0 ? 0 : (ix >= a.Win ? a.Win - 1 : ix);
                const size_t sp = ((size_t)b * a.Hin + iyc) * a.Win + ixc;
                if (ABL & 4) sv[it] = make_float4(0.5f, 0.25f, -0.5f, 0.125f);
                else sv[it] = *reinterpret_cast<const float4*>(src + sp * cs);
                if constexpr (PRO == PRO_COLSM) {
                    const size_t so = ((size_t)b * a.Win + ixc) * a.c0 + ((s0 && cok) ? c : 0);
                    mxv[it] = *reinterpret_cast<const float4*>(a.cs_mx + so);
                    smv[it] = *reinterpret_cast<const float4*>(a.cs_sm + so);
                }
            }
            }
            if constexpr (GNP) {
                gq4 = *reinterpret_cast<const float4*>(a.gamma + (cok ? c : 0));
                bq4 = *reinterpret_cast<const float4*>(a.beta + (cok ? c : 0));
            }
        };
        auto stage_write = [&](int ph) {
            const int c = ph * PC + c4 * 4;
            const bool cok = c < Ctot;
            const bool colsm = (PRO == PRO_COLSM) && cok && c < a.c0;
            const int slab_l = (c4 * 4) / 16, cin_slab = (c4 * 4) % 16;  // slab within the phase, channel within the slab
            float ga[4] = {1.f, 1.f, 1.f, 1.f}, gb[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (GNP) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ga[i] = (&gq4.x)[i] * rstd;
                    gb[i] = (&bq4.x)[i] - mean * ga[i];
                    if constexpr (F16 && PRO == PRO_GN) {  // no activation behind the normalisation: scale the affine map itself
                        ga[i] *= DDIF_F16_ASCALE;
                        gb[i] *= DDIF_F16_ASCALE;
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int pix = ROWS ? it * LW + p0 + LP : p0 + it * PSTEP;
                if constexpr (ROWS) {
                    if (!((okm >> it) & 1u)) {  // a row of zero padding (workgroup-uniform): nothing was loaded, nothing is normalised
                        float* d = &As[it * AROW + (p0 + LP) * APIX + slab_l * SLF + cin_slab / 2];
#pragma unroll
                        for (int q = 0; q < NPL; ++q) *reinterpret_cast<uint2*>(d + 8 * q) = make_uint2(0u, 0u);
                        continue;
                    }
                }
                const bool ok = ROWS ? cok : (okm >> it) & 1u;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float x = (&sv[it].x)[i];
                    if constexpr (GNP) {
                        x = fmaf(x, ga[i], gb[i]);  // (F16, PRO_GN: the activation scale is folded into ga / gb above)
                        if constexpr (PRO == PRO_GN_SILU) x = F16 ? dd_silu_scaled(x, 1.0f / DDIF_F16_ASCALE) : dd_silu(x);
                    } else if constexpr (F16) {
                        x *= DDIF_F16_ASCALE;
                    }
                    if constexpr (PRO == PRO_COLSM) {
                        if (colsm) x = dd_exp2_fast((x - (&mxv[it].x)[i]) * 1.4426950408889634f) * dd_rcp_fast((&smv[it].x)[i]);
                    }
                    v[i] = ok ? x : 0.f;  // zero padding comes AFTER the activation (and channels past the end are zero)
                }
                if constexpr (RANGE) r_max = fmaxf(fmaxf(r_max, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
                if (ROWS || pix < LH * LW) {
                    float* d = ROWS ? &As[it * AROW + (p0 + LP) * APIX + slab_l * SLF + cin_slab / 2] : &As[(pix / LW) * AROW + (pix % LW) * APIX + slab_l * SLF + cin_slab / 2];
                    if constexpr (B1) {
                        *reinterpret_cast<uint2*>(d) = make_uint2(dd_bf16_pair(v[0], v[1]), dd_bf16_pair(v[2], v[3]));
                    } else if constexpr (F16) {
                        unsigned h01, l01, h23, l23;
                        dd_split2_pair(v[0], v[1], &h01, &l01);
                        dd_split2_pair(v[2], v[3], &h23, &l23);
                        *reinterpret_cast<uint2*>(d) = make_uint2(h01, h23);
                        *reinterpret_cast<uint2*>(d + 8) = make_uint2(l01, l23);
                    } else {
                        unsigned h01, m01, l01, h23, m23, l23;
                        dd_split3_pair(v[0], v[1], &h01, &m01, &l01);
                        dd_split3_pair(v[2], v[3], &h23, &m23, &l23);
                        *reinterpret_cast<uint2*>(d) = make_uint2(h01, h23);
                        *reinterpret_cast<uint2*>(d + 8) = make_uint2(m01, m23);
                        *reinterpret_cast<uint2*>(d + 16) = make_uint2(l01, l23);
                    }
                }
            }
        };
        stage_load(0);
        if constexpr (ROWS) {  // the two halo columns: zero padding for every row and phase of the item (the K-partial exchange of the previous item overwrote them)
            constexpr int Q = (APIX - 4) / 4;  // float4 per staged pixel
            for (int i = tid; i < LH * 2 * Q; i += 256) {
                const int r = i / (2 * Q), e = i - r * (2 * Q);
                *reinterpret_cast<float4*>(&As[r * AROW + (e >= Q ? (LW - 1) * APIX + (e - Q) * 4 : e * 4)]) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        // the time-bias row LAST: its address waits for the step counter (a dependent scalar load); issued earlier, every load behind it in program order
        // would wait for that round trip too
        const float4 tq = *reinterpret_cast<const float4*>(a.tbias + (size_t)dd_late(stepv) * a.tb_rowstride + (size_t)b * a.tbias_stride + coc);

        // ================= (2) first uses =================
        if (new_b) {  // every wavefront reduces the producer's partials itself (no barrier)
            gn_reduce_partials(gp, a.st0, a.np0, a.st1, a.np1, b, (double)Ctot * a.Hin * a.Win, &mean, &rstd);
            gn_b = b;
        }
        f32x16 acc[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;

        for (int ph = 0; ph < NP; ++ph) {
            if (ph > 0) __syncthreads();  // every wave is done reading the previous phase's tile
            stage_write(ph);
            if (ph + 1 < NP) stage_load(ph + 1);  // in flight during this phase's MFMAs
            __syncthreads();
            // ---- this wave's K steps of the phase: slabs ph*SP + wave, + 4, ... (ring slot = tap for 3x3, slab-in-phase for 1x1)
            // (round 6) The A fragments of step n + 1 are read from LDS BEFORE the MFMAs of step n are issued (two register sets), and the products of a step go out
            // product-major over the accumulator blocks: two consecutive MFMAs never share an accumulator, so the waits hipcc puts between them cost an issue slot
            // and not the ~43-cycle same-accumulator cliff (guide: MI355X_MICROARCH constants table).  Per accumulator the order of the products is unchanged: same bits.
            const int s_end = (ph + 1) * SP < NS ? (ph + 1) * SP : NS;
            float4 xa[2][MB][NPL];
            auto a_load = [&](int buf, int sl, int tap) {
                const int aoff = (tap / KS) * AROW + (tap % KS) * APIX + sl * SLF;
#pragma unroll
                for (int q = 0; q < NPL; ++q)
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) xa[buf][mb][q] = *reinterpret_cast<const float4*>(&As[abase[mb] + aoff + q * 8]);
            };
#pragma unroll
            for (int k = 0; k < SPW; ++k) {
                const int sl = wave + 4 * k;  // slab within the phase
                if (ph * SP + sl < s_end) {   // wave-uniform; only the last phase can be ragged
                    a_load(0, sl, 0);         // (one exposed LDS latency per slab; the taps behind it are prefetched)
#pragma unroll
                    for (int tap = 0; tap < TAPS; ++tap) {
                        constexpr int UU = U;
                        const int u = (k * TAPS + tap) % UU;
                        const int cur = tap & 1;
                        if (tap + 1 < TAPS) a_load(cur ^ 1, sl, tap + 1);
                        LR_PREFETCH_FENCE();
                        if (ABL & 2) {
#pragma unroll
                            for (int mb = 0; mb < MB; ++mb) acc[mb][0] += wr[u][0].x * xa[cur][mb][0].x + wr[u][NPL / 2].y * xa[cur][mb][NPL / 2].y + wr[u][NPL - 1].z * xa[cur][mb][NPL - 1].z;
                        } else if constexpr (B1) {
#pragma unroll
                            for (int mb = 0; mb < MB; ++mb) acc[mb] = DDIF_MFMA_32x32x16_BF16(wr[u][0], xa[cur][mb][0], acc[mb]);
                        } else if constexpr (F16) {
#pragma unroll
                            for (int mb = 0; mb < MB; ++mb) acc[mb] = DDIF_MFMA_32x32x16_F16(wr[u][NPL - 1], xa[cur][mb][0], acc[mb]);  // lo * hi
#pragma unroll
                            for (int mb = 0; mb < MB; ++mb) acc[mb] = DDIF_MFMA_32x32x16_F16(wr[u][0], xa[cur][mb][NPL - 1], acc[mb]);  // hi * lo
#pragma unroll
                            for (int mb = 0; mb < MB; ++mb) acc[mb] = DDIF_MFMA_32x32x16_F16(wr[u][0], xa[cur][mb][0], acc[mb]);        // hi * hi
                        } else if constexpr (NPL == 3) {
#pragma unroll
                            for (int mb = 0; mb < MB; ++mb) acc[mb] = DDIF_MFMA_32x32x16_BF16(wr[u][2], xa[cur][mb][0], acc[mb]);  // lo * hi
#pragma unroll
                            for (int mb = 0; mb < MB; ++mb) acc[mb] = DDIF_MFMA_32x32x16_BF16(wr[u][0], xa[cur][mb][2], acc[mb]);  // hi * lo
#pragma unroll
                            for (int mb = 0; mb < MB; ++mb) acc[mb] = DDIF_MFMA_32x32x16_BF16(wr[u][1], xa[cur][mb][1], acc[mb]);  // mid * mid
#pragma unroll
                            for (int mb = 0; mb < MB; ++mb) acc[mb] = DDIF_MFMA_32x32x16_BF16(wr[u][1], xa[cur][mb][0], acc[mb]);  // mid * hi
#pragma unroll
                            for (int mb = 0; mb < MB; ++mb) acc[mb] = DDIF_MFMA_32x32x16_BF16(wr[u][0], xa[cur][mb][1], acc[mb]);  // hi * mid
#pragma unroll
                            for (int mb = 0; mb < MB; ++mb) acc[mb] = DDIF_MFMA_32x32x16_BF16(wr[u][0], xa[cur][mb][0], acc[mb]);  // hi * hi
                        }
                        if (pf_slab < NS) ring_load(u);  // the step U ahead of this one (wave-uniform branch)
                        // NOTE: do not wrap the MFMA group in __builtin_amdgcn_sched_barrier here.  With fences before / after it
                        // this kernel produced wrong results on gfx950 (3e-2 errors in every -m gpu parity test, deterministic;
                        // the host emulator cannot see it).  The unfenced build is the one the parity suite -- including the
                        // bit-equality tests at B = 64 -- verifies.
                    }
                }
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
        [[maybe_unused]] float colv[COLST ? MB : 1][4];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            float4 pp[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) pp[w] = *reinterpret_cast<const float4*>(&Red[(((w * MB + mb) * 4 + wave) * 64 + lane) * 4]);
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float x = ((&pp[0].x)[i] + (&pp[1].x)[i]) + ((&pp[2].x)[i] + (&pp[3].x)[i]);
                x = F16 ? fmaf(x, DDIF_F16_OSCALE, (&bq.x)[i] + (&tq.x)[i]) : x + ((&bq.x)[i] + (&tq.x)[i]);
                if constexpr (FILM) x = x * (1.f + (&e_fs[mb].x)[i]) + (&e_fh[mb].x)[i];
                if constexpr (SILU) x = dd_silu(x);
                if constexpr (RES) x += (&e_res[mb].x)[i];
                v[i] = x;
            }
            if constexpr (COLST) {
#pragma unroll
                for (int i = 0; i < 4; ++i) colv[mb][i] = pok[mb] ? v[i] : -INFINITY;
            }
            if (pok[mb]) {
                if (!(ABL & 8) || v[0] == 12345.678f) *reinterpret_cast<float4*>(a.out + opix[mb] * a.Cout + co) = make_float4(v[0], v[1], v[2], v[3]);
                s1 += (v[0] + v[1]) + (v[2] + v[3]);
                s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            }
        }
        if constexpr (COLST) {
            // softmax_H statistics of the output (q.softmax(dim=-2), sr3_dwt.py:545): the tile holds whole columns (TW = 8:
            // pixel m = 8 y + x, lane j <-> rows j/8 + 4 mb of column j%8), so max / sum(exp(. - max)) over H are an in-register
            // reduction over mb and two xor-shuffles (8, 16) inside the 32-lane half; lanes j < 8 write [b][x][cout quad]
            float cm[4], cs[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float m = colv[0][i];
#pragma unroll
                for (int mb = 1; mb < MB; ++mb) m = fmaxf(m, colv[mb][i]);
                m = fmaxf(m, __shfl_xor(m, 8));
                m = fmaxf(m, __shfl_xor(m, 16));
                float e = 0.f;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) e += dd_exp(colv[mb][i] - m);  // rows outside the image: exp(-inf) = 0
                e += __shfl_xor(e, 8);
                e += __shfl_xor(e, 16);
                cm[i] = m;
                cs[i] = e;
            }
            const int ox = ox0 + (j & 7);
            if (j < 8 && ox < a.Wout && cok_o) {
                const size_t so = ((size_t)b * a.Wout + ox) * a.Cout + co;
                *reinterpret_cast<float4*>(a.cso_mx + so) = make_float4(cm[0], cm[1], cm[2], cm[3]);
                *reinterpret_cast<float4*>(a.cso_sm + so) = make_float4(cs[0], cs[1], cs[2], cs[3]);
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
    if constexpr (RANGE) {
        if (a.range_flag && !(r_max < 65520.f)) *a.range_flag = 1;
    }
}

}  // namespace ddif
