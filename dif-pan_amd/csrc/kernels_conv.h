// Implicit-GEMM convolution (3x3 / 1x1, NHWC fp32) for gfx950.
//
// Replaces what the reference runs as torch `nn.Conv2d` + the elementwise ops around it: models/sr3_dwt.py:288-300
// (Block: GroupNorm -> Swish -> conv3x3), :303-327 (ResnetBlock: + time bias, + residual), :266-282 (Upsample /
// Downsample), :376-396 (CondInjection x_conv + FiLM), :528-533 (FFN convs), :338-339 (attention 1x1s).
//
// One workgroup = 4 wavefronts computes a TH x TW pixel tile x (32*NB*WN) output channels of one tile (sample):
//   M = pixels, N = output channels, K = taps x input channels.
//   * the input halo tile is staged ONCE per CK-channel chunk into LDS, with the producer-side GroupNorm(1 group)
//     + SiLU prologue applied while staging (zero padding is applied after the activation, as in the reference);
//     two input tensors can be concatenated along channels without materialising the concat (skip connections,
//     self-conditioning cat[x, x]); nearest x2 upsampling and stride 2 are index maps of the staging pass;
//   * A fragments come from LDS as ds_read_b128 (row stride CK+4 floats); the weight chunk (pre-packed on the host in
//     exactly the per-lane B-fragment order) is staged through LDS as well, one conflict-free 1 KiB ds_read_b128 per
//     wave and (tap, k8) step -- so the MFMA loop issues NO global load: vmcnt retires in order, and a B fragment
//     fetched from L2 inside the loop would make its first use wait for the older HBM tile prefetch too;
//   * the contraction runs on v_mfma_f32_32x32x2_f32 (exact fp32, bitwise an fmaf chain: no precision change vs.
//     an FMA loop, but 64 FLOP/clk/SIMD from one wave);
//   * epilogue: + bias, + per-sample time bias, FiLM (1+scale)*y+shift, SiLU, + residual, store, and the
//     per-(tile, cout-tile) {sum, sum^2} partials the NEXT GroupNorm needs (fp64, deterministic order, no atomics).
//   * PERSISTENT + SOFTWARE-PIPELINED: a launch is 2-3 workgroups per CU; each workgroup walks a contiguous range of
//     (cout-tile, pixel-tile) work items x channel chunks.  While the MFMAs of item i run out of LDS buffer i&1, the
//     global loads of item i+1 are in flight into registers; afterwards the prologue is applied and buffer (i+1)&1 is
//     written (one barrier per item).  Without this every workgroup of a launch ran load / prologue / MFMA / store in
//     lockstep (rocprof: waves alive 40 us for 4 us of MFMA work, 38 % MfmaUtil).
#pragma once
#include "ddif_dev.h"

namespace ddif {

struct ConvArgs {
    const float* in0;
    const float* in1;
    int c0, c1;              // channels of the two concatenated sources (c1 = 0: single source)
    int B, Hin, Win;         // source spatial size
    int Hout, Wout, Cout;
    const float* w;          // packed weights, see pack_conv_weights()
    int n_chunks;            // ceil((c0 + c1) / CK)
    const float* bias;       // [Cout] or null
    const float* tbias;      // time bias rows or null; row of sample b = tbias + step * tb_rowstride + b * tbias_stride
    int tbias_stride;
    const int* step_ptr;     // device step counter of the running sampler (null: step = 0)
    int tb_rowstride;
    const double* st0;       // GroupNorm partials of in0 / in1 (prologue), [B][np][2]
    int np0;
    const double* st1;
    int np1;
    const float* gamma;      // [c0 + c1]
    const float* beta;
    const float* res;        // residual, same shape as out, or null
    const float* film;       // [B, Hout, Wout, 2*Cout]: scale | shift, or null
    int act_silu;            // SiLU on the output
    float* out;
    double* st_out;          // partials of out [B][tiles_x*tiles_y*gridDim.y][2] or null
    int tiles_x, tiles_y;
    int n_ct;                // number of cout tiles (work item = cout tile x pixel tile; partial index uses it)
    long long w_bstride;     // floats between the packed weights of consecutive samples (0: shared weights)
    const float* cs_mx;      // PRO_COLSM: column-softmax max / sum of source 0, [B][Win][c0]
    const float* cs_sm;
    const float* dw_w;       // PRO_GN_DW: depthwise 3x3 weights [9][c0 + c1] applied to the normalised input
    float* out_xn;           // PRO_GN_DW: the normalised input itself, [B,H,W,c0+c1] (consumed by attn_res) or null
    long long* dbg;          // microbenchmark instrumentation (ABL & 16) only
};


// ABL (microbenchmark ablations only, tools/mbench.cpp): 1 = no MFMA, 2 = no input loads, 4 = no stores, 8 = no weight loads
// GROUPS = 2: one 512-thread workgroup = two 4-wave groups, each with its own work range, LDS buffers and pipeline,
// running in ANTI-PHASE: while group 0 issues its MFMAs (phase H1) group 1 does its epilogue + LDS staging (phase H2),
// swapped at every barrier.  rocprof showed why this is needed: two independent workgroups per CU drift into lockstep,
// both waves of a SIMD queue on the one MFMA pipe (SQ_WAIT_INST_ANY = 2 x the MFMA time) and then both leave it idle
// while they stage.
template <int KS, int STRIDE, int UPS, int TH, int TW, int CK, int WM, int WN, int MB, int NB, int PRO, int VEC, int GROUPS = 1, int ABL = 0>
__global__ __launch_bounds__(256 * GROUPS) void conv_mfma_kernel(ConvArgs a) {
    constexpr int PAD = KS / 2;
    constexpr int IH = (TH - 1) * STRIDE + KS, IW = (TW - 1) * STRIDE + KS;
    constexpr int LDA = CK + 4;
    constexpr int TAPS = KS * KS;
    constexpr int K8 = CK / 8;
    constexpr int C4 = CK / 4;
    constexpr int NF = TAPS * K8;                         // (tap, k8) steps per chunk
    constexpr int ABUF = IH * IW * LDA;                   // floats per LDS buffer
    // PRO_GN_DW (1x1 conv over depthwise3x3(GroupNorm(x))): the LOAD tile has a one-pixel halo and goes to a scratch
    // LDS region; the depthwise conv turns it into the A tile of the 1x1 contraction.
    constexpr bool DWM = (PRO == PRO_GN_DW);
    static_assert(!DWM || (KS == 1 && STRIDE == 1 && !UPS && VEC), "depthwise staging is for plain 1x1 convs");
    constexpr int LPAD = DWM ? 1 : PAD;
    constexpr int LH = DWM ? TH + 2 : IH, LW = DWM ? TW + 2 : IW;   // extent of the loaded tile
    constexpr int HBUF = DWM ? LH * LW * LDA : 0;                  // scratch for the normalised halo tile
    constexpr int DWMAX = DWM ? 9 * 256 : 0;                       // depthwise weights of up to 256 channels
    constexpr int DITEMS = (TH * TW * C4 + 255) / 256;
    constexpr int NITEMS = (LH * LW * C4 + 255) / 256;    // float4 input-staging items per thread and chunk
    constexpr int WBUF = NB * WN * NF * 256;              // floats of one weight chunk (all n-blocks of the cout tile)
    constexpr int WITEMS = (WBUF / 4 + 255) / 256;        // float4 weight-staging items per thread and chunk
    static_assert(WM * WN == 4, "4 wavefronts per workgroup");
    static_assert(TH * TW == 32 * MB * WM, "pixel tile must match the wave layout");
    static_assert(CK % 8 == 0 && 256 % C4 == 0, "chunk size");
    static_assert(NITEMS <= 32, "valid mask is 32 bits");

    DDIF_DYN_SMEM(smem_all);
    constexpr size_t GSZ = (size_t)(2 * (ABUF + WBUF) + HBUF + DWMAX) * sizeof(float) + 16 * sizeof(double);  // LDS bytes per group
    const int grp = GROUPS == 1 ? 0 : (int)(threadIdx.x >> 8);
    char* smem = smem_all + grp * GSZ;
    float* As = reinterpret_cast<float*>(smem);   // [2][ABUF]  input halo tile of one channel chunk
    float* Ws = As + 2 * ABUF;                    // [2][WBUF]  weight chunk in B-fragment order
    double* red = reinterpret_cast<double*>(smem + (size_t)2 * (ABUF + WBUF) * sizeof(float));  // [2][8]
    float* Hs = reinterpret_cast<float*>(smem + (size_t)2 * (ABUF + WBUF) * sizeof(float) + 16 * sizeof(double));  // [HBUF]
    float* DWs = Hs + HBUF;                                                                                      // [9][Ctot]

    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;  // group-local thread / wave index
    const int wm = wave / WN, wn = wave % WN;
    const int h = lane >> 5, j = lane & 31;
    const int tiles = a.tiles_x * a.tiles_y;
    const int ntiles = a.B * tiles;
    const int nwork = ntiles * a.n_ct;
    const int nvb = gridDim.x * GROUPS;  // virtual workgroups (one per 4-wave group)
    const int vb = blockIdx.x * GROUPS + grp;
    const int w0 = (int)((long long)vb * nwork / nvb);
    const int w1 = (int)((long long)(vb + 1) * nwork / nvb);
    if (GROUPS == 1 && w0 >= w1) return;  // whole workgroup leaves together (a group of a pair must keep its barriers)
    const int Hc = UPS ? a.Hin * 2 : a.Hin, Wc = UPS ? a.Win * 2 : a.Win;
    const int Ctot = a.c0 + a.c1;
    const int c4 = tid % C4;

    int abase[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int m = (wm * MB + mb) * 32 + j;
        abase[mb] = ((m / TW) * STRIDE * IW + (m % TW) * STRIDE) * LDA + 4 * h;
    }
    constexpr int WCHUNK = NF * 256;  // floats per (n-block, chunk)

    // ---- per-thread staging geometry: constant for the whole kernel (no divisions inside the stage loop) ----
    int a_py[NITEMS], a_px[NITEMS], a_lds[NITEMS];
    unsigned a_in = 0;
#pragma unroll
    for (int it = 0; it < NITEMS; ++it) {
        const int pixr = (tid + it * 256) / C4;
        const bool in = pixr < LH * LW;
        const int pix = in ? pixr : LH * LW - 1;
        a_py[it] = pix / LW;
        a_px[it] = pix % LW;
        a_lds[it] = pix * LDA + c4 * 4;
        a_in |= (in ? 1u : 0u) << it;
    }
    int w_goff[WITEMS], w_lds[WITEMS];
    unsigned w_in = 0;
#pragma unroll
    for (int it = 0; it < WITEMS; ++it) {
        const int qr = tid + it * 256;
        const bool in = qr < WBUF / 4;
        const int q = in ? qr : WBUF / 4 - 1;
        w_goff[it] = (q / (NF * 64)) * (a.n_chunks * WCHUNK) + (q % (NF * 64)) * 4;
        w_lds[it] = q * 4;
        w_in |= (in ? 1u : 0u) << it;
    }

    // ---- work-item positions (workgroup-uniform); one division set per ITEM, none per chunk ----
    struct Pos { int work, ct, b, oy0, ox0; };
    auto locate = [&](int work) {
        Pos p;
        p.work = work;
        p.ct = work / ntiles;
        const int pt = work - p.ct * ntiles;
        p.b = pt / tiles;
        const int t = pt - p.b * tiles;
        const int ty = t / a.tiles_x;
        p.oy0 = ty * TH;
        p.ox0 = (t - ty * a.tiles_x) * TW;
        return p;
    };

    auto next_pos = [&](Pos p) {  // work + 1 without divisions
        p.work += 1;
        p.ox0 += TW;
        if (p.ox0 >= a.tiles_x * TW) {
            p.ox0 = 0;
            p.oy0 += TH;
            if (p.oy0 >= a.tiles_y * TH) {
                p.oy0 = 0;
                if (++p.b == a.B) {
                    p.b = 0;
                    ++p.ct;
                }
            }
        }
        return p;
    };

    // ---- loading side: runs TWO stages (channel chunks) ahead of the MFMAs, into two register sets ----
    // Measured on MI355X (tools/mbench.cpp, in-kernel clock stamps): with all workgroups of a launch prefetching in
    // bursts a global load takes ~5 us to land, a stage of MFMAs ~2-3 us.  vmcnt retires IN ORDER, so (a) nothing that
    // is consumed inside a stage may be loaded after a prefetch (epilogue operands are issued first), (b) one stage
    // of lookahead is not enough.
    Pos L = locate(w0);
    int l_ch = 0;
    int l_sp[NITEMS];   // clamped source pixel index of every staging item of work item L
    int l_cs[NITEMS];   // PRO_COLSM: (b * Win + x) of the item, index into the column-softmax statistics
    unsigned l_ok = 0;  // which of them are real (inside the image): zero padding otherwise
    auto item_geometry = [&]() {
        const int iy0 = L.oy0 * STRIDE - LPAD, ix0 = L.ox0 * STRIDE - LPAD;
        l_ok = 0;
#pragma unroll
        for (int it = 0; it < NITEMS; ++it) {
            const int iy = iy0 + a_py[it], ix = ix0 + a_px[it];
            const bool ok = (iy >= 0) & (iy < Hc) & (ix >= 0) & (ix < Wc);
            l_ok |= (ok ? 1u : 0u) << it;
            const int iyc = iy < 0 ? 0 : (iy >= Hc ? Hc - 1 : iy), ixc = ix < 0 ? 0 : (ix >= Wc ? Wc - 1 : ix);
            l_sp[it] = (L.b * a.Hin + (UPS ? (iyc >> 1) : iyc)) * a.Win + (UPS ? (ixc >> 1) : ixc);
            if (PRO == PRO_COLSM) l_cs[it] = L.b * a.Win + ixc;
        }
        l_ok &= a_in;
    };

    struct StageRegs {
        float4 sv[NITEMS], wv[WITEMS];
        float4 mxv[PRO == PRO_COLSM ? NITEMS : 1], smv[PRO == PRO_COLSM ? NITEMS : 1];
        float gq[4], bq[4];
        unsigned ok;
        int cbase, ch;
        Pos pos;
    };
    StageRegs R0, R1;
    Pos bufpos[2];  // work item staged in LDS buffer 0 / 1
    int bufch[2] = {0, 0};
    int gn_b = -1;
    float mean = 0.f, rstd = 1.f;

    // Branch-free: every item loads from a CLAMPED (always valid) address; validity is a bit mask applied when the
    // tile is written to LDS (with divergent bounds / source branches hipcc waits vmcnt(0) after every load).
    auto issue_loads = [&](StageRegs& R) {
        const int cbase = l_ch * CK + c4 * 4;
        if (ABL & 2) {
#pragma unroll
            for (int it = 0; it < NITEMS; ++it) R.sv[it] = make_float4(0.5f, 0.25f, -0.5f, 0.125f);
        } else if (VEC) {
            const int cb = cbase < Ctot ? cbase : Ctot - 4;
            const bool s0 = cb < a.c0;
            const float* base = s0 ? a.in0 + cb : a.in1 + (cb - a.c0);
            const int cs = s0 ? a.c0 : a.c1;
#pragma unroll
            for (int it = 0; it < NITEMS; ++it) R.sv[it] = *reinterpret_cast<const float4*>(base + (size_t)l_sp[it] * cs);
        } else {
#pragma unroll
            for (int it = 0; it < NITEMS; ++it) {
                float e[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = cbase + i < Ctot ? cbase + i : Ctot - 1;
                    const bool s0 = c < a.c0;
                    const float* base = s0 ? a.in0 + c : a.in1 + (c - a.c0);
                    e[i] = base[(size_t)l_sp[it] * (s0 ? a.c0 : a.c1)];
                }
                R.sv[it] = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
        if (PRO == PRO_COLSM) {  // softmax_H(q) statistics of the channels of source 0 (c0 is a multiple of CK)
            const int cb = cbase < a.c0 ? cbase : 0;
#pragma unroll
            for (int it = 0; it < NITEMS; ++it) {
                R.mxv[it] = *reinterpret_cast<const float4*>(a.cs_mx + (size_t)l_cs[it] * a.c0 + cb);
                R.smv[it] = *reinterpret_cast<const float4*>(a.cs_sm + (size_t)l_cs[it] * a.c0 + cb);
            }
        }
        const float* wbase = a.w + (size_t)L.b * a.w_bstride + ((size_t)L.ct * (NB * WN) * a.n_chunks + l_ch) * WCHUNK;
#pragma unroll
        for (int it = 0; it < WITEMS; ++it) {
            if (ABL & 8) R.wv[it] = make_float4(0.01f, 0.02f, 0.03f, 0.04f);
            else R.wv[it] = *reinterpret_cast<const float4*>(wbase + w_goff[it]);
        }
        if (PRO == PRO_GN || PRO == PRO_GN_SILU || PRO == PRO_GN_DW) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = cbase + i < Ctot ? cbase + i : Ctot - 1;
                R.gq[i] = a.gamma[c];
                R.bq[i] = a.beta[c];
            }
        }
        R.ok = l_ok;
        R.cbase = cbase;
        R.ch = l_ch;
        R.pos = L;
        // advance the loader
        if (++l_ch == a.n_chunks) {
            l_ch = 0;
            if (L.work + 1 < w1) {
                L = next_pos(L);
                item_geometry();
            }
        }
    };
    auto finish_stage = [&](StageRegs& R, int buf) {
        float* dst = As + buf * ABUF;
        float* wdst = Ws + buf * WBUF;
        float ga[4] = {0.f, 0.f, 0.f, 0.f}, gb[4] = {0.f, 0.f, 0.f, 0.f};
        if (PRO == PRO_GN || PRO == PRO_GN_SILU || PRO == PRO_GN_DW) {
            if (R.pos.b != gn_b) {  // workgroup-uniform; every wavefront reduces the partials itself (no barrier)
                gn_finalize_wave(a.st0, a.np0, a.st1, a.np1, R.pos.b, (double)Ctot * a.Hin * a.Win, &mean, &rstd);
                gn_b = R.pos.b;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ga[i] = R.gq[i] * rstd;
                gb[i] = R.bq[i] - mean * ga[i];
            }
        }
        bool cok[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) cok[i] = R.cbase + i < Ctot;
#pragma unroll
        for (int it = 0; it < NITEMS; ++it) {
            const bool ok = (R.ok >> it) & 1u;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float x = (&R.sv[it].x)[i];
                if (PRO == PRO_GN || PRO == PRO_GN_SILU || PRO == PRO_GN_DW) {
                    x = fmaf(x, ga[i], gb[i]);
                    if (PRO == PRO_GN_SILU) x = dd_silu(x);
                }
                if (PRO == PRO_COLSM) {
                    if (R.cbase < a.c0) x = dd_exp2_fast((x - (&R.mxv[it].x)[i]) * 1.4426950408889634f) * dd_rcp_fast((&R.smv[it].x)[i]);
                }
                v[i] = (ok && cok[i]) ? x : 0.f;  // zero padding comes AFTER the activation
            }
            if (DWM) {
                if ((a_in >> it) & 1u) {
                    *reinterpret_cast<float4*>(&Hs[a_lds[it]]) = make_float4(v[0], v[1], v[2], v[3]);
                    // centre pixels inside the image: this IS xn = GroupNorm(cat[h, skip]) (attn_res input)
                    if (a.out_xn && ok && cok[0] && a_py[it] >= 1 && a_py[it] <= TH && a_px[it] >= 1 && a_px[it] <= TW)
                        *reinterpret_cast<float4*>(a.out_xn + ((size_t)((R.pos.b * a.Hin + R.pos.oy0 + a_py[it] - 1) * a.Win + R.pos.ox0 + a_px[it] - 1)) * Ctot + R.cbase) =
                            make_float4(v[0], v[1], v[2], v[3]);
                }
            } else {
                if ((a_in >> it) & 1u) *reinterpret_cast<float4*>(&dst[a_lds[it]]) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        if (DWM) {
            __syncthreads();  // halo tile complete in Hs
            const int cb = R.cbase < Ctot ? R.cbase : 0;
#pragma unroll
            for (int it = 0; it < DITEMS; ++it) {
                const int item = tid + it * 256;
                const int p = item / C4;
                if (p < TH * TW) {
                    const int ty = p / TW, tx = p % TW;
                    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
                    for (int k = 0; k < 9; ++k) {
                        const float4 hv = *reinterpret_cast<const float4*>(&Hs[((ty + k / 3) * LW + tx + k % 3) * LDA + c4 * 4]);
                        const float4 wk = *reinterpret_cast<const float4*>(&DWs[k * Ctot + cb]);
                        s0 = fmaf(hv.x, wk.x, s0);
                        s1 = fmaf(hv.y, wk.y, s1);
                        s2 = fmaf(hv.z, wk.z, s2);
                        s3 = fmaf(hv.w, wk.w, s3);
                    }
                    *reinterpret_cast<float4*>(&dst[p * LDA + c4 * 4]) = make_float4(s0, s1, s2, s3);
                }
            }
        }
#pragma unroll
        for (int it = 0; it < WITEMS; ++it)
            if ((w_in >> it) & 1u) *reinterpret_cast<float4*>(&wdst[w_lds[it]]) = R.wv[it];
        bufpos[buf] = R.pos;
        bufch[buf] = R.ch;
    };

    // ---- the same staging work as finish_stage(), cut into pieces that h1 interleaves BETWEEN the MFMA groups of the
    //      running stage (MFMA issue is asynchronous: the wave keeps issuing these VALU / ds_write instructions while
    //      the matrix pipe works), so a wave overlaps its own staging with its own MFMAs.  Not used for PRO_GN_DW.
    float pga[4], pgb[4];
    bool pcok[4];
    auto finish_prepare = [&](StageRegs& R) {
        if (PRO == PRO_GN || PRO == PRO_GN_SILU) {
            if (R.pos.b != gn_b) {
                gn_finalize_wave(a.st0, a.np0, a.st1, a.np1, R.pos.b, (double)Ctot * a.Hin * a.Win, &mean, &rstd);
                gn_b = R.pos.b;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                pga[i] = R.gq[i] * rstd;
                pgb[i] = R.bq[i] - mean * pga[i];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) pcok[i] = R.cbase + i < Ctot;
    };
    auto finish_piece = [&](StageRegs& R, int buf, int piece) {  // piece: 0..NITEMS-1 input items, then weight items
        if (piece < NITEMS) {
            const int it = piece;
            const bool ok = (R.ok >> it) & 1u;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float x = (&R.sv[it].x)[i];
                if (PRO == PRO_GN || PRO == PRO_GN_SILU) {
                    x = fmaf(x, pga[i], pgb[i]);
                    if (PRO == PRO_GN_SILU) x = dd_silu(x);
                }
                if (PRO == PRO_COLSM) {
                    if (R.cbase < a.c0) x = dd_exp2_fast((x - (&R.mxv[it].x)[i]) * 1.4426950408889634f) * dd_rcp_fast((&R.smv[it].x)[i]);
                }
                v[i] = (ok && pcok[i]) ? x : 0.f;
            }
            if ((a_in >> it) & 1u) *reinterpret_cast<float4*>(&As[buf * ABUF + a_lds[it]]) = make_float4(v[0], v[1], v[2], v[3]);
        } else if (piece < NITEMS + WITEMS) {
            const int it = piece - NITEMS;
            if ((w_in >> it) & 1u) *reinterpret_cast<float4*>(&Ws[buf * WBUF + w_lds[it]]) = R.wv[it];
        }
    };
    auto finish_commit = [&](StageRegs& R, int buf) {
        bufpos[buf] = R.pos;
        bufch[buf] = R.ch;
    };

    bool pend = false;  // a statistics partial of work item pend_pos sits in red[pend_par]
    Pos pend_pos = L;
    int pend_par = 0;
    auto flush_stats = [&]() {
        if (a.st_out && pend && tid == 0) {
            const Pos p = pend_pos;
            const int t = (p.oy0 / TH) * a.tiles_x + p.ox0 / TW;
            const double* r = red + pend_par * 8;
            const size_t pi = ((size_t)p.b * (tiles * a.n_ct) + (size_t)t * a.n_ct + p.ct) * 2;
            a.st_out[pi + 0] = (r[0] + r[2]) + (r[4] + r[6]);
            a.st_out[pi + 1] = (r[1] + r[3]) + (r[5] + r[7]);
        }
        pend = false;
    };

    int dbg_n = 0;
    auto stamp = [&]() {
        if ((ABL & 16) && a.dbg && tid == 0 && dbg_n < 120) a.dbg[vb * 128 + dbg_n++] = (long long)wall_clock64();
    };
    stamp();
    f32x16 acc[MB][NB];
    const int nflat = (w1 - w0) * a.n_chunks;

    // One pipeline step of stage `flat`, in two halves:
    //   H1: epilogue-operand loads, prefetch of stage flat+2 into Rf, the MFMAs out of LDS buffer `cur`;
    //   H2: epilogue (last chunk of an item), then stage flat+1 (loaded one step ago, Rn) -> LDS buffer cur^1.
    Pos Cp = L;
    int c_ch = 0, nbg0 = 0;
    bool last = false;
    float e_bias[NB], e_tb[NB];
    float e_res[MB][NB][16];
    constexpr bool INTERLEAVE = false;  // measured slower (the pieces drag vmcnt/lgkmcnt waits into the MFMA loop); kept for A/B
    constexpr int PPF = (NITEMS + WITEMS + NF - 1) / NF;  // staging pieces per (tap, k8) step
    auto h1 = [&](int flat, int cur, StageRegs& Rn, StageRegs& Rf) {
        const float* Ac = As + cur * ABUF;
        Cp = bufpos[cur];
        c_ch = bufch[cur];
        last = c_ch == a.n_chunks - 1;
        nbg0 = (Cp.ct * WN + wn) * NB;
        // (1) epilogue operands FIRST (older than the prefetch below, so the epilogue's counted vmcnt wait does not
        //     include the prefetch)
        if (last) {
            const float* tb = a.tbias ? a.tbias + (a.step_ptr ? (size_t)(*a.step_ptr) * a.tb_rowstride : 0) + (size_t)Cp.b * a.tbias_stride : nullptr;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int co = (nbg0 + nb) * 32 + j;
                const int coc = co < a.Cout ? co : a.Cout - 1;
                e_bias[nb] = a.bias ? a.bias[coc] : 0.f;
                e_tb[nb] = tb ? tb[coc] : 0.f;
                if (a.res) {
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int m = (wm * MB + mb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                            int oy = Cp.oy0 + m / TW, ox = Cp.ox0 + m % TW;
                            oy = oy < a.Hout ? oy : a.Hout - 1;
                            ox = ox < a.Wout ? ox : a.Wout - 1;
                            e_res[mb][nb][r] = a.res[(size_t)((Cp.b * a.Hout + oy) * a.Wout + ox) * a.Cout + coc];
                        }
                }
            }
        }
        // (2) prefetch stage flat+2
        if (flat + 2 < nflat) issue_loads(Rf);
        flush_stats();
        stamp();
        if (c_ch == 0) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
        }
        // (3) contraction over taps x chunk channels: A and B fragments both from LDS, no global load in here;
        //     stage flat+1 (register set Rn, loaded one step ago) is written to the OTHER LDS buffer piece by piece
        //     between the MFMA groups
        const bool stage_next = INTERLEAVE && (flat + 1 < nflat) && !(ABL & 32);
        if (stage_next) finish_prepare(Rn);
        const float* Wc = Ws + cur * WBUF + (wn * NB) * (NF * 256) + h * 128 + j * 4;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int tap = f / K8, k8 = f % K8;
            const int aoff = ((tap / KS) * IW + (tap % KS)) * LDA;
            float4 af[MB], bf[NB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) af[mb] = *reinterpret_cast<const float4*>(&Ac[abase[mb] + aoff + k8 * 8]);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bf[nb] = *reinterpret_cast<const float4*>(&Wc[nb * (NF * 256) + f * 256]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        if (ABL & 1) acc[mb][nb][i] += (&af[mb].x)[i] * (&bf[nb].x)[i];
                        else acc[mb][nb] = DDIF_MFMA_32x32x2((&af[mb].x)[i], (&bf[nb].x)[i], acc[mb][nb]);
            if (stage_next) {
#pragma unroll
                for (int pp = 0; pp < PPF; ++pp) finish_piece(Rn, cur ^ 1, f * PPF + pp);
            }
        }
        if (stage_next) finish_commit(Rn, cur ^ 1);
        stamp();
    };
    auto h2 = [&](int flat, int cur, StageRegs& Rn) {
        if (last && !(ABL & 64)) {
            // (4) epilogue of work item Cp.  Addresses = tile base + compile-time pixel offsets; bounds are only
            //     checked for tiles that stick out of the image.
            float s1 = 0.f, s2 = 0.f;
            const bool full = (Cp.oy0 + TH <= a.Hout) & (Cp.ox0 + TW <= a.Wout);
            const size_t tile_pix = (size_t)((Cp.b * a.Hout + Cp.oy0) * a.Wout + Cp.ox0);
            const int rowc = a.Wout * a.Cout;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int co = (nbg0 + nb) * 32 + j;
                const bool cok = co < a.Cout;
                float* obase = a.out + tile_pix * a.Cout + co;
                const float* fbase = a.film ? a.film + tile_pix * 2 * a.Cout + co : nullptr;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = (wm * MB + mb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const int my = m / TW, mx = m % TW;
                        if (cok && (full || (Cp.oy0 + my < a.Hout && Cp.ox0 + mx < a.Wout))) {
                            const int poff = my * rowc + mx * a.Cout;
                            float v = acc[mb][nb][r] + e_bias[nb];
                            v += e_tb[nb];
                            if (a.film) {
                                const float sc = fbase[2 * poff], sh = fbase[2 * poff + a.Cout];
                                v = v * (1.f + sc) + sh;
                            }
                            if (a.act_silu) v = dd_silu(v);
                            if (a.res) v += e_res[mb][nb][r];
                            if (!(ABL & 4) || v == 12345.678f) obase[poff] = v;
                            s1 += v;
                            s2 += v * v;
                        }
                    }
                }
            }
            if (a.st_out) {
                const double d1 = (double)wave_sum_fast(s1), d2 = (double)wave_sum_fast(s2);  // fp32 tree in-wave, fp64 beyond
                pend_par ^= 1;
                if (lane == 63) {
                    red[pend_par * 8 + wave * 2 + 0] = d1;
                    red[pend_par * 8 + wave * 2 + 1] = d2;
                }
                pend = true;
                pend_pos = Cp;
            }
        }
        stamp();
        // (5) stage flat+1 (loaded one step ago) -> the other LDS buffer
        if (!INTERLEAVE && flat + 1 < nflat && !(ABL & 32)) finish_stage(Rn, cur ^ 1);
        if ((ABL & 32) && flat + 1 < nflat) finish_commit(Rn, cur ^ 1);
        stamp();
    };

    if (DWM) {
        for (int i = tid; i < 9 * Ctot; i += 256) DWs[i] = a.dw_w[i];
        __syncthreads();
    }
    if (nflat > 0) {
        item_geometry();
        issue_loads(R0);
        if (nflat > 1) issue_loads(R1);
        finish_stage(R0, 0);
    }
    __syncthreads();
    stamp();
    if (GROUPS == 1) {
        for (int flat = 0; flat < nflat; flat += 2) {
            h1(flat, 0, R1, R0);
            h2(flat, 0, R1);
            __syncthreads();
            if (flat + 1 < nflat) {
                h1(flat + 1, 1, R0, R1);
                h2(flat + 1, 1, R0);
                __syncthreads();
            }
        }
    } else {
        // anti-phase schedule: group g runs H1(f) in phase 2f+g and H2(f) in phase 2f+g+1; one barrier per phase
        const int o0 = (int)((long long)(blockIdx.x * 2) * nwork / nvb), o1 = (int)((long long)(blockIdx.x * 2 + 1) * nwork / nvb),
                  o2 = (int)((long long)(blockIdx.x * 2 + 2) * nwork / nvb);
        const int nf0 = (o1 - o0) * a.n_chunks, nf1 = (o2 - o1) * a.n_chunks;
        const int nphase = 2 * nf0 > 2 * nf1 + 1 ? 2 * nf0 : 2 * nf1 + 1;
        int flat = 0;
        for (int ph = 0; ph < nphase; ++ph) {
            if (flat < nflat) {
                if (((ph + grp) & 1) == 0) {
                    if (flat & 1) h1(flat, 1, R0, R1);
                    else h1(flat, 0, R1, R0);
                } else if (ph >= grp + 1) {
                    if (flat & 1) h2(flat, 1, R0);
                    else h2(flat, 0, R1);
                    ++flat;
                }
            }
            __syncthreads();
        }
    }
    flush_stats();
}


template <int KS, int STRIDE, int UPS, int TH, int TW, int CK, int NBT, int GROUPS = 1, int PRO = 0>
constexpr size_t conv_smem_bytes() {  // NBT = n-blocks (of 32 couts) per 4-wave group = NB * WN
    constexpr int IH = (TH - 1) * STRIDE + KS, IW = (TW - 1) * STRIDE + KS;
    constexpr size_t dw = PRO == PRO_GN_DW ? (size_t)((TH + 2) * (TW + 2) * (CK + 4) + 9 * 256) : 0;
    return GROUPS * ((size_t)(2 * (IH * IW * (CK + 4) + NBT * KS * KS * (CK / 8) * 256) + dw) * sizeof(float) + 16 * sizeof(double));
}

}  // namespace ddif
