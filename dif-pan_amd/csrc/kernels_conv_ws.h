// Wave-specialised 3x3 convolution (NHWC fp32 in/out, bf16x3 products) for gfx950: the high-resolution 3x3 convs of the
// denoiser (models/sr3_dwt.py:288-327 Block / ResnetBlock, :521-533 FFN convs, :266-282 Upsample).
//
// Same arithmetic, data layout, prologues and epilogues as conv_mfma_kernel<..., MATH = 1> (kernels_conv.h), different
// execution structure.  With the products on v_mfma_f32_32x32x16_bf16 the matrix core and the vector ALU are separate
// pipes, so the staging work (global loads, GroupNorm + SiLU, three-way bf16 split, LDS writes) can run BESIDE the MFMAs
// -- but only from a different wavefront, and the waves of conv_mfma_kernel all stage, then all multiply.  Here one
// 512-thread workgroup = 4 CONSUMER waves (wave w: 64 pixels x 32 couts of the 16x16-pixel tile, 108 MFMAs per stage,
// epilogue) + 4 PRODUCER waves (stage s+1 of the pipeline: loads issued four stages ahead into three register sets,
// prologue, split, LDS writes), one s_barrier per stage, A and W tiles double-buffered in LDS (125 KiB).
// Each role has its own straight-line vmcnt stream: producers only wait for their oldest register set (counted,
// two younger sets stay in flight), consumers only for their epilogue operands.
#pragma once
#include "kernels_conv.h"

namespace ddif {

template <int UPS, int PRO, int EPI, int ABL = 0>
__global__ __launch_bounds__(512) void conv3_ws_kernel(ConvArgs a) {
    constexpr int TH = 16, TW = 16, CK = 16, KS = 3, TAPS = 9, IH = 18, IW = 18, LDA = 28, C4 = 4;
    constexpr int ABUF = IH * IW * LDA;        // floats per A buffer (3 bf16 planes x 32 B + 16 B pad per pixel)
    constexpr int WCHUNK = TAPS * 3 * 256;     // floats per weight chunk (32 couts x 16 cins x 9 taps x 3 planes of bf16)
    constexpr int NPT = 256;                   // producer threads
    constexpr int NITEMS = (IH * IW * C4 + NPT - 1) / NPT;   // 6
    constexpr int WITEMS = (WCHUNK / 4 + NPT - 1) / NPT;     // 7
    constexpr int DUMMY = 24;
    constexpr int MB = 2;
    constexpr bool GNP = (PRO == PRO_GN_SILU);
    constexpr bool RES = (EPI & EPI_RES) != 0, SILU = (EPI & EPI_SILU) != 0, TBS = (EPI & EPI_TBS) != 0;
    static_assert(PRO == PRO_NONE || PRO == PRO_GN_SILU, "prologues of the plain 3x3 convs");
    static_assert((EPI & (EPI_FILM | EPI_SOUT)) == 0, "FiLM / scalar-output epilogues stay on conv_mfma_kernel");

    DDIF_DYN_SMEM(smem);
    float* As = reinterpret_cast<float*>(smem);   // [2][ABUF]
    float* Ws = As + 2 * ABUF;                    // [2][WCHUNK]
    double* red = reinterpret_cast<double*>(smem + (size_t)2 * (ABUF + WCHUNK) * sizeof(float));  // [2][8]
    float* GBs = reinterpret_cast<float*>(smem + (size_t)2 * (ABUF + WCHUNK) * sizeof(float) + 16 * sizeof(double));  // gamma | beta
    const int GBN = a.n_chunks * CK;
    float* BTs = GBs + (GNP ? 2 * GBN : 0);       // bias (+ the step's time-bias row)

    const int tid = threadIdx.x, lane = tid & 63;
#ifdef DDIF_EMU
    const int wave = tid >> 6;
#else
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const bool consumer = wave < 4;
    const int h = lane >> 5, j = lane & 31;
    const int tiles = a.tiles_x * a.tiles_y;
    const int nwork = a.B * tiles * a.n_ct;
    const int w0 = (int)((long long)blockIdx.x * nwork / gridDim.x);
    const int w1 = (int)((long long)(blockIdx.x + 1) * nwork / gridDim.x);
    if (w0 >= w1) return;
    const int Hc = UPS ? a.Hin * 2 : a.Hin, Wc = UPS ? a.Win * 2 : a.Win;
    const int Ctot = a.c0 + a.c1;
    const int nflat = (w1 - w0) * a.n_chunks;
    const float* tbrow = a.tbias + (a.step_ptr ? (size_t)(*a.step_ptr) * a.tb_rowstride : 0);

    struct Pos { int work, ct, b, oy0, ox0; };
    auto locate = [&](int work) {
        Pos p;
        p.work = work;
        const int pt = work / a.n_ct;
        p.ct = work - pt * a.n_ct;
        p.b = pt / tiles;
        const int t = pt - p.b * tiles;
        const int ty = t / a.tiles_x;
        p.oy0 = ty * TH;
        p.ox0 = (t - ty * a.tiles_x) * TW;
        return p;
    };
    auto next_pos = [&](Pos p) {
        p.work += 1;
        if (++p.ct < a.n_ct) return p;
        p.ct = 0;
        p.ox0 += TW;
        if (p.ox0 >= a.tiles_x * TW) {
            p.ox0 = 0;
            p.oy0 += TH;
            if (p.oy0 >= a.tiles_y * TH) {
                p.oy0 = 0;
                ++p.b;
            }
        }
        return p;
    };

    // ---- tables shared by both roles
    if (GNP) {
        for (int i = tid; i < GBN; i += 512) {
            const int c = i < Ctot ? i : Ctot - 1;
            GBs[i] = a.gamma[c];
            GBs[GBN + i] = a.beta[c];
        }
    }
    for (int i = tid; i < a.n_ct * 32; i += 512) {
        const int c = i < a.Cout ? i : a.Cout - 1;
        BTs[i] = a.bias[c] + (TBS ? 0.f : tbrow[c]);
    }

#ifndef DDIF_EMU
    if (ABL & 32) {  // experiment: issue priority for one role
        if (!consumer) __builtin_amdgcn_s_setprio(3);
    }
    if (ABL & 64) {
        if (consumer) __builtin_amdgcn_s_setprio(3);
    }
#endif
    if (!consumer) {
        // =================================================================================== PRODUCER (waves 4..7)
        const int ptid = tid - 256;
        const int c4 = ptid % C4;
        // staging geometry is recomputed from ptid where it is needed instead of living in registers (three 52-VGPR
        // register sets leave little room): item `it` of this thread is pixel (ptid + it * 256) / 4 of the 18x18 halo tile,
        // weight item `it` is float4 number ptid + it * 256 of the 1728 in a chunk; the last item of each kind is partial
        const int p0 = ptid / C4;  // pixel of item 0; item `it` is pixel p0 + 64 * it
        const int a_lds0 = p0 * LDA + c4 * 2;
        const bool last_a_in = p0 + 64 * (NITEMS - 1) < IH * IW;         // threads 0..15 only
        const bool last_w_in = ptid + NPT * (WITEMS - 1) < WCHUNK / 4;  // threads 0..191 only
        const unsigned a_in = last_a_in ? (1u << NITEMS) - 1u : (1u << (NITEMS - 1)) - 1u;
        Pos L = locate(w0);
        int l_ch = 0;
        unsigned l_o0[NITEMS], l_o1[NITEMS];
        unsigned l_ok = 0;
        auto item_geometry = [&]() {
            const int iy0 = L.oy0 - 1, ix0 = L.ox0 - 1;
            l_ok = 0;
#pragma unroll
            for (int it = 0; it < NITEMS; ++it) {
                const int pixr = p0 + 64 * it;
                const int pix = pixr < IH * IW ? pixr : IH * IW - 1;
                const int iy = iy0 + pix / IW, ix = ix0 + pix % IW;
                const bool ok = (iy >= 0) & (iy < Hc) & (ix >= 0) & (ix < Wc);
                l_ok |= (ok ? 1u : 0u) << it;
                const int iyc = iy < 0 ? 0 : (iy >= Hc ? Hc - 1 : iy), ixc = ix < 0 ? 0 : (ix >= Wc ? Wc - 1 : ix);
                const int sp = (L.b * a.Hin + (UPS ? (iyc >> 1) : iyc)) * a.Win + (UPS ? (ixc >> 1) : ixc);
                l_o0[it] = (unsigned)sp * (unsigned)(a.c0 * 4);
                if (a.c1) l_o1[it] = (unsigned)sp * (unsigned)(a.c1 * 4);
            }
            l_ok &= a_in;
        };
        struct StageRegs {
            float4 sv[NITEMS], wv[WITEMS];
            unsigned ok;
            int cb, b;
        };
        auto issue_loads = [&](StageRegs& R) {  // unconditional and branch-free, like conv_mfma_kernel's
            const int cb = l_ch * CK;
            const bool s0 = cb < a.c0;
            const int nvalid = (s0 ? a.c0 : Ctot) - cb;
            const int c4c = c4 * 4 < nvalid ? c4 * 4 : nvalid - 4;
            const char* base = reinterpret_cast<const char*>(s0 ? a.in0 + cb : a.in1 + (cb - a.c0));
            const unsigned co = (unsigned)c4c * 4u;
#pragma unroll
            for (int it = 0; it < NITEMS; ++it) {
                if (ABL & 2) R.sv[it] = make_float4(0.5f, 0.25f, -0.5f, 0.125f);
                else R.sv[it] = *reinterpret_cast<const float4*>(base + (((s0 || !a.c1) ? l_o0[it] : l_o1[it]) + co));
            }
            const char* wbase = reinterpret_cast<const char*>(a.w + ((size_t)L.ct * a.n_chunks + l_ch) * WCHUNK);
#pragma unroll
            for (int it = 0; it < WITEMS; ++it) {
                if (ABL & 8) R.wv[it] = make_float4(0.01f, 0.02f, 0.03f, 0.04f);
                else R.wv[it] = *reinterpret_cast<const float4*>(wbase + (unsigned)(((it == WITEMS - 1 && !last_w_in) ? WCHUNK / 4 - 1 : ptid + it * NPT) * 16));
            }
            R.ok = l_ok;
            R.cb = cb;
            R.b = L.b;
            if (++l_ch == a.n_chunks) {
                l_ch = 0;
                if (L.work + 1 < w1) {
                    const bool same_tile = L.ct + 1 < a.n_ct;
                    L = next_pos(L);
                    if (!same_tile) item_geometry();
                }
            }
        };
        int gn_b = -1;
        float mean = 0.f, rstd = 1.f;
        auto finish_stage = [&](StageRegs& R, int buf) {
            float* dst = As + buf * ABUF;
            float* wdst = Ws + buf * WCHUNK;
            float ga[4] = {0.f, 0.f, 0.f, 0.f}, gb[4] = {0.f, 0.f, 0.f, 0.f};
            if (GNP) {
                if (R.b != gn_b) {
                    gn_finalize_wave(a.st0, a.np0, a.st1, a.np1, R.b, (double)Ctot * a.Hin * a.Win, &mean, &rstd);
                    gn_b = R.b;
                }
                const float4 gq = *reinterpret_cast<const float4*>(&GBs[R.cb + c4 * 4]);
                const float4 bq = *reinterpret_cast<const float4*>(&GBs[GBN + R.cb + c4 * 4]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ga[i] = (&gq.x)[i] * rstd;
                    gb[i] = (&bq.x)[i] - mean * ga[i];
                }
            }
#pragma unroll
            for (int it = 0; it < NITEMS; ++it) {
                const bool ok = (R.ok >> it) & 1u;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float x = (&R.sv[it].x)[i];
                    if (GNP) x = dd_silu(fmaf(x, ga[i], gb[i]));
                    v[i] = ok ? x : 0.f;  // zero padding comes AFTER the activation
                }
                unsigned h01, m01, l01, h23, m23, l23;
                dd_split3_pair(v[0], v[1], &h01, &m01, &l01);
                dd_split3_pair(v[2], v[3], &h23, &m23, &l23);
                const bool in = it < NITEMS - 1 || last_a_in;
                const int ps = in ? 8 : 0;
                const int al = in ? a_lds0 + it * 64 * LDA : DUMMY;
                *reinterpret_cast<uint2*>(&dst[al]) = make_uint2(h01, h23);
                *reinterpret_cast<uint2*>(&dst[al + ps]) = make_uint2(m01, m23);
                *reinterpret_cast<uint2*>(&dst[al + 2 * ps]) = make_uint2(l01, l23);
            }
#pragma unroll
            for (int it = 0; it < WITEMS; ++it) {
                float* wp = (it < WITEMS - 1 || last_w_in) ? wdst + (ptid + it * NPT) * 4 : dst + DUMMY;
                *reinterpret_cast<float4*>(wp) = R.wv[it];
            }
        };

        // THREE register sets (four spill: 4 x 52 VGPRs + the staging geometry exceed the 256 a wave may hold at two
        // waves per SIMD): the loads of stage s+4 are issued at stage s and consumed at stage s+3
        StageRegs R0, R1, R2;
        item_geometry();
        issue_loads(R0);  // stages 0..2; past the end of the range the loader re-reads the last item (never consumed)
        issue_loads(R1);
        issue_loads(R2);
        __syncthreads();  // tables
        finish_stage(R0, 0);
        issue_loads(R0);  // stage 3
        __syncthreads();  // stage 0 is in LDS
        // stage s: stage s+1 -> LDS buffer (s+1)&1 out of set (s+1)%3, then the loads of stage s+4 into the same set
        int s = 0;
        for (; s + 3 <= nflat; s += 3) {
            finish_stage(R1, (s + 1) & 1);
            issue_loads(R1);
            __syncthreads();
            finish_stage(R2, s & 1);
            issue_loads(R2);
            __syncthreads();
            finish_stage(R0, (s + 1) & 1);
            issue_loads(R0);
            __syncthreads();
        }
        if (s < nflat) {
            finish_stage(R1, (s + 1) & 1);
            __syncthreads();
            if (s + 1 < nflat) {
                finish_stage(R2, s & 1);
                __syncthreads();
            }
        }
        return;
    }

    // ======================================================================================= CONSUMER (waves 0..3)
    int abase[MB], e_my[MB], e_mx[MB];
    unsigned e_off[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int m = (wave * MB + mb) * 32 + j;
        abase[mb] = ((m / TW) * IW + (m % TW)) * LDA + 4 * h;
        e_my[mb] = m / TW;
        e_mx[mb] = m % TW;
        e_off[mb] = (unsigned)(((e_my[mb] * a.Wout + e_mx[mb]) * a.Cout + 4 * h) * 4);
    }
    Pos Cp = locate(w0);
    int c_ch = 0;
    bool pend = false;
    Pos pend_pos = Cp;
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
    f32x16 acc[MB];
    auto consume = [&](auto kind, const int cur) {
        constexpr bool LAST = decltype(kind)::LAST;
        const float* Ac = As + cur * ABUF;
        const float* Wc = Ws + cur * WCHUNK + h * 128 + j * 4;
        [[maybe_unused]] float4 e_t[4];
        [[maybe_unused]] float4 e_res[MB][4];
        bool full = true;
        unsigned e_po[MB];
        bool e_pok[MB];
        size_t tile_el = 0;
        if constexpr (LAST) {
            full = (Cp.oy0 + TH <= a.Hout) & (Cp.ox0 + TW <= a.Wout);
            const size_t tile_pix = (size_t)((Cp.b * a.Hout + Cp.oy0) * a.Wout + Cp.ox0);
            tile_el = tile_pix * a.Cout + Cp.ct * 32;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                e_pok[mb] = full || ((Cp.oy0 + e_my[mb] < a.Hout) & (Cp.ox0 + e_mx[mb] < a.Wout));
                e_po[mb] = e_pok[mb] ? e_off[mb] : (unsigned)(16 * h);
            }
            [[maybe_unused]] const char* tb = reinterpret_cast<const char*>(tbrow + (size_t)Cp.b * a.tbias_stride + Cp.ct * 32);
            [[maybe_unused]] const char* rbase = reinterpret_cast<const char*>(a.res + tile_el);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const unsigned cq = (Cp.ct * 32 + 8 * g + 4 * h < a.Cout) ? (unsigned)(8 * g * 4) : (unsigned)(-16 * h);
                if constexpr (TBS) e_t[g] = *reinterpret_cast<const float4*>(tb + (cq + 16u * h));
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
                    if constexpr (RES) e_res[mb][g] = *reinterpret_cast<const float4*>(rbase + (e_po[mb] + cq));
            }
        }
        flush_stats();
        if (c_ch == 0) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
        }
        float4 xa[2][MB][3], wb[2][3];
        auto load_frags3 = [&](int tap, int slot) {
            const int aoff = ((tap / KS) * IW + (tap % KS)) * LDA;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) xa[slot][mb][q] = *reinterpret_cast<const float4*>(&Ac[abase[mb] + aoff + q * 8]);
                wb[slot][q] = *reinterpret_cast<const float4*>(&Wc[(tap * 3 + q) * 256]);
            }
        };
        load_frags3(0, 0);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            if (tap + 1 < TAPS) load_frags3(tap + 1, (tap + 1) & 1);
            DDIF_SCHED_FENCE();
            const int sl = tap & 1;
            if (!(ABL & 1)) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) acc[mb] = DDIF_MFMA_32x32x16_BF16(wb[sl][2], xa[sl][mb][0], acc[mb]);  // lo * hi
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) acc[mb] = DDIF_MFMA_32x32x16_BF16(wb[sl][0], xa[sl][mb][2], acc[mb]);  // hi * lo
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) acc[mb] = DDIF_MFMA_32x32x16_BF16(wb[sl][1], xa[sl][mb][1], acc[mb]);  // mid * mid
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) acc[mb] = DDIF_MFMA_32x32x16_BF16(wb[sl][1], xa[sl][mb][0], acc[mb]);  // mid * hi
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) acc[mb] = DDIF_MFMA_32x32x16_BF16(wb[sl][0], xa[sl][mb][1], acc[mb]);  // hi * mid
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) acc[mb] = DDIF_MFMA_32x32x16_BF16(wb[sl][0], xa[sl][mb][0], acc[mb]);  // hi * hi
            }
            DDIF_SCHED_FENCE();
        }
        if constexpr (LAST) {
            float s1 = 0.f, s2 = 0.f;
            char* obase = reinterpret_cast<char*>(a.out + tile_el);
            auto epi = [&](auto guard) {
                constexpr bool GUARD = decltype(guard)::LAST;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int co = Cp.ct * 32 + 8 * g + 4 * h;
                        const float4 bt = *reinterpret_cast<const float4*>(&BTs[co]);
                        float v[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float x = acc[mb][4 * g + i] + (&bt.x)[i];
                            if constexpr (TBS) x += (&e_t[g].x)[i];
                            if constexpr (SILU) x = dd_silu(x);
                            if constexpr (RES) x += (&e_res[mb][g].x)[i];
                            v[i] = x;
                        }
                        if (!GUARD || (e_pok[mb] && co < a.Cout)) {
                            if (!(ABL & 4) || v[0] == 12345.678f)
                                *reinterpret_cast<float4*>(obase + (e_po[mb] + (unsigned)(8 * g * 4))) = make_float4(v[0], v[1], v[2], v[3]);
                            s1 += (v[0] + v[1]) + (v[2] + v[3]);
                            s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                        }
                    }
            };
            if (full && (Cp.ct + 1) * 32 <= a.Cout) epi(StageKind<0>{});
            else epi(StageKind<1>{});
            if (a.st_out) {
                const double d1 = (double)wave_sum_fast(s1), d2 = (double)wave_sum_fast(s2);
                pend_par ^= 1;
                if (lane == 63) {
                    red[pend_par * 8 + wave * 2 + 0] = d1;
                    red[pend_par * 8 + wave * 2 + 1] = d2;
                }
                pend = true;
                pend_pos = Cp;
            }
        }
    };

    __syncthreads();  // tables
    __syncthreads();  // stage 0 is in LDS
    for (int s = 0; s < nflat; ++s) {
        if (c_ch == a.n_chunks - 1) consume(StageKind<1>{}, s & 1);
        else consume(StageKind<0>{}, s & 1);
        if (++c_ch == a.n_chunks) {
            c_ch = 0;
            Cp = next_pos(Cp);
        }
        __syncthreads();
    }
    flush_stats();  // the partial of the last item: red[] was written by this role's waves before the final barrier
}

constexpr size_t conv3_ws_smem_bytes() { return (size_t)2 * (18 * 18 * 28 + 9 * 3 * 256) * sizeof(float) + 16 * sizeof(double); }

}  // namespace ddif
