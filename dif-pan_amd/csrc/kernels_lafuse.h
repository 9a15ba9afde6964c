// The attention half of the decoder's FastAttnCondInjection at the high-resolution levels as ONE kernel (inference plans).
//
// Reference (models/sr3_dwt.py:536-566; the cond-only k / v / context side is hoisted into set_cond and folded into per-sample
// weights M_b = scale * W_out * blockdiag(ctx_b^T), kernels_misc.h pack_mix_weights_x3_kernel):
//     xn = prenorm_x(cat[h, skip])                 GroupNorm(1 group)
//     q  = q.1(q.0(xn))                            depthwise 3x3, then 1x1 (+ bias)
//     p  = softmax over image ROWS of q            (dim = -2: one softmax per (column, channel))
//     a  = attn_out(ctx . p) + attn_res(xn)        = M_b p + W_res xn + (b_out + b_res)
// Until round 4 this was three launches per block -- q = 1x1(dw3x3(GN(cat))) writing q AND xn, a column-statistics pass re-reading q,
// and a 1x1 conv over cat[softmax(q), xn] -- 436 MB of traffic per block at 64 x 64 x 64 channels (B = 64) for 100 MB of real input /
// output, 1.37 ms of a 4.7 ms denoising step.  Here a workgroup owns WHOLE COLUMNS of one sample (TH = H rows x TW columns = 256
// pixels, a one-pixel halo for the depthwise conv), so the column softmax is local and neither q nor xn ever leaves the CU:
//
//   per 16-channel chunk:  raw halo tile (prefetched one chunk ahead, registers) -> GroupNorm -> Hs (fp32, LDS) | barrier |
//                          depthwise 3x3 -> dwq as two half planes (f16x2, ddif_dev.h), centre of Hs -> xn as three bf16 planes | barrier |
//                          acc_q += W_q[chunk] dwq  (f16x2, 3 products)      acc_a += W_res[chunk] xn  (bf16x3, 6 products)
//   then:                  q = acc_q + b_q; column max / sum(exp) over the 32 pixel lanes of a wave (pixels are numbered column-major:
//                          a wave holds (half) a column; DPP + one xor-16 shuffle), the two waves of a 64-row column meet in LDS;
//                          p -> bf16 planes in the wave's own LDS region -> acc_a += M_b[32-channel block] p  (bf16x3) ; + bias ; store.
//   Eight wavefronts: wave w owns pixels 32w .. 32w+31 and ALL output channels (acc_q: qd / 32 blocks, acc_a: dout / 32 blocks), so
//   weights are the only shared operand: every wave streams the same pre-packed B fragments from L1 / L2 (16 KB + 24 KB per item).
//   The weights of a chunk are requested before its GroupNorm / depthwise stages and consumed after them.
#pragma once
#include "ddif_dev.h"

namespace ddif {

struct LaFuseArgs {
    const float* in0;        // h     [B, H, W, c0]
    const float* in1;        // skip  [B, H, W, c1]
    int c0, c1;              // multiples of 16; fea = c0 + c1 = 32 * NBQ
    int B, H, W, b0;         // the launch covers samples [b0, b0 + B); H == TH
    const double* st0;       // GroupNorm partials of h / skip
    int np0;
    const double* st1;
    int np1;
    const float* gamma;      // prenorm_x [fea]
    const float* beta;
    const float* dw_w;       // q.0 depthwise weights [9][fea]
    const float* wq;         // q.1 as f16x2 planes (ddif_net.cpp pack_conv_f16, ck = 32): [n-block][chunk][k16][hi | lo][256 floats]
    int nchq;                // 32-channel chunks of q.1 (= NBQ)
    const float* bq;         // q.1 bias [fea]
    const float* wmix;       // per-sample bf16x3 planes of [M_b | W_res] (pack_mix_weights_x3_kernel, ck = 32): [n-block][chunk][k16][3][256]
    long long wmix_bstride;  // floats per sample
    int nch_mix;             // 32-channel chunks of that pack (= 2 * NBQ)
    const float* bias;       // attn_out.bias + attn_res.bias [dout]
    float* out;              // [B, H, W, dout]
    int dout;
    long long* dbg;          // microbenchmark instrumentation (ABL & 64) only
    int xcd;                 // wg_work_range: XCD-contiguous work partition (ddif_dev.h)
};

template <int TH, int TW, int NBQ, int NBA>
struct LaFuseGeom {
    static constexpr int NPX = TH * TW, NWV = NPX / 32;  // pixels / wavefronts of a workgroup (256 / 8, or 128 / 4: round 6)
    static constexpr int HH = TH + 2, HW = TW + 2;      // halo tile
    static constexpr int LDH = 24;                      // floats per halo pixel (16 channels + pad: conflict-free depthwise reads)
    static constexpr int LDQ = 20;                      // dwq: 2 half planes x 32 B + 16 B pad
    static constexpr int LDX = 28;                      // xn: 3 bf16 planes x 32 B + 16 B pad
    static constexpr int APS = 60;                      // p: 2 slabs x (3 planes x 32 B) + pad, floats per pixel (conflict-free fragment reads)
    static constexpr int HS = HH * HW * LDH, AQ = NPX * LDQ, AX = NPX * LDX;
    static constexpr int NPC = 2 * NBQ + 3 * NBA;       // 1 KiB weight pieces of one chunk: q.1 (hi | lo per block), attn_res (3 planes per block)
    static constexpr int WC = NPC * 256;                // floats
    static constexpr int NPF = 6 * NBA;                 // pieces of one 32-channel block of M_b: [block][k16][plane]
    static constexpr int FM = 2 * NPF * 256;            // double-buffered by block parity
    static constexpr int AP = NPX * APS;                // aliases Hs | Aq | Ax once the chunk loop is over (written behind the first block's barrier)
    static constexpr int FEA = 32 * NBQ;
    static constexpr int TAB = 2 * FEA + 9 * FEA + FEA + 32 * NBA;  // gamma | beta | depthwise | q bias | output bias
    static constexpr int CST = 2 * NWV * 32 * 2;        // (max, sum) exchange of the wave pairs of 64-row columns, double-buffered by block parity: [parity][wave][j][max | sum]
    // Fm is NOT aliased: block 0's fragments are written while other waves may still read the last chunk's Ax / Wc
    static constexpr int MAIN = ((HS + AQ + AX + WC) > AP ? (HS + AQ + AX + WC) : AP) + FM;
    static constexpr size_t smem = (size_t)(MAIN + TAB + CST) * sizeof(float);
};

// max / sum over the 32 lanes of each wavefront half, result in every lane of the half: four DPP steps inside the 16-lane rows, then
// v_permlane16_swap_b32 (gfx950) exchanges the odd rows of one copy with the even rows of the other -- pure VALU, no LDS round trip.
// v_permlane16_swap_b32 inside s_nop guards: hipcc (ROCm 7.2) schedules the builtin without the wait states the instruction needs
// next to the DPP ops that feed / follow it -- the unguarded form returned wrong maxima on MI355X (2e-2 errors in every forward
// parity test; tools/probes/permlane_swap.cpp shows the instruction itself does what the ISA says), the guarded form is exact
__device__ __forceinline__ void row_pair_swap(float* x, float* y) {
#ifndef DDIF_EMU
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(*x), "+v"(*y));
#endif
}
template <bool PAIR = true>  // PAIR: over the 32 lanes of a wavefront half; else over the 16-lane rows (a 16-row image column per row)
__device__ __forceinline__ float half_allmax(float v) {
#ifdef DDIF_EMU
#pragma unroll
    for (int m = 1; m <= (PAIR ? 16 : 8); m <<= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
#else
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false)));   // quad_perm [1,0,3,2]
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false)));   // quad_perm [2,3,0,1]
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false)));  // row_ror:4
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false)));  // row_ror:8
    if constexpr (!PAIR) return v;
    float x = v, y = v;  // every lane holds its 16-lane row's maximum; swap: x = rows [0, 0, 2, 2], y = rows [1, 1, 3, 3]
    row_pair_swap(&x, &y);
    return fmaxf(x, y);
#endif
}
template <bool PAIR = true>
__device__ __forceinline__ float half_allsum(float v) {
#ifdef DDIF_EMU
#pragma unroll
    for (int m = 1; m <= (PAIR ? 16 : 8); m <<= 1) v += __shfl_xor(v, m);
    return v;
#else
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    if constexpr (!PAIR) return v;
    float x = v, y = v;
    row_pair_swap(&x, &y);
    return x + y;  // (lower row + upper row in every lane of the half)
#endif
}
// a wavefront's LDS writes become visible to its own later reads (other lanes): the LDS executes a wave's instructions in order, the
// compiler must only be kept from moving the read above the write; the host emulator runs lanes as fibers and needs a real rendezvous
#ifdef DDIF_EMU
#define DDIF_WAVE_LDS_SYNC() __syncthreads()
#else
#define DDIF_WAVE_LDS_SYNC() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")
#endif

// ABL (tools/mbench_la.cpp only): 1 = no input loads, 2 = no weight loads, 4 = no MFMAs, 8 = no depthwise / split stage, 16 = no softmax + attn_out part,
// 32 = no output stores, 64 = s_memtime stamps of thread 0 into a.dbg
// Round 6: TH * TW = 256 pixels on eight wavefronts, or 128 pixels on FOUR.  A wave's work does not depend on the workgroup's size (32 pixels x all output channels,
// weights shared through LDS), and the kernel is bound by vector issue with two waves per SIMD: when the eight-wave grid would leave CUs idle (16 x 16 level at
// B = 64: 64 workgroups; every level at 8-16 tiles per GPU) the host launches twice as many four-wave workgroups and every wave has a SIMD to itself.  Same
// per-pixel arithmetic in the same order: results do not depend on the choice (tests/test_env_switches.py DDIF_LA_NW).
template <int TH, int TW, int NBQ, int NBA, int ABL = 0>
__global__ __launch_bounds__(TH * TW * 2) void linattn_fused_kernel(LaFuseArgs a) {
    using G = LaFuseGeom<TH, TW, NBQ, NBA>;
    constexpr int NW = G::NWV, NTHR = 64 * NW;
    static_assert((TH * TW == 256 || TH * TW == 128) && (TH == 16 || TH == 32 || TH == 64), "a workgroup owns 256 or 128 pixels = whole columns of the sample");
    static_assert(TH != 64 || NW % 2 == 0, "64-row columns span a wave pair");
    constexpr int HH = G::HH, HW = G::HW, LDH = G::LDH, LDQ = G::LDQ, LDX = G::LDX, APS = G::APS, FEA = G::FEA;
    constexpr int NCH = 2 * NBQ;                              // 16-channel chunks
    constexpr int NIT = (HH * HW * 4 + NTHR - 1) / NTHR;      // raw float4 items per thread and chunk
    constexpr int NPC = G::NPC, NPF = G::NPF;
    constexpr int WIT = (NPC + NW - 1) / NW, FIT = (NPF + NW - 1) / NW;   // weight pieces per wave (a piece = 64 lanes x 16 B)
    constexpr float L2E = 1.4426950408889634f;
    constexpr bool PAIRDW = (NBQ == 2 && NBA == 1);  // depthwise stage on vertically adjacent pixel pairs (round 6): the 32 + 32 -> 32 blocks of the 64 x 64 level

    dd_touch_kernargs<sizeof(LaFuseArgs)>();  // every line of the argument block in ONE round trip (ddif_dev.h)
    DDIF_DYN_SMEM(smem);
    float* Hs = reinterpret_cast<float*>(smem);
    float* Aq = Hs + G::HS;
    float* Ax = Aq + G::AQ;
    float* Wc = Ax + G::AX;            // this chunk's weight pieces, B-fragment order
    float* Ap = Hs;                    // after the chunk loop
    float* Fm = Hs + G::MAIN - G::FM;  // [2][NPF][256]
    float* GB = Hs + G::MAIN;          // gamma [FEA] | beta [FEA]
    float* DW = GB + 2 * FEA;          // [9][FEA]
    float* BQ = DW + 9 * FEA;          // [FEA]
    float* BO = BQ + FEA;              // [32 * NBA]
    float* Cst = BO + 32 * NBA;

    const int tid = threadIdx.x, lane = tid & 63;
#ifdef DDIF_EMU
    const int wave = tid >> 6;
#else
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const int h = lane >> 5, j = lane & 31;
    [[maybe_unused]] int dbg_n = 0;
    auto stamp = [&]() {
#ifndef DDIF_EMU
        if ((ABL & 64) && a.dbg && tid == 0 && dbg_n < 127) a.dbg[blockIdx.x * 128 + dbg_n++] = (long long)__builtin_amdgcn_s_memtime();
#endif
    };
    const int nstrips = (a.W + TW - 1) / TW;
    const int nwork = a.B * nstrips;
    int w0, w1;
    wg_work_range(nwork, &w0, &w1, a.xcd);
    if (w0 >= w1) return;

    // raw staging: item = halo pixel * 4 + channel quad (geometry recomputed per use: divisions by constants, no registers held)
    const int c4 = tid & 3;
    float4 raw[NIT];
    unsigned rok = 0;
    auto load_raw = [&](int b, int x0, int k) {  // chunk k of strip (b, x0): clamped addresses, validity as a mask
        const int cb = 16 * k;
        const bool s0 = cb < a.c0;
        const float* src = s0 ? a.in0 + cb + 4 * c4 : a.in1 + (cb - a.c0) + 4 * c4;
        const int cs = s0 ? a.c0 : a.c1;
        rok = 0;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int pix = (tid + it * NTHR) >> 2;
            const bool in = pix < HH * HW;
            const int pc = in ? pix : HH * HW - 1;
            const int y = pc / HW - 1, x = x0 + pc % HW - 1;
            const bool ok = in & (y >= 0) & (y < a.H) & (x >= 0) & (x < a.W);
            rok |= (ok ? 1u : 0u) << it;
            const int yc = y < 0 ? 0 : (y >= a.H ? a.H - 1 : y), xc = x < 0 ? 0 : (x >= a.W ? a.W - 1 : x);
            if (ABL & 1) raw[it] = make_float4(0.5f, -0.25f, 0.125f, 1.f);
            else raw[it] = *reinterpret_cast<const float4*>(src + ((size_t)(b * a.H + yc) * a.W + xc) * cs);
        }
    };
    // weight pieces of chunk k: wave w fetches pieces w, w + 8 (1 KiB each, lane-linear) -- q.1 block nb plane pl: piece 2 nb + pl; attn_res block nb plane pl: 2 NBQ + 3 nb + pl
    float4 wst[WIT];
    auto load_weights = [&](const float* wmix_b, int k) {
#pragma unroll
        for (int i = 0; i < WIT; ++i) {
            const int pc = wave + NW * i;
            const int p = pc < NPC ? pc : NPC - 1;  // (waves past the end re-read the last piece; not written)
            const float* src;
            if (p < 2 * NBQ) src = a.wq + ((size_t)(((p >> 1) * a.nchq + (k >> 1)) * 2 + (k & 1)) * 2 + (p & 1)) * 256;
            else src = wmix_b + ((size_t)((((p - 2 * NBQ) / 3) * a.nch_mix + ((NCH + k) >> 1)) * 2 + (k & 1)) * 3 + (p - 2 * NBQ) % 3) * 256;
            wst[i] = (ABL & 2) ? make_float4(1e-3f, 2e-3f, 3e-3f, (float)k) : *reinterpret_cast<const float4*>(src + lane * 4);
        }
    };
    auto store_weights = [&]() {
#pragma unroll
        for (int i = 0; i < WIT; ++i)
            if (wave + NW * i < NPC) *reinterpret_cast<float4*>(&Wc[(wave + NW * i) * 256 + lane * 4]) = wst[i];
    };
    // the 6 NBA pieces of block nb of M_b: piece (na * 2 + k16) * 3 + pl
    float4 fst[FIT];
    auto load_mix = [&](const float* wmix_b, int nb) {
#pragma unroll
        for (int i = 0; i < FIT; ++i) {
            const int pc = wave + NW * i;
            const int p = pc < NPF ? pc : NPF - 1;
            const float* src = wmix_b + ((size_t)(((p / 6) * a.nch_mix + nb) * 2 + (p % 6) / 3) * 3 + p % 3) * 256;
            fst[i] = (ABL & 2) ? make_float4(1e-3f, 2e-3f, 3e-3f, (float)nb) : *reinterpret_cast<const float4*>(src + lane * 4);
        }
    };
    auto store_mix = [&](int par) {
#pragma unroll
        for (int i = 0; i < FIT; ++i)
            if (wave + NW * i < NPF) *reinterpret_cast<float4*>(&Fm[(par * NPF + wave + NW * i) * 256 + lane * 4]) = fst[i];
    };

    int gn_b = -1;
    float mean = 0.f, rstd = 1.f;
    GnPartials gp0;  // the first sample's GroupNorm partials: requested with the first strip (one round trip on the cold caches of a launch), reduced behind the table fills
    {
        const int b = a.b0 + w0 / nstrips;
        gn_load_partials(a.st0, a.np0, a.st1, a.np1, b, &gp0);
        load_raw(b, (w0 % nstrips) * TW, 0);
    }
    // tables (once per launch) -- AFTER the first strip's loads have been issued: the table fills wait for their own loads, and on the cold caches every launch starts with that is a
    // round trip the first tile should share, not follow
    // (round 6: every table load is issued before any is stored -- the load -> store loops of the first form were one dependent round trip per iteration)
    {
        constexpr int NDW = (9 * FEA + NTHR - 1) / NTHR, NG = (FEA + NTHR - 1) / NTHR;
        static_assert(32 * NBA <= NTHR, "output bias table: one pass");
        float tg[NG], tb[NG], tq[NG], tdw[NDW];
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const int i = tid + k * NTHR, ic = i < FEA ? i : FEA - 1;
            tg[k] = a.gamma[ic];
            tb[k] = a.beta[ic];
            tq[k] = a.bq[ic];
        }
        const float to = a.bias[tid < a.dout ? tid : a.dout - 1];
#pragma unroll
        for (int k = 0; k < NDW; ++k) {
            const int i = tid + k * NTHR;
            tdw[k] = a.dw_w[i < 9 * FEA ? i : 9 * FEA - 1];
        }
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const int i = tid + k * NTHR;
            if (i < FEA) {
                GB[i] = tg[k];
                GB[FEA + i] = tb[k];
                BQ[i] = tq[k];
            }
        }
        if (tid < 32 * NBA) BO[tid] = tid < a.dout ? to : 0.f;
#pragma unroll
        for (int k = 0; k < NDW; ++k) {
            const int i = tid + k * NTHR;
            if (i < 9 * FEA) DW[i] = tdw[k];
        }
    }
    {
        const int b = a.b0 + w0 / nstrips;
        gn_reduce_partials(gp0, a.st0, a.np0, a.st1, a.np1, b, (double)FEA * a.H * a.W, &mean, &rstd);
        gn_b = b;
    }
    __syncthreads();  // tables

    for (int work = w0; work < w1; ++work) {
        const int bl = work / nstrips, b = a.b0 + bl, x0 = (work - bl * nstrips) * TW;
        if (b != gn_b) {  // workgroup-uniform; every wavefront reduces the producers' partials itself
            gn_finalize_wave(a.st0, a.np0, a.st1, a.np1, b, (double)FEA * a.H * a.W, &mean, &rstd);
            gn_b = b;
        }
        f32x16 accq[NBQ], acca[NBA];
#pragma unroll
        for (int nb = 0; nb < NBQ; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) accq[nb][r] = 0.f;
#pragma unroll
        for (int nb = 0; nb < NBA; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acca[nb][r] = 0.f;
        const float* wmix_b = a.wmix + (size_t)b * a.wmix_bstride;
        if constexpr (!(ABL & 16)) load_mix(wmix_b, 0);  // block 0 of M_b: requested before the chunk loop, written to LDS behind it

#pragma unroll 1
        for (int k = 0; k < NCH; ++k) {
            stamp();  // 0: chunk start
            // (a) this chunk's weights: requested now, written to LDS in stage (c), read back as fragments in stage (d)
            load_weights(wmix_b, k);
            // (b) GroupNorm of the raw halo tile -> Hs (zero padding comes after the normalisation: the depthwise conv pads xn)
            {
                const float4 gq = *reinterpret_cast<const float4*>(&GB[16 * k + 4 * c4]);
                const float4 bq4 = *reinterpret_cast<const float4*>(&GB[FEA + 16 * k + 4 * c4]);
                float ga[4], gb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ga[i] = (&gq.x)[i] * rstd;
                    gb[i] = (&bq4.x)[i] - mean * ga[i];
                }
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const bool ok = (rok >> it) & 1u;
                    const int pix = (tid + it * NTHR) >> 2;
                    float4 v;
                    v.x = ok ? fmaf(raw[it].x, ga[0], gb[0]) : 0.f;
                    v.y = ok ? fmaf(raw[it].y, ga[1], gb[1]) : 0.f;
                    v.z = ok ? fmaf(raw[it].z, ga[2], gb[2]) : 0.f;
                    v.w = ok ? fmaf(raw[it].w, ga[3], gb[3]) : 0.f;
                    if (pix < HH * HW) *reinterpret_cast<float4*>(&Hs[pix * LDH + 4 * c4]) = v;
                }
            }
            stamp();  // 1: GroupNorm stage done (includes the wait for the raw tile)
            // the next chunk's (or the next strip's first) raw tile flies during the stages below
            if (k + 1 < NCH) {
                load_raw(b, x0, k + 1);
            } else {
                const int wn = work + 1 < w1 ? work + 1 : work;
                const int bn = wn / nstrips;
                load_raw(a.b0 + bn, (wn - bn * nstrips) * TW, 0);
            }
            __syncthreads();
            stamp();  // 2: barrier
            // (c) depthwise 3x3 of the chunk -> dwq (two half planes, pre-scaled), and the centre pixels = xn (three bf16 planes);
            //     pixels are numbered column-major: p = x * TH + y.  Round 6: a thread owns TWO VERTICALLY ADJACENT pixels (p0 = 2 (tid / 4), p0 + 1: same column,
            //     TH is even) and walks the four halo rows they share once -- 12 tile reads + 9 weight reads per thread and chunk instead of 18 + 18: the stage
            //     issues 36 ds_read_b128 per thread otherwise (295 KB per chunk at 128 B / clk ~ the 2.0-2.4 k ticks the stamps show; measured 73.8 -> 69.6 us
            //     per launch at 64 x 64, the stage 1.8-2.4 k -> 1.5-2.0 k ticks: the rest is its FMAs and splits).  Only where it does not spill (PAIRDW).  One halo row
            //     and two weight rows live at a time (all reads hoisted, as hipcc prefers, push the accumulators into scratch); same products, same order per pixel.
            if constexpr (PAIRDW) {
              if (!(ABL & 8)) {
                const int p0 = (tid >> 2) * 2, x = p0 / TH, y = p0 % TH;
                float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
                float4 wprev[3], wcur[3];
#pragma unroll
                for (int r = 0; r < 4; ++r) {  // halo row y + r: tap row r of pixel a (r < 3), tap row r - 1 of pixel b (r > 0)
                    float4 hv[3];
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx) {
                        hv[tx] = *reinterpret_cast<const float4*>(&Hs[((y + r) * HW + x + tx) * LDH + 4 * c4]);
                        if (r > 0) wprev[tx] = wcur[tx];
                        if (r < 3) wcur[tx] = *reinterpret_cast<const float4*>(&DW[(3 * r + tx) * FEA + 16 * k + 4 * c4]);
                    }
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx) {
                        if (r < 3) {
                            sa[0] = fmaf(hv[tx].x, wcur[tx].x, sa[0]);
                            sa[1] = fmaf(hv[tx].y, wcur[tx].y, sa[1]);
                            sa[2] = fmaf(hv[tx].z, wcur[tx].z, sa[2]);
                            sa[3] = fmaf(hv[tx].w, wcur[tx].w, sa[3]);
                        }
                        if (r > 0) {
                            sb[0] = fmaf(hv[tx].x, wprev[tx].x, sb[0]);
                            sb[1] = fmaf(hv[tx].y, wprev[tx].y, sb[1]);
                            sb[2] = fmaf(hv[tx].z, wprev[tx].z, sb[2]);
                            sb[3] = fmaf(hv[tx].w, wprev[tx].w, sb[3]);
                        }
                    }
                    if (r == 1 || r == 2) {  // the centre tap of pixel a / b IS xn = GroupNorm(cat[h, skip]): three bf16 planes, written at once (nothing kept)
                        const int p = p0 + (r - 1);
                        unsigned h01, m01, l01, h23, m23, l23;
                        dd_split3_pair(hv[1].x, hv[1].y, &h01, &m01, &l01);
                        dd_split3_pair(hv[1].z, hv[1].w, &h23, &m23, &l23);
                        *reinterpret_cast<uint2*>(&Ax[p * LDX + 2 * c4]) = make_uint2(h01, h23);
                        *reinterpret_cast<uint2*>(&Ax[p * LDX + 8 + 2 * c4]) = make_uint2(m01, m23);
                        *reinterpret_cast<uint2*>(&Ax[p * LDX + 16 + 2 * c4]) = make_uint2(l01, l23);
                    }
                    DDIF_SCHED_FENCE();
                }
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int p = p0 + it;
                    const float* sv = it ? sb : sa;
                    unsigned h01, l01, h23, l23;
                    dd_split2_pair(sv[0] * DDIF_F16_ASCALE, sv[1] * DDIF_F16_ASCALE, &h01, &l01);
                    dd_split2_pair(sv[2] * DDIF_F16_ASCALE, sv[3] * DDIF_F16_ASCALE, &h23, &l23);
                    *reinterpret_cast<uint2*>(&Aq[p * LDQ + 2 * c4]) = make_uint2(h01, h23);
                    *reinterpret_cast<uint2*>(&Aq[p * LDQ + 8 + 2 * c4]) = make_uint2(l01, l23);
                }
            }
            } else {
            // (the form of rounds 4-5: two items per thread, 18 + 18 reads each -- kept where the pair form spills: measured 45.5 vs 48.6 us at 32 x 32, NBQ = 4)
            //     pixels are numbered column-major: p = x * TH + y.  One tap ROW at a time (3 + 3 LDS reads in flight): all 18 reads of an
            //     item hoisted, as hipcc prefers, cost 72 registers and pushed the accumulators into scratch.
#pragma unroll
            for (int it = 0; it < ((ABL & 8) ? 0 : 2); ++it) {
                const int item = tid + it * NTHR, p = item >> 2;
                const int x = p / TH, y = p % TH;
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
                float4 cen = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int ty = 0; ty < 3; ++ty) {
                    float4 hv[3], wk[3];
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx) {
                        hv[tx] = *reinterpret_cast<const float4*>(&Hs[((y + ty) * HW + x + tx) * LDH + 4 * c4]);
                        wk[tx] = *reinterpret_cast<const float4*>(&DW[(3 * ty + tx) * FEA + 16 * k + 4 * c4]);
                    }
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx) {
                        s0 = fmaf(hv[tx].x, wk[tx].x, s0);
                        s1 = fmaf(hv[tx].y, wk[tx].y, s1);
                        s2 = fmaf(hv[tx].z, wk[tx].z, s2);
                        s3 = fmaf(hv[tx].w, wk[tx].w, s3);
                    }
                    if (ty == 1) cen = hv[1];
                    DDIF_SCHED_FENCE();
                }
                unsigned h01, l01, h23, l23;
                dd_split2_pair(s0 * DDIF_F16_ASCALE, s1 * DDIF_F16_ASCALE, &h01, &l01);
                dd_split2_pair(s2 * DDIF_F16_ASCALE, s3 * DDIF_F16_ASCALE, &h23, &l23);
                *reinterpret_cast<uint2*>(&Aq[p * LDQ + 2 * c4]) = make_uint2(h01, h23);
                *reinterpret_cast<uint2*>(&Aq[p * LDQ + 8 + 2 * c4]) = make_uint2(l01, l23);
                unsigned m01, m23;
                dd_split3_pair(cen.x, cen.y, &h01, &m01, &l01);
                dd_split3_pair(cen.z, cen.w, &h23, &m23, &l23);
                *reinterpret_cast<uint2*>(&Ax[p * LDX + 2 * c4]) = make_uint2(h01, h23);
                *reinterpret_cast<uint2*>(&Ax[p * LDX + 8 + 2 * c4]) = make_uint2(m01, m23);
                *reinterpret_cast<uint2*>(&Ax[p * LDX + 16 + 2 * c4]) = make_uint2(l01, l23);
            }
            }
            store_weights();
            stamp();  // 3: depthwise + split stage done
            __syncthreads();
            stamp();  // 4: barrier
            // (d) contraction of the chunk: this wave's 32 pixels x all output channels; weight fragments from LDS (1 KiB conflict-free reads)
            {
                const int p = 32 * wave + j;
                float4 xq[2], xr[3];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) xq[pl] = *reinterpret_cast<const float4*>(&Aq[p * LDQ + pl * 8 + 4 * h]);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) xr[pl] = *reinterpret_cast<const float4*>(&Ax[p * LDX + pl * 8 + 4 * h]);
#pragma unroll
                for (int nb = 0; nb < NBQ; ++nb) {
                    const float4 f0 = *reinterpret_cast<const float4*>(&Wc[(2 * nb + 0) * 256 + lane * 4]);
                    const float4 f1 = *reinterpret_cast<const float4*>(&Wc[(2 * nb + 1) * 256 + lane * 4]);
                    f32x16 c = accq[nb];
                    if (ABL & 4) {
                        c[0] += f1.x * xq[0].x + f0.y * xq[1].y;
                        accq[nb] = c;
                        continue;
                    }
                    // D = X W: rows = the wave's pixels, columns = the block's q channels (see the softmax section) -- the products and their order are those of W X
                    c = DDIF_MFMA_32x32x16_F16(xq[0], f1, c);  // hi * lo
                    c = DDIF_MFMA_32x32x16_F16(xq[1], f0, c);  // lo * hi
                    c = DDIF_MFMA_32x32x16_F16(xq[0], f0, c);  // hi * hi
                    accq[nb] = c;
                }
#pragma unroll
                for (int nb = 0; nb < NBA; ++nb) {
                    float4 f[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) f[pl] = *reinterpret_cast<const float4*>(&Wc[(2 * NBQ + 3 * nb + pl) * 256 + lane * 4]);
                    f32x16 c = acca[nb];
                    if (ABL & 4) {
                        c[0] += f[2].x * xr[0].x + f[1].y * xr[1].y + f[0].z * xr[2].z;
                        acca[nb] = c;
                        continue;
                    }
                    c = DDIF_MFMA_32x32x16_BF16(f[2], xr[0], c);
                    c = DDIF_MFMA_32x32x16_BF16(f[0], xr[2], c);
                    c = DDIF_MFMA_32x32x16_BF16(f[1], xr[1], c);
                    c = DDIF_MFMA_32x32x16_BF16(f[1], xr[0], c);
                    c = DDIF_MFMA_32x32x16_BF16(f[0], xr[1], c);
                    c = DDIF_MFMA_32x32x16_BF16(f[0], xr[0], c);
                    acca[nb] = c;
                }
            }
            stamp();  // 5: contraction issued
            // (no barrier here: the next chunk's GroupNorm stage writes Hs, which nobody reads any more; Aq / Ax / Wc are rewritten behind its barrier)
        }

        // ---- q complete: column softmax over H, one 32-channel block of q at a time.  Round 6: the q contraction is issued as D = X W (rows = pixels, columns = q
        //      channels -- the operands of stage (d) swapped), so lane (j, h) of wave w holds, per block nb, channel 32 nb + j of the SIXTEEN pixels
        //      32 w + 8 g + 4 h + i: rows of one image column (TH >= 32) or of two (TH = 16: g < 2 / g >= 2).  Max and sum over a column are then in-register
        //      trees + ONE cross-half exchange (lanes l, l ^ 32) instead of sixteen 5-step cross-lane all-reduces each (the section was 43 % of a work item, all
        //      VALU / DPP issue: tools/mbench_la.cpp stamps), the bias and the reciprocal are one value per lane instead of sixteen, and the two wave halves of a
        //      64-row column meet ONCE per block: each normalises with its own maximum and the pair exchanges (max, sum) -- the online-softmax identity
        //      p = e^(q - m_w) e^(m_w - M) / (S_w e^(m_w - M) + S_p e^(m_p - M)) -- behind the barrier that publishes the block's M_b fragments.
        //      p goes back to the pixel-per-lane operand layout through a per-wave fp32 tile in LDS ([32 pixels][32 channels + 4]), is split into bf16 planes on the
        //      way out and feeds a += M_b p straight from registers.
        constexpr int NCW = (TH == 16) ? 2 : 1;  // image columns per wavefront
        constexpr int RPC = 16 / NCW;            // accumulator registers per column
        constexpr int TLD = 36;                  // floats per pixel row of the transpose tile (9 sixteen-byte slots: conflict-free float4 reads)
        float* Tw = Ap + wave * (32 * TLD);
#pragma unroll
        for (int nb = 0; nb < ((ABL & 16) ? 0 : NBQ); ++nb) {
            DDIF_SCHED_FENCE();
            float e[16], mx[NCW], sm[NCW];
            float* Cs = Cst + (nb & 1) * (NW * 32 * 2);  // [wave][j][max | sum]
            const float bj = BQ[32 * nb + j];
#pragma unroll
            for (int r = 0; r < 16; ++r) e[r] = fmaf(accq[nb][r], DDIF_F16_OSCALE, bj);
#pragma unroll
            for (int cg = 0; cg < NCW; ++cg) {
                float m = e[cg * RPC];
#pragma unroll
                for (int r = 1; r < RPC; ++r) m = fmaxf(m, e[cg * RPC + r]);
                m = fmaxf(m, __shfl_xor(m, 32));  // the other half's 8 (TH = 16) / 16 rows of the same column and channel
                float sacc = 0.f;
#pragma unroll
                for (int r = 0; r < RPC; ++r) {
                    const float ex = dd_exp2_fast((e[cg * RPC + r] - m) * L2E);
                    e[cg * RPC + r] = ex;
                    sacc += ex;
                }
                sacc += __shfl_xor(sacc, 32);
                mx[cg] = m;
                sm[cg] = sacc;
            }
            if constexpr (TH == 64) {  // a column spans the wave pair (2x, 2x + 1): (max, sum) of this half
                if (h == 0) *reinterpret_cast<float2*>(&Cs[(wave * 32 + j) * 2]) = make_float2(mx[0], sm[0]);
            }
            // first block: every wave is past the last chunk's fragment reads (Ap aliases Hs, Aq, Ax); every block: M_b fragments + the pair's statistics visible
            store_mix(nb & 1);
            if (nb + 1 < NBQ) load_mix(wmix_b, nb + 1);
            __syncthreads();
            float fac[NCW];
            if constexpr (TH == 64) {
                const float2 o = *reinterpret_cast<const float2*>(&Cs[((wave ^ 1) * 32 + j) * 2]);
                const float M = fmaxf(mx[0], o.x);
                const float cme = dd_exp2_fast((mx[0] - M) * L2E), cpt = dd_exp2_fast((o.x - M) * L2E);
                // (both waves of the pair form the same bits: even wave's term + odd wave's term)
                const float S = (wave & 1) ? (o.y * cpt + sm[0] * cme) : (sm[0] * cme + o.y * cpt);
                fac[0] = cme * dd_rcp_fast(S);
            } else {
#pragma unroll
                for (int cg = 0; cg < NCW; ++cg) fac[cg] = dd_rcp_fast(sm[cg]);
            }
            if (nb > 0) DDIF_WAVE_LDS_SYNC();  // the previous block's tile reads are done
#pragma unroll
            for (int r = 0; r < 16; ++r)  // pixel 8 g + 4 h + i of the wave, channel j
                Tw[((r >> 2) * 8 + 4 * h + (r & 3)) * TLD + j] = e[r] * fac[r / RPC];
            DDIF_WAVE_LDS_SYNC();
            const float* Fb = Fm + (nb & 1) * NPF * 256 + lane * 4;
#pragma unroll
            for (int k16 = 0; k16 < 2; ++k16) {
                // lane (j, h) as B operand: pixel j, channels 16 k16 + 8 h .. + 7 of the block
                const float4 t0 = *reinterpret_cast<const float4*>(&Tw[j * TLD + 16 * k16 + 8 * h]);
                const float4 t1 = *reinterpret_cast<const float4*>(&Tw[j * TLD + 16 * k16 + 8 * h + 4]);
                unsigned hh[4], mm[4], ll[4];
                dd_split3_pair(t0.x, t0.y, &hh[0], &mm[0], &ll[0]);
                dd_split3_pair(t0.z, t0.w, &hh[1], &mm[1], &ll[1]);
                dd_split3_pair(t1.x, t1.y, &hh[2], &mm[2], &ll[2]);
                dd_split3_pair(t1.z, t1.w, &hh[3], &mm[3], &ll[3]);
                float4 xp[3];
                xp[0] = make_float4(__builtin_bit_cast(float, hh[0]), __builtin_bit_cast(float, hh[1]), __builtin_bit_cast(float, hh[2]), __builtin_bit_cast(float, hh[3]));
                xp[1] = make_float4(__builtin_bit_cast(float, mm[0]), __builtin_bit_cast(float, mm[1]), __builtin_bit_cast(float, mm[2]), __builtin_bit_cast(float, mm[3]));
                xp[2] = make_float4(__builtin_bit_cast(float, ll[0]), __builtin_bit_cast(float, ll[1]), __builtin_bit_cast(float, ll[2]), __builtin_bit_cast(float, ll[3]));
#pragma unroll
                for (int na = 0; na < NBA; ++na) {
                    float4 f[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) f[pl] = *reinterpret_cast<const float4*>(&Fb[((na * 2 + k16) * 3 + pl) * 256]);
                    f32x16 c = acca[na];
                    if (ABL & 4) {
                        c[0] += f[2].x * xp[0].x + f[1].y * xp[1].y + f[0].z * xp[2].z;
                        acca[na] = c;
                        continue;
                    }
                    c = DDIF_MFMA_32x32x16_BF16(f[2], xp[0], c);
                    c = DDIF_MFMA_32x32x16_BF16(f[0], xp[2], c);
                    c = DDIF_MFMA_32x32x16_BF16(f[1], xp[1], c);
                    c = DDIF_MFMA_32x32x16_BF16(f[1], xp[0], c);
                    c = DDIF_MFMA_32x32x16_BF16(f[0], xp[1], c);
                    c = DDIF_MFMA_32x32x16_BF16(f[0], xp[0], c);
                    acca[na] = c;
                }
            }
        }
        DDIF_SCHED_FENCE();
        stamp();  // softmax + attn_out done
        // ---- epilogue: + bias, NHWC float4 stores
        {
            const int p = 32 * wave + j, x = x0 + p / TH, y = p % TH;
            if (x < a.W) {
                float* o = a.out + ((size_t)(b * a.H + y) * a.W + x) * a.dout;
#pragma unroll
                for (int na = 0; na < NBA; ++na)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int co = 32 * na + 8 * g + 4 * h;
                        if (co < a.dout) {
                            const float4 bo = *reinterpret_cast<const float4*>(&BO[co]);
                            if (!(ABL & 32) || acca[na][4 * g] == 12345.678f)
                                *reinterpret_cast<float4*>(o + co) =
                                    make_float4(acca[na][4 * g + 0] + bo.x, acca[na][4 * g + 1] + bo.y, acca[na][4 * g + 2] + bo.z, acca[na][4 * g + 3] + bo.w);
                        }
                    }
            }
        }
        __syncthreads();  // Ap / Fm (aliasing Hs / Aq / Ax / Wc) and Cst are free for the next strip
    }
}

}  // namespace ddif
