// The attention half of the decoder's FastAttnCondInjection at the 8 x 8 level as ONE kernel (inference plans; round 6).
//
// Same reference ops and the same arithmetic as kernels_lafuse.h (models/sr3_dwt.py:536-566: xn = prenorm_x(cat[h, skip]); q = q.1(q.0(xn)) on f16x2
// split products; p = softmax over image ROWS of q; a = M_b p + W_res xn + bias on bf16x3 split products with the per-sample folded weights
// M_b = scale * W_out * blockdiag(ctx_b^T)), a different decomposition for the regime of 64 pixels x 192-256 channels per sample:
//
//   Until round 6 this level ran three launches per block -- gn_dw3x3_small (xn, dwq to memory), the q.1 1x1 conv of the low-resolution kernel with the
//   column statistics in its epilogue, and the 1x1 conv over cat[softmax(q), xn] -- 46-59 us per block at B = 64 and 30-39 us at B = 8: three cold-start chains
//   for 75 MFLOP of issued matrix work per sample.  kernels_lafuse.h cannot take it: its waves split the PIXELS (256 per workgroup) and keep every output
//   channel in registers; here there are only 64 pixels and up to 256 + 128 output channels.
//
//   A workgroup (8 wavefronts) owns HALF a sample: 4 image columns x 8 rows = 32 pixels = one MFMA pixel block (the column softmax stays local, the
//   depthwise conv needs a one-pixel halo: 6 x 10 staged pixels), so a sample runs on TWO CUs and B = 64 fills 128 of them.  The waves split the OUTPUT channels:
//     chunk loop (64 input channels at a time):  raw halo tile (registers, prefetched one chunk ahead) -> GroupNorm -> Hs (fp32, LDS) | barrier |
//         depthwise 3x3 -> dwq as two half planes, centre of Hs -> xn as three bf16 planes | barrier |
//         wave w:  acc_q[32 pixels x q channels 32 w ..]  += W_q[w][chunk] dwq      (3 products x 4 slabs)
//                  acc_a[32 pixels x out channels 32 (w & 3) ..] += W_res[w & 3][slabs 2 (w >> 2), + 1 of the chunk] xn   (6 products x 2 slabs)
//     the B fragments (weights) are not shared between waves and never touch LDS: each wave reads its own pre-packed 1 KiB pieces straight from L2 into the
//     MFMA operand registers, requested at the top of the chunk and consumed behind its two staging stages;
//     softmax over the 8 rows of a column = over 8 neighbouring lanes (pixels are numbered column-major): three DPP steps per value; p -> LDS as bf16 planes;
//     wave w:  acc_a += M_b[w & 3][K half w >> 2] p ;  the two K halves of an output block meet in LDS ((w < 4) + (w >= 4)), + bias, float4 NHWC stores.
#pragma once
#include "kernels_lafuse.h"

namespace ddif {

template <int NBQ>
struct LaFuse8Geom {
    static constexpr int FEA = 32 * NBQ, NCHK = FEA / 64, NS = FEA / 16;  // channels, 64-channel chunks, 16-channel slabs
    static constexpr int HW = 6, HH = 10, NHP = HW * HH;                  // halo tile: 6 columns x 10 rows
    static constexpr int LDH = 68;                                        // floats per halo pixel: 64 channels + 16 B pad
    static constexpr int LDQ = 68;                                        // dwq: 4 slabs x 2 half planes x 32 B + 16 B pad
    static constexpr int LDX = 100;                                       // xn:  4 slabs x 3 bf16 planes x 32 B + 16 B pad
    static constexpr int LDP = NS * 24 + 4;                               // p:   NS slabs x 3 bf16 planes x 32 B + 16 B pad (odd in 16-byte slots: conflict-free fragment reads)
    static constexpr int HS = NHP * LDH, AQ = 32 * LDQ, AX = 32 * LDX, AP = 32 * LDP;
    static constexpr int MAIN = (HS + AQ + AX) > AP ? (HS + AQ + AX) : AP;  // Ap aliases Hs | Aq | Ax once the chunk loop is over
    static constexpr int RED = 4 * 64 * 16;                                 // K-half exchange of the four output blocks
    static constexpr int TAB = 2 * FEA + 9 * FEA + FEA + 128;               // gamma | beta | depthwise | q bias | output bias
    static constexpr size_t smem = (size_t)(MAIN + RED + TAB) * sizeof(float);
};

// max / sum over the 8 lanes 8 m .. 8 m + 7 (one image column of 8 rows), result in every lane of the group: quad_perm xor 1, xor 2, row_half_mirror
__device__ __forceinline__ float oct_allmax(float v) {
#ifdef DDIF_EMU
#pragma unroll
    for (int m = 1; m <= 4; m <<= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
#else
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false)));   // quad_perm [1,0,3,2]
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false)));   // quad_perm [2,3,0,1]
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false)));  // row_half_mirror: lane l <- lane 7 - l of its 8
    return v;
#endif
}
__device__ __forceinline__ float oct_allsum(float v) {
#ifdef DDIF_EMU
#pragma unroll
    for (int m = 1; m <= 4; m <<= 1) v += __shfl_xor(v, m);
    return v;
#else
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
    return v;
#endif
}

// LaFuseArgs as for linattn_fused_kernel with H = W = 8, c0 % 64 == 0, c1 % 64 == 0, c0 + c1 = 32 NBQ, dout = 128.
// ABL (tools/mbench_la8.cpp only): 64 = s_memtime stamps of thread 0 into a.dbg, 128 = six more per 64-channel chunk
template <int NBQ, int ABL = 0>
__global__ __launch_bounds__(512) void linattn8_fused_kernel(LaFuseArgs a) {
    using G = LaFuse8Geom<NBQ>;
    constexpr int FEA = G::FEA, NCHK = G::NCHK, NS = G::NS, HW = G::HW, HH = G::HH, NHP = G::NHP;
    constexpr int LDH = G::LDH, LDQ = G::LDQ, LDX = G::LDX, LDP = G::LDP;
    constexpr int NSH = NS / 2;       // slabs of one K half of the M_b contraction
    constexpr int NR0 = NSH / 2, NR1 = NSH - NR0;  // ... fetched in two rounds
    constexpr float L2E = 1.4426950408889634f;
    static_assert(NBQ == 8 || NBQ == 6, "256 or 192 feature channels");

    dd_touch_kernargs<sizeof(LaFuseArgs)>();  // every line of the argument block in ONE round trip (ddif_dev.h)
    DDIF_DYN_SMEM(smem);
    float* Hs = reinterpret_cast<float*>(smem);
    float* Aq = Hs + G::HS;
    float* Ax = Aq + G::AQ;
    float* Ap = Hs;                 // after the chunk loop
    float* Red = Hs + G::MAIN;      // [4 output blocks][64 lanes][16]
    float* GB = Red + G::RED;       // gamma [FEA] | beta [FEA]
    float* DW = GB + 2 * FEA;       // [9][FEA]
    float* BQ = DW + 9 * FEA;       // [FEA]
    float* BO = BQ + FEA;           // [128]

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
        if ((ABL & 64) && a.dbg && tid == 0 && dbg_n < 63) a.dbg[blockIdx.x * 64 + dbg_n++] = (long long)__builtin_amdgcn_s_memtime();
#endif
    };
    stamp();
    const int nwork = a.B * 2;  // (sample, half)
    int w0, w1;
    wg_work_range(nwork, &w0, &w1, a.xcd);
    if (w0 >= w1) return;

    const int c4 = tid & 15;          // channel quad of the 64-channel chunk (both staging items and the depthwise item of a thread)
    const int nr = wave & 3, kh = wave >> 2;  // output block / K half of this wave in the attn_res and M_b contractions
    const bool qw = wave < NBQ;       // this wave owns q block `wave`

    // raw staging: item = tid + 512 it -> halo pixel item >> 4 (row-major in the 6-wide halo tile), channel quad c4; clamped addresses, validity as a mask
    float4 raw[2];
    unsigned rok = 0;
    auto load_raw = [&](int b, int x0, int k) {
        const int cb = 64 * k;
        const bool s0 = cb < a.c0;
        const float* src = s0 ? a.in0 + cb + 4 * c4 : a.in1 + (cb - a.c0) + 4 * c4;
        const int cs = s0 ? a.c0 : a.c1;
        rok = 0;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int pix = (tid + it * 512) >> 4;
            const bool in = pix < NHP;
            const int pc = in ? pix : NHP - 1;
            const int y = pc / HW - 1, x = x0 + pc % HW - 1;
            const bool ok = in & (y >= 0) & (y < 8) & (x >= 0) & (x < 8);
            rok |= (ok ? 1u : 0u) << it;
            const int yc = y < 0 ? 0 : (y > 7 ? 7 : y), xc = x < 0 ? 0 : (x > 7 ? 7 : x);
            raw[it] = *reinterpret_cast<const float4*>(src + ((size_t)(b * 8 + yc) * 8 + xc) * cs);
        }
    };

    int gn_b = -1;
    float mean = 0.f, rstd = 1.f;
    GnPartials gp0;
    {
        const int b = a.b0 + (w0 >> 1);
        gn_load_partials(a.st0, a.np0, a.st1, a.np1, b, &gp0);
        load_raw(b, (w0 & 1) * 4, 0);
    }
    // tables (once per launch), requested behind the first tile's loads: one round trip on the cold caches of a launch
    // (every table load is issued before any is stored: the load -> store loops of the first form were one dependent round trip per iteration, five in a row
    //  for the depthwise table -- 3.4 us of a 20 us launch with warm caches, tools/mbench_la8.cpp)
    {
        constexpr int NDW = (9 * FEA + 511) / 512;
        const int ic = tid < FEA ? tid : FEA - 1, io = tid < 128 ? tid : 127;
        const float tg = a.gamma[ic], tb = a.beta[ic], tq = a.bq[ic], to = a.bias[io];
        float tdw[NDW];
#pragma unroll
        for (int k = 0; k < NDW; ++k) {
            const int i = tid + k * 512;
            tdw[k] = a.dw_w[i < 9 * FEA ? i : 9 * FEA - 1];
        }
        if (tid < FEA) {
            GB[tid] = tg;
            GB[FEA + tid] = tb;
            BQ[tid] = tq;
        }
        if (tid < 128) BO[tid] = to;
#pragma unroll
        for (int k = 0; k < NDW; ++k) {
            const int i = tid + k * 512;
            if (i < 9 * FEA) DW[i] = tdw[k];
        }
    }
    {
        const int b = a.b0 + (w0 >> 1);
        gn_reduce_partials(gp0, a.st0, a.np0, a.st1, a.np1, b, (double)FEA * 64, &mean, &rstd);
        gn_b = b;
    }
    __syncthreads();  // tables
    stamp();

    for (int work = w0; work < w1; ++work) {
        const int b = a.b0 + (work >> 1), x0 = (work & 1) * 4;
        if (b != gn_b) {  // workgroup-uniform; every wavefront reduces the producers' partials itself
            gn_finalize_wave(a.st0, a.np0, a.st1, a.np1, b, (double)FEA * 64, &mean, &rstd);
            gn_b = b;
        }
        f32x16 accq, acca;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            accq[r] = 0.f;
            acca[r] = 0.f;
        }
        const float* wmix_b = a.wmix + (size_t)b * a.wmix_bstride;
        const float* wq_w = a.wq + (size_t)(qw ? wave : 0) * a.nchq * (2 * 2 * 256);  // (waves without a q block re-read block 0; never consumed)
        const float* wr_w = wmix_b + ((size_t)nr * a.nch_mix + NBQ + kh) * (2 * 3 * 256);
        // this wave's weight pieces (1 KiB each, lane-linear): wave-uniform base + 32-bit lane offset
        unsigned lo4 = (unsigned)lane * 4u;
#ifndef DDIF_EMU
        asm volatile("" : "+v"(lo4));  // opaque per item: keeps hipcc from hoisting the piece offsets (loop-invariant VGPR adds) out of the work loop
#endif

#pragma unroll 1
        for (int k = 0; k < NCHK; ++k) {
            // (a) this chunk's B fragments: requested now, consumed in stage (d)
            // (one wave-uniform base per chunk and constant piece offsets: the per-piece index expressions of the first form cost ~1 k ticks of address
            //  arithmetic per chunk before the fourteen loads were out, tools/mbench_la8.cpp)
            float4 wqr[4][2], wrr[2][3];
            const float* wqk = wq_w + (size_t)k * (2 * 2 * 2 * 256);   // this wave's q block: two 32-channel chunks x two k16 x two planes per 64-channel chunk
            const float* wrk = wr_w + (size_t)k * (2 * 2 * 3 * 256);   // this wave's output block / K half: chunk NBQ + 2 k + kh
#pragma unroll
            for (int sl = 0; sl < 4; ++sl)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) wqr[sl][pl] = *reinterpret_cast<const float4*>(wqk + ((sl >> 1) * 4 + (sl & 1) * 2 + pl) * 256 + lo4);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) wrr[t][pl] = *reinterpret_cast<const float4*>(wrk + (t * 3 + pl) * 256 + lo4);
            if (ABL & 128) stamp();  // c0: chunk's fragment loads issued
            // (b) GroupNorm of the raw halo tile -> Hs (zero padding comes after the normalisation: the depthwise conv pads xn)
            {
                const float4 gq = *reinterpret_cast<const float4*>(&GB[64 * k + 4 * c4]);
                const float4 bq4 = *reinterpret_cast<const float4*>(&GB[FEA + 64 * k + 4 * c4]);
                float ga[4], gb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ga[i] = (&gq.x)[i] * rstd;
                    gb[i] = (&bq4.x)[i] - mean * ga[i];
                }
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const bool ok = (rok >> it) & 1u;
                    const int pix = (tid + it * 512) >> 4;
                    float4 v;
                    v.x = ok ? fmaf(raw[it].x, ga[0], gb[0]) : 0.f;
                    v.y = ok ? fmaf(raw[it].y, ga[1], gb[1]) : 0.f;
                    v.z = ok ? fmaf(raw[it].z, ga[2], gb[2]) : 0.f;
                    v.w = ok ? fmaf(raw[it].w, ga[3], gb[3]) : 0.f;
                    if (pix < NHP) *reinterpret_cast<float4*>(&Hs[pix * LDH + 4 * c4]) = v;
                }
            }
            if (ABL & 128) stamp();  // c1: GroupNorm stage done (includes the wait for the raw tile)
            // the next chunk's (or the next item's first) raw tile flies during the stages below
            if (k + 1 < NCHK) {
                load_raw(b, x0, k + 1);
            } else {
                const int wn = work + 1 < w1 ? work + 1 : work;
                load_raw(a.b0 + (wn >> 1), (wn & 1) * 4, 0);
            }
            __syncthreads();
            if (ABL & 128) stamp();  // c2: barrier
            // (c) depthwise 3x3 of the chunk: one (pixel, channel quad) per thread; pixels are numbered column-major, p = 8 xl + y
            {
                const int p = tid >> 4, xl = p >> 3, y = p & 7;
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
                float4 cen = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int ty = 0; ty < 3; ++ty) {
                    float4 hv[3], wk[3];
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx) {
                        hv[tx] = *reinterpret_cast<const float4*>(&Hs[((y + ty) * HW + xl + tx) * LDH + 4 * c4]);
                        wk[tx] = *reinterpret_cast<const float4*>(&DW[(3 * ty + tx) * FEA + 64 * k + 4 * c4]);
                    }
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx) {
                        s0 = fmaf(hv[tx].x, wk[tx].x, s0);
                        s1 = fmaf(hv[tx].y, wk[tx].y, s1);
                        s2 = fmaf(hv[tx].z, wk[tx].z, s2);
                        s3 = fmaf(hv[tx].w, wk[tx].w, s3);
                    }
                    if (ty == 1) cen = hv[1];
                }
                const int sl = c4 >> 2, e2 = (c4 & 3) * 2;  // slab of the chunk, float offset of this quad's 4 halves inside a plane
                unsigned h01, l01, h23, l23, m01, m23;
                dd_split2_pair(s0 * DDIF_F16_ASCALE, s1 * DDIF_F16_ASCALE, &h01, &l01);
                dd_split2_pair(s2 * DDIF_F16_ASCALE, s3 * DDIF_F16_ASCALE, &h23, &l23);
                *reinterpret_cast<uint2*>(&Aq[p * LDQ + sl * 16 + e2]) = make_uint2(h01, h23);
                *reinterpret_cast<uint2*>(&Aq[p * LDQ + sl * 16 + 8 + e2]) = make_uint2(l01, l23);
                dd_split3_pair(cen.x, cen.y, &h01, &m01, &l01);
                dd_split3_pair(cen.z, cen.w, &h23, &m23, &l23);
                *reinterpret_cast<uint2*>(&Ax[p * LDX + sl * 24 + e2]) = make_uint2(h01, h23);
                *reinterpret_cast<uint2*>(&Ax[p * LDX + sl * 24 + 8 + e2]) = make_uint2(m01, m23);
                *reinterpret_cast<uint2*>(&Ax[p * LDX + sl * 24 + 16 + e2]) = make_uint2(l01, l23);
            }
            if (ABL & 128) stamp();  // c3: depthwise + split stage done
            __syncthreads();
            if (ABL & 128) stamp();  // c4: barrier
            // (d) contraction of the chunk: 32 pixels x this wave's 32 q channels (4 slabs) and x its output block (2 of the 4 slabs)
            if (qw) {
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) {
                    const float4 x0q = *reinterpret_cast<const float4*>(&Aq[j * LDQ + sl * 16 + 4 * h]);
                    const float4 x1q = *reinterpret_cast<const float4*>(&Aq[j * LDQ + sl * 16 + 8 + 4 * h]);
                    accq = DDIF_MFMA_32x32x16_F16(wqr[sl][1], x0q, accq);  // lo * hi
                    accq = DDIF_MFMA_32x32x16_F16(wqr[sl][0], x1q, accq);  // hi * lo
                    accq = DDIF_MFMA_32x32x16_F16(wqr[sl][0], x0q, accq);  // hi * hi
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int sl = 2 * kh + t;
                float4 xr[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) xr[pl] = *reinterpret_cast<const float4*>(&Ax[j * LDX + sl * 24 + pl * 8 + 4 * h]);
                acca = DDIF_MFMA_32x32x16_BF16(wrr[t][2], xr[0], acca);
                acca = DDIF_MFMA_32x32x16_BF16(wrr[t][0], xr[2], acca);
                acca = DDIF_MFMA_32x32x16_BF16(wrr[t][1], xr[1], acca);
                acca = DDIF_MFMA_32x32x16_BF16(wrr[t][1], xr[0], acca);
                acca = DDIF_MFMA_32x32x16_BF16(wrr[t][0], xr[1], acca);
                acca = DDIF_MFMA_32x32x16_BF16(wrr[t][0], xr[0], acca);
            }
            if (ABL & 128) stamp();  // c5: contraction issued (includes the wait for the fragments)
            // (no barrier here: the next chunk's GroupNorm stage writes Hs, which nobody reads any more; Aq / Ax are rewritten behind its barrier)
        }
        stamp();

        // ---- M_b fragments of this wave's K half, first round: in flight during the softmax
        float4 wm0[NR0][3], wm1[NR1][3];
        auto mix_piece = [&](int sl, int pl) {  // slab sl of all NS, plane pl
            return *reinterpret_cast<const float4*>(wmix_b + ((size_t)((((nr * a.nch_mix + (sl >> 1)) * 2 + (sl & 1)) * 3 + pl)) * 256) + lo4);
        };
#pragma unroll
        for (int i = 0; i < NR0; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wm0[i][pl] = mix_piece(kh * NSH + i, pl);
        // ... and the second round with it: the per-sample folded weights come from HBM / Infinity Cache at B = 64 (25 MB per launch), requested behind the barrier
        // below they stood in front of the contraction (9.6 k ticks at B = 64 against 4.8 k at B = 8, tools/mbench_la8.cpp); the chunk loop's fragments are dead here
#pragma unroll
        for (int i = 0; i < NR1; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wm1[i][pl] = mix_piece(kh * NSH + NR0 + i, pl);

        // ---- q complete: softmax over the 8 rows of every column.  Lane (j, h) of wave w holds channels 32 w + 8 g + 4 h + i of pixel j = 8 xl + y: the lanes
        //      8 xl .. 8 xl + 7 are ONE column
        float e[16];
        if (qw) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bq4 = *reinterpret_cast<const float4*>(&BQ[32 * wave + 8 * g + 4 * h]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float q = fmaf(accq[4 * g + i], DDIF_F16_OSCALE, (&bq4.x)[i]);
                    const float m = oct_allmax(q);
                    const float ex = dd_exp2_fast((q - m) * L2E);
                    e[4 * g + i] = ex * dd_rcp_fast(oct_allsum(ex));
                }
            }
        }
        __syncthreads();  // every wave is past the last chunk's fragment reads: Ap (aliasing Hs | Aq | Ax) may be written
        if (qw) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // channels 32 w + 8 g + 4 h + i  ->  slab 2 w + g / 2, k half g % 2, elements 4 h + i
                unsigned h01, m01, l01, h23, m23, l23;
                dd_split3_pair(e[4 * g + 0], e[4 * g + 1], &h01, &m01, &l01);
                dd_split3_pair(e[4 * g + 2], e[4 * g + 3], &h23, &m23, &l23);
                float* d = &Ap[j * LDP + (2 * wave + (g >> 1)) * 24 + (g & 1) * 4 + 2 * h];
                *reinterpret_cast<uint2*>(d) = make_uint2(h01, h23);
                *reinterpret_cast<uint2*>(d + 8) = make_uint2(m01, m23);
                *reinterpret_cast<uint2*>(d + 16) = make_uint2(l01, l23);
            }
        }
        __syncthreads();  // p complete
        stamp();
        // ---- acc_a += M_b[output block nr][K half kh] p
        auto mix_step = [&](int sl, const float4* f) {
            float4 xp[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) xp[pl] = *reinterpret_cast<const float4*>(&Ap[j * LDP + sl * 24 + pl * 8 + 4 * h]);
            acca = DDIF_MFMA_32x32x16_BF16(f[2], xp[0], acca);
            acca = DDIF_MFMA_32x32x16_BF16(f[0], xp[2], acca);
            acca = DDIF_MFMA_32x32x16_BF16(f[1], xp[1], acca);
            acca = DDIF_MFMA_32x32x16_BF16(f[1], xp[0], acca);
            acca = DDIF_MFMA_32x32x16_BF16(f[0], xp[1], acca);
            acca = DDIF_MFMA_32x32x16_BF16(f[0], xp[0], acca);
        };
#pragma unroll
        for (int i = 0; i < NR0; ++i) mix_step(kh * NSH + i, wm0[i]);
#pragma unroll
        for (int i = 0; i < NR1; ++i) mix_step(kh * NSH + NR0 + i, wm1[i]);
        stamp();
        // ---- the two K halves of an output block meet in LDS; waves 0-3 finish: (kh = 0) + (kh = 1) + bias, NHWC float4 stores
        if (kh == 1) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(&Red[((nr * 64 + lane) * 4 + g) * 4]) = make_float4(acca[4 * g + 0], acca[4 * g + 1], acca[4 * g + 2], acca[4 * g + 3]);
        }
        __syncthreads();
        if (kh == 0) {
            const int xl = j >> 3, y = j & 7;
            float* o = a.out + ((size_t)(b * 8 + y) * 8 + x0 + xl) * a.dout + 32 * nr + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 o1 = *reinterpret_cast<const float4*>(&Red[((nr * 64 + lane) * 4 + g) * 4]);
                const float4 bo = *reinterpret_cast<const float4*>(&BO[32 * nr + 8 * g + 4 * h]);
                *reinterpret_cast<float4*>(o + 8 * g) = make_float4((acca[4 * g + 0] + o1.x) + bo.x, (acca[4 * g + 1] + o1.y) + bo.y, (acca[4 * g + 2] + o1.z) + bo.z,
                                                                   (acca[4 * g + 3] + o1.w) + bo.w);
            }
        }
        stamp();
        // (no barrier: the next item's first LDS writes -- Hs, aliasing Ap -- come after every wave's fragment reads of this item (they precede the barrier above);
        //  Red is rewritten only behind the next item's barriers)
    }
}

}  // namespace ddif
