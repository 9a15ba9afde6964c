// Fused bottleneck self-attention block for 64-token tiles (the 8x8 level of a 64x64 tile), gfx950.
//
// Reference: SelfAttention.forward (models/sr3_dwt.py:341-360) = GroupNorm(1 group) -> 1x1 conv 128 -> 384 (no bias) ->
// per-head [q | k | v] split -> softmax(q k^T / sqrt(C)) v -> 1x1 conv 128 -> 128 + bias -> + input.
// As three launches (qkv conv, attention core, out conv) this block costs 49 us per instance at B = 64 and there are
// eight instances per denoising step; none of the three has more than 3 us of work.  Here ONE workgroup (4 wavefronts)
// owns a sample and keeps everything between the input and the output on chip:
//   1. xn = GroupNorm(x) staged into LDS as three bf16 planes (64 tokens x 128 channels);
//   2. qkv = xn . Wqkv^T on v_mfma_f32_32x32x16_bf16 with bf16x3 split products (wave w computes cout blocks 3w..3w+2,
//      weights straight from L2 into the MFMA operand registers, next slab prefetched), result to LDS as fp32;
//   3. attention, two heads per wave, exact fp32 on v_mfma_f32_16x16x4_f32 -- the algorithm of self_attn_mfma_kernel
//      (S^T = K Q^T so that the accumulator layout IS the A-operand layout of P V; max pass, then exp / sum / PV pass),
//      reading q, k, v from LDS; head hd's 16 output channels are exactly slab hd of the next contraction and are written
//      back to LDS as bf16 planes;
//   4. out = o . Wout^T + bias + x (wave w computes cout block w), float4 NHWC stores, GroupNorm partial of the output.
// Arithmetic per stage is that of the unfused kernels (bf16x3 for the two 1x1 convs, exact-fp32 MFMAs for the attention
// core) except that the softmax exponentials use v_exp_f32 (~1 ulp, as SiLU does everywhere) instead of expf.
// Measured (B = 64): 26 us per launch against 49 us for the three launches it replaces.  A three-slab weight ring and
// fetching all out-projection slabs before the attention core measured slower (33 us: 512 registers, 52 spilled).
// Round 6: NW = 8 wavefronts per sample instead of 4 (two per SIMD: with one, every LDS / L2 latency of the chain stood in front of an idle
// matrix pipe).  The (cout block, token block) pairs of the two 1x1 convs and the eight heads are dealt out over the waves -- wave w takes
// token block w & 1 of cout blocks 3 (w >> 1) .. + 2 (qkv) / w >> 1 (out) and head w -- so every product, every accumulation order and every
// stored value is the same as with four waves; only the order in which the per-wave statistics partials are added differs.
#pragma once
#include "kernels_conv.h"
#include "attn_args.h"

namespace ddif {

struct AttnBlockGeom {
    static constexpr int N = 64, C = 128, NSLAB = 8;
    static constexpr int APIX = NSLAB * 24 + 4;  // floats per token of the bf16-plane tile (8 slabs x 3 planes x 32 B + 16 B pad)
    static constexpr int QROW = 3 * C + 4;       // floats per token of the fp32 qkv tile
    static constexpr int AFL = N * APIX, QFL = N * QROW;
    static constexpr size_t smem = (size_t)(AFL + QFL + 16) * sizeof(float);
};

// ABL (tools/mbench_attn.cpp only): 1 = s_memtime stamps of thread 0 at the stage boundaries into a.dbg
// SPLIT = 2 / 4 (round 6): a sample on TWO / FOUR workgroups = CUs (4: sixteen query tokens each; the out-projection then runs on a half-used 32-token block).  The block is bound by the matrix pipe of its CU (profiles/r06/attn_block_stamps.txt) and B = 64
// samples occupy 64 of 256 CUs: workgroup (b, half) stages the whole sample and computes q, k, v of all 64 tokens (k and v of every token are needed by every query;
// q of the other half is the redundant sixth), then runs the attention core and the out-projection + residual for ITS 32 query tokens only -- 464 instead of 640
// matrix instructions per wave, no exchange between the two workgroups, one statistics partial each.  Every value is computed by the same instruction sequence as
// with one workgroup per sample.
template <int NW, int ABL = 0, int SPLIT = 1, bool F16Q = false>
__global__ __launch_bounds__(64 * NW) void attn_block_kernel(AttnBlockArgs a) {
    constexpr int QPL = F16Q ? 2 : 3;  // operand planes of the qkv contraction (f16x2: hi, lo; bf16x3: hi, mid, lo)
    static_assert(NW == 4 || NW == 8, "4 or 8 wavefronts per sample");
    static_assert(SPLIT == 1 || ((SPLIT == 2 || SPLIT == 4) && NW == 4), "token split: four wavefronts");
    constexpr int QT = 4 / SPLIT;     // 16-query tiles of a workgroup
    constexpr int MBO = (8 / NW) / SPLIT > 0 ? (8 / NW) / SPLIT : 1;  // token blocks per wave in the out-projection
    constexpr int MBW = 8 / NW;       // token blocks per wave in the two 1x1 convs (NW = 8: one, chosen by the wave's parity)
    constexpr int HPW = 8 / NW;       // heads per wave
    constexpr int TPP = 2 * NW;       // tokens per staging pass (32 float4 channel groups per token)
    constexpr int NIT = 64 / TPP;
    using G = AttnBlockGeom;
    constexpr int N = G::N, C = G::C, NSLAB = G::NSLAB, APIX = G::APIX, QROW = G::QROW, D = 16;
    dd_touch_kernargs<sizeof(AttnBlockArgs)>();  // every line of the argument block in ONE round trip (ddif_dev.h)
    DDIF_DYN_SMEM(smem);
    float* As = reinterpret_cast<float*>(smem);  // bf16 planes: xn, later o
    float* Qs = As + G::AFL;                     // fp32 qkv [token][384]
    float* Sst = Qs + G::QFL;

    const int tid = threadIdx.x, lane = tid & 63;
#ifdef DDIF_EMU
    const int wave = tid >> 6;
#else
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const int h = lane >> 5, j = lane & 31;
    const int c4 = tid & 31, p0 = tid >> 5;  // staging: 32 float4 channel groups x TPP tokens per pass
    [[maybe_unused]] int dbg_n = 0;
    auto stamp = [&]() {
#ifndef DDIF_EMU
        if ((ABL & 1) && a.dbg && tid == 0 && dbg_n < 31) a.dbg[blockIdx.x * 32 + dbg_n++] = (long long)__builtin_amdgcn_s_memtime();
#endif
    };
    stamp();  // 0: entry
    const int mb0 = (NW == 8) ? (wave & 1) : 0;   // first token block of this wave in the 1x1 convs
    const int cw = (NW == 8) ? (wave >> 1) : wave;  // its cout-block group

    const unsigned ga = gridDim.x;
    const int bw = (a.xcd && (ga & 7u) == 0u) ? (int)((blockIdx.x & 7u) * (ga >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;  // XCD-contiguous sample order
    for (int wk = bw; wk < a.B * SPLIT; wk += gridDim.x) {
        const int b = a.b0 + wk / SPLIT;
        const int half = wk % SPLIT;  // this workgroup's 64 / SPLIT query tokens: 16 QT half .. + 16 QT - 1
        // ---- (1) loads in one burst: GroupNorm partials, the sample, affine parameters, first weight slab
        GnPartials gp;
        gn_load_partials(a.st, a.np, nullptr, 0, b, &gp);
        float4 sv[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) sv[it] = *reinterpret_cast<const float4*>(a.x + ((size_t)b * N + p0 + it * TPP) * C + c4 * 4);
        const float4 gq = *reinterpret_cast<const float4*>(a.gamma + c4 * 4);
        const float4 bq = *reinterpret_cast<const float4*>(a.beta + c4 * 4);
        // qkv weights of this wave: cout blocks 3 cw .. 3 cw + 2, slab s, plane q at ((nb * 8 + s) * 3 + q) KiB
        // (wave-uniform base + 32-bit lane offset: the loads take the SGPR-base form; per-lane 64-bit addresses of all 72 fragments were precomputed and spilled)
        const char* wq = reinterpret_cast<const char*>(F16Q ? a.wqkv_f16 : a.wqkv) + (size_t)(3 * cw) * (NSLAB * QPL * 1024);
        unsigned lo16 = (unsigned)lane * 16u;
#ifndef DDIF_EMU
        asm volatile("" : "+v"(lo16));  // opaque per sample: keeps hipcc from hoisting the 96 fragment offsets (loop-invariant VGPR adds) out of the sample loop into spilled registers
#endif
        float4 wr[2][3][QPL];  // [ring slot][cout block][plane]
        auto load_qkv_w = [&](int slot, int s) {
#pragma unroll
            for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                for (int q = 0; q < QPL; ++q) wr[slot][nb][q] = *reinterpret_cast<const float4*>(wq + ((nb * NSLAB + s) * QPL + q) * 1024 + lo16);
        };
        load_qkv_w(0, 0);
        stamp();  // 1: first burst issued
        float mean, rstd;
        gn_reduce_partials(gp, a.st, a.np, nullptr, 0, b, (double)C * N, &mean, &rstd);
        stamp();  // 2: statistics reduced (the partials have arrived)
        {
            float ga[4], gb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ga[i] = (&gq.x)[i] * rstd;
                gb[i] = (&bq.x)[i] - mean * ga[i];
                if constexpr (F16Q) {  // the activation scale of the f16x2 split rides in the affine map
                    ga[i] *= DDIF_F16_ASCALE;
                    gb[i] *= DDIF_F16_ASCALE;
                }
            }
            const int slab = c4 >> 2, cin_slab = (c4 & 3) * 4;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int tok = p0 + it * TPP;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaf((&sv[it].x)[i], ga[i], gb[i]);
                float* d = &As[tok * APIX + slab * 24 + cin_slab / 2];  // (f16x2 uses two of the slab's three plane slots)
                if constexpr (F16Q) {
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
        stamp();  // 3: xn staged (the sample has arrived)
        __syncthreads();
        stamp();  // 4: barrier

        // ---- (2) qkv = xn . Wqkv^T : MBW token blocks x 3 cout blocks per wave, K = 8 slabs
        {
            f32x16 acc[MBW][3];
#pragma unroll
            for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
                for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
#pragma unroll
            for (int s = 0; s < NSLAB; ++s) {
                if (s + 1 < NSLAB) load_qkv_w((s + 1) & 1, s + 1);
                float4 xa[MBW][QPL];
#pragma unroll
                for (int q = 0; q < QPL; ++q)
#pragma unroll
                    for (int mb = 0; mb < MBW; ++mb) xa[mb][q] = *reinterpret_cast<const float4*>(&As[((mb0 + mb) * 32 + j) * APIX + s * 24 + q * 8 + 4 * h]);
                DDIF_SCHED_FENCE();
#pragma unroll
                for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                    for (int mb = 0; mb < MBW; ++mb) {
                        f32x16 c = acc[mb][nb];
                        const float4* w = wr[s & 1][nb];
                        if constexpr (F16Q) {
                            c = DDIF_MFMA_32x32x16_F16(w[1], xa[mb][0], c);  // lo * hi
                            c = DDIF_MFMA_32x32x16_F16(w[0], xa[mb][1], c);  // hi * lo
                            c = DDIF_MFMA_32x32x16_F16(w[0], xa[mb][0], c);  // hi * hi
                        } else {
                            c = DDIF_MFMA_32x32x16_BF16(w[QPL - 1], xa[mb][0], c);  // lo * hi
                            c = DDIF_MFMA_32x32x16_BF16(w[0], xa[mb][QPL - 1], c);  // hi * lo
                            c = DDIF_MFMA_32x32x16_BF16(w[1], xa[mb][1], c);  // mid * mid
                            c = DDIF_MFMA_32x32x16_BF16(w[1], xa[mb][0], c);  // mid * hi
                            c = DDIF_MFMA_32x32x16_BF16(w[0], xa[mb][1], c);  // hi * mid
                            c = DDIF_MFMA_32x32x16_BF16(w[0], xa[mb][0], c);  // hi * hi
                        }
                        acc[mb][nb] = c;
                    }
                DDIF_SCHED_FENCE();
            }
            // lane (j, h) owns token (mb0 + mb)*32 + j and, per quad g, couts nb*32 + 8g + 4h .. +3 (f16x2: the accumulator carries the two operand scales)
            constexpr float QS = F16Q ? DDIF_F16_OSCALE : 1.0f;
#pragma unroll
            for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
                for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<float4*>(&Qs[((mb0 + mb) * 32 + j) * QROW + (3 * cw + nb) * 32 + 8 * g + 4 * h]) =
                            make_float4(acc[mb][nb][4 * g + 0] * QS, acc[mb][nb][4 * g + 1] * QS, acc[mb][nb][4 * g + 2] * QS, acc[mb][nb][4 * g + 3] * QS);
        }
        // out-projection weights of this wave (cout block `cw`): first slab, in flight during the attention core
        const char* wo = reinterpret_cast<const char*>(a.wout) + (size_t)cw * (NSLAB * 3 * 1024);
        float4 wo_r[2][3];
        auto load_out_w = [&](int slot, int s) {
#pragma unroll
            for (int q = 0; q < 3; ++q) wo_r[slot][q] = *reinterpret_cast<const float4*>(wo + (s * 3 + q) * 1024 + lo16);
        };
        load_out_w(0, 0);
        stamp();  // 5: qkv contraction done and written
        __syncthreads();  // qkv complete in LDS; xn (As) is dead
        stamp();  // 6: barrier

        // ---- (3) attention core: heads HPW w .. + HPW - 1; n = 64 keys = one key block
        {
            const int jj = lane & 15, g4 = lane >> 4;
#pragma unroll 1
            for (int hh = 0; hh < HPW; ++hh) {
                const int hd = HPW * wave + hh;
                const float* base = Qs + hd * 3 * D;  // + token * QROW + {0, D, 2D} + d
                float4 qf[QT], kf[4];
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) kf[t4] = *reinterpret_cast<const float4*>(base + (t4 * 16 + jj) * QROW + D + 4 * g4);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) qf[qt] = *reinterpret_cast<const float4*>(base + ((half * QT + qt) * 16 + jj) * QROW + 4 * g4);
                f32x4 st[4][QT];  // [kt][qt]: S^T tiles; value r <-> key 16 kt + 4 g4 + r, query 16 (half QT + qt) + jj
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) {
                        f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) c = DDIF_MFMA_16x16x4((&kf[kt].x)[kk], (&qf[qt].x)[kk], c);
#pragma unroll
                        for (int r = 0; r < 4; ++r) c[r] *= a.scale;
                        st[kt][qt] = c;
                    }
                // softmax over keys = over the rows of S^T: 16 values per lane (4 key tiles x 4 regs) + xor-shuffles 16, 32.  The
                // tiles are overwritten with p = exp(s - max) (one v_exp_f32 per logit, reused by the P V pass)
                float inv[QT];
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    float bm = -INFINITY;
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) bm = fmaxf(bm, st[kt][qt][r]);
                    bm = fmaxf(bm, __shfl_xor(bm, 16));
                    bm = fmaxf(bm, __shfl_xor(bm, 32));
                    float ps = 0.f;
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float e = dd_exp2_fast((st[kt][qt][r] - bm) * 1.4426950408889634f);
                            st[kt][qt][r] = e;
                            ps += e;
                        }
                    ps += __shfl_xor(ps, 16);
                    ps += __shfl_xor(ps, 32);
                    inv[qt] = 1.f / ps;
                }
                f32x4 oacc[QT];
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) oacc[qt][r] = 0.f;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    float vf[4];  // B operand of P V: V[key 16 kt + 4 g4 + r][channel jj]
#pragma unroll
                    for (int r = 0; r < 4; ++r) vf[r] = base[(kt * 16 + 4 * g4 + r) * QROW + 2 * D + jj];
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) oacc[qt] = DDIF_MFMA_16x16x4(st[kt][qt][r] * inv[qt], vf[r], oacc[qt]);
                }
                // O tile: col = jj = channel hd*16 + jj, row = 4 g4 + r = query 16 (half QT + qt) + 4 g4 + r  -> slab hd of the o planes
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        unsigned q0, q1, q2;
                        dd_split3(oacc[qt][r], &q0, &q1, &q2);
                        unsigned short* d16 = reinterpret_cast<unsigned short*>(&As[((half * QT + qt) * 16 + 4 * g4 + r) * APIX + hd * 24]) + jj;
                        d16[0] = (unsigned short)q0;
                        d16[16] = (unsigned short)q1;  // next plane: + 8 floats = 16 halves
                        d16[32] = (unsigned short)q2;
                    }
            }
        }
        stamp();  // 7: attention core done
        __syncthreads();  // o planes complete
        stamp();  // 8: barrier

        // ---- (4) out = o . Wout^T + bias + x : cout block `cw`, MBO token blocks (SPLIT = 2: this workgroup's half), K = 8 slabs
        {
            const int mo0 = (SPLIT == 2) ? half : (SPLIT == 4 ? (half >> 1) : mb0);
            // SPLIT = 4: the out-projection runs on the 32-token block that holds this workgroup's 16 tokens (the other rows: stale finite planes, results dropped)
            const bool mine = (SPLIT != 4) || ((j >> 4) == (half & 1));
            f32x16 acc[MBO];
#pragma unroll
            for (int mb = 0; mb < MBO; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
            // epilogue operands: bias and the residual rows of this lane's tokens
            float4 bo[4], er[MBO][4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bo[g] = *reinterpret_cast<const float4*>(a.bout + cw * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int mb = 0; mb < MBO; ++mb) er[mb][g] = *reinterpret_cast<const float4*>(a.x + ((size_t)b * N + (mo0 + mb) * 32 + j) * C + cw * 32 + 8 * g + 4 * h);
            }
#pragma unroll
            for (int s = 0; s < NSLAB; ++s) {
                if (s + 1 < NSLAB) load_out_w((s + 1) & 1, s + 1);
                float4 xa[MBO][3];
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int mb = 0; mb < MBO; ++mb) xa[mb][q] = *reinterpret_cast<const float4*>(&As[((mo0 + mb) * 32 + j) * APIX + s * 24 + q * 8 + 4 * h]);
                DDIF_SCHED_FENCE();
#pragma unroll
                for (int mb = 0; mb < MBO; ++mb) {
                    f32x16 c = acc[mb];
                    const float4* w = wo_r[s & 1];
                    c = DDIF_MFMA_32x32x16_BF16(w[2], xa[mb][0], c);
                    c = DDIF_MFMA_32x32x16_BF16(w[0], xa[mb][2], c);
                    c = DDIF_MFMA_32x32x16_BF16(w[1], xa[mb][1], c);
                    c = DDIF_MFMA_32x32x16_BF16(w[1], xa[mb][0], c);
                    c = DDIF_MFMA_32x32x16_BF16(w[0], xa[mb][1], c);
                    c = DDIF_MFMA_32x32x16_BF16(w[0], xa[mb][0], c);
                    acc[mb] = c;
                }
                DDIF_SCHED_FENCE();
            }
            stamp();  // 9: out contraction issued
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int mb = 0; mb < MBO; ++mb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = (acc[mb][4 * g + i] + (&bo[g].x)[i]) + (&er[mb][g].x)[i];
                    if (mine) {
                        *reinterpret_cast<float4*>(a.out + ((size_t)b * N + (mo0 + mb) * 32 + j) * C + cw * 32 + 8 * g + 4 * h) = make_float4(v[0], v[1], v[2], v[3]);
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
        }
        stamp();  // 10: epilogue stores issued
        __syncthreads();  // (also: As / Qs are free for the next sample)
        stamp();  // 11: barrier
        if (a.st_out && tid == 0) {
            double t0 = ((double)Sst[0] + (double)Sst[2]) + ((double)Sst[4] + (double)Sst[6]);
            double t1 = ((double)Sst[1] + (double)Sst[3]) + ((double)Sst[5] + (double)Sst[7]);
            if (NW == 8) {
                t0 += ((double)Sst[8] + (double)Sst[10]) + ((double)Sst[12] + (double)Sst[14]);
                t1 += ((double)Sst[9] + (double)Sst[11]) + ((double)Sst[13] + (double)Sst[15]);
            }
            a.st_out[((size_t)b * SPLIT + half) * 2 + 0] = t0;  // (one partial per workgroup: np = SPLIT)
            a.st_out[((size_t)b * SPLIT + half) * 2 + 1] = t1;
        }
    }
}

}  // namespace ddif
