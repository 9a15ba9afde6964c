// MEASURED AND NOT ADOPTED (profiles/r05/c_resblock_resident_gate.txt; kept under tools/ with its micro-benchmark, not part of the product).
// Resident ResnetBlock of the 8 x 8 level (round 5): GroupNorm -> SiLU -> conv3x3 + time bias -> GroupNorm -> SiLU -> conv3x3 + residual
// in ONE launch, one workgroup per sample, the whole sample on chip between the two convolutions.
//
// Replaces two launches of conv_lr_kernel (kernels_lr.h) per reference ResnetBlock (models/sr3_dwt.py:303-327: block1 = Block(dim, dim_out)
// :288-300, noise_func = FeatureWiseAffine :241-258 (a per-channel bias here), block2, res_conv = Identity at these levels) where a sample
// is 64 pixels x 128 channels = 32 KB: GroupNorm(1 group) needs the statistics of the WHOLE sample, which is what forces a kernel boundary
// behind every conv at the higher levels -- here the sample never leaves the workgroup, so the second GroupNorm's statistics are a
// workgroup-local reduction and the intermediate tensor h never touches HBM.
//
//   * 8 wavefronts: wave = (cout block cb = wave >> 1, K half kh = wave & 1).  A wave contracts the 16-channel slabs 4 kh .. 4 kh + 3 (x 9
//     taps) of BOTH 32-pixel blocks of the sample against the 32 couts of block cb: 216 v_mfma_f32_32x32x16_f16 per conv and wave (f16x2
//     split products, ddif_dev.h), 13.8 K matrix cycles per SIMD = 5.8 us per conv -- the floor of this form; the two launches it replaces
//     take 14-16 us each at B = 64 for 1.5 us of matrix work on four workgroups per sample.
//   * weights stream L2 -> MFMA operand registers through a ring of one slab (9 taps x 2 planes = 72 VGPRs), never through LDS; every
//     weight byte is read once per workgroup; the ring runs on across the boundary between the two convs (conv2's first slab is in flight
//     while conv1's K-partials are reduced and GroupNorm 2 is formed).
//   * the staged tile is the 10 x 10 halo image of the sample as two half planes per 16-channel slab (528 B per pixel, row stride = 8 mod 16
//     sixteen-byte slots: conflict-free ds_read_b128 A fragments for the 8-wide tile, bank enumeration in DESIGN section 3).
//   * the two K-partials of a (cout block, pixel block) meet in LDS in a fixed order (kh 0 + kh 1); conv1's epilogue (x 2^-14, + bias + time
//     bias) leaves h as fp32 in LDS and its sum / sum of squares; conv2's adds the residual (re-read from L2) and writes the output and its
//     ONE GroupNorm partial per sample.
// Arithmetic per element is that of conv_lr_kernel<3, 2, PRO_GN_SILU, ., ., F16>: same staging expressions, same products; only the K
// summation tree differs (two partials instead of four), so results agree to fp32 rounding, not bitwise.  Every value depends on its own
// sample only: tiles of a batch are bit-equal to single-tile runs.
#pragma once
#include "ddif_dev.h"
#include "rb_args.h"

namespace ddif {

struct RbGeom {
    static constexpr int C = 128, HW = 8, NPIX = 64;
    static constexpr int NS = C / 16;              // 16-channel slabs
    static constexpr int NCB = C / 32;             // 32-cout blocks
    static constexpr int MB = 2;                   // 32-pixel blocks of a sample
    static constexpr int IW = HW + 2;              // halo image
    static constexpr int APIX = NS * 16 + 4;       // floats per staged pixel: NS x (hi 32 B | lo 32 B) + 16 B pad  (33 slots = 1 mod 16)
    static constexpr int AROW = IW * APIX + 56;    // floats per staged row: 344 slots = 8 mod 16
    static constexpr int AFL = IW * AROW;
    static constexpr int RFL = 8 * MB * 4 * 64 * 4;  // K-partials [wave][mb][quad g][lane] float4
    static constexpr int REG0 = AFL > RFL ? AFL : RFL;
    static constexpr int HPIX = C + 4;             // floats per pixel of the fp32 intermediate h (33 slots)
    static constexpr int HFL = NPIX * HPIX;
    static constexpr size_t smem = (size_t)(REG0 + HFL + 32) * sizeof(float) + 16 * sizeof(double);
};

// ABL (tools/mbench_rb.cpp only): 1 = no weight loads, 2 = no MFMAs
template <int ABL = 0>
__global__ __launch_bounds__(512) void resblock8_kernel(RbArgs a) {
    using G = RbGeom;
    constexpr int C = G::C, NPIX = G::NPIX, MB = G::MB, IW = G::IW, APIX = G::APIX, AROW = G::AROW, HPIX = G::HPIX;
    constexpr int TAPS = 9, NPL = 2, WSTEP = NPL * 1024;  // bytes of one (slab, tap) step of the packed weights
    constexpr int SPW = 4;                                // slabs per wave and conv (K half)
    constexpr int NSTEP = SPW * TAPS;                     // ring steps per conv
    constexpr int NIT = 7;                                // staging items per thread: 100 halo pixels / 16 pixels per pass
    DDIF_DYN_SMEM(smem);
    float* As = reinterpret_cast<float*>(smem);
    float* Red = As;                 // K-partial exchange, reuses the tile after a conv's last MFMA (barriers in between)
    float* Hs = As + G::REG0;        // fp32 h = conv1 output, [64][HPIX]
    float* Sst = Hs + G::HFL;        // [8 waves][2] statistics partials (fp32 wave sums)

    const int tid = threadIdx.x, lane = tid & 63;
#ifdef DDIF_EMU
    const int wave = tid >> 6;
#else
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const int h = lane >> 5, j = lane & 31;
    const int cb = wave >> 1, kh = wave & 1;
    const int c4 = tid & 31, p0 = tid >> 5;  // staging: channel quad, first pixel of the pass
    const float* tbrow = a.tbias + (a.step_ptr ? (size_t)(*a.step_ptr) * a.tb_rowstride : 0);

    // A-fragment base of this lane for pixel block mb: pixel m = 32 mb + j of the sample (+ tap offset later)
    int abase[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int m = mb * 32 + j;
        abase[mb] = (m / 8) * AROW + (m % 8) * APIX + 4 * h;
    }
    // staging geometry: halo pixel pix = p0 + 16 it -> (py, px) in the 10 x 10 image; interior pixels are sample pixels.  Recomputed where it is
    // used (a few integer operations) instead of being held in 15 registers across the matrix phases.
    auto s_geom = [&](int it, int* src, int* lds) -> bool {
        const int pix = p0 + 16 * it;
        const int py = pix / IW, px = pix - py * IW;
        const bool real = pix < IW * IW;
        const bool inner = real && py >= 1 && py <= 8 && px >= 1 && px <= 8;
        *src = inner ? (py - 1) * 8 + (px - 1) : 0;
        *lds = real ? py * AROW + px * APIX + (c4 >> 2) * 16 + (c4 & 3) * 2 : -1;
        return inner;
    };
    // epilogue share of this wave: cout block cb, accumulator quads g = 2 kh, 2 kh + 1, both pixel blocks
    const int e_co0 = cb * 32 + 4 * h;  // + 8 g

    // weight stream of this wave: per sample conv1's steps 0 .. 35 (slab 4 kh + s / 9, tap s % 9), then conv2's; ring slot = tap.  The cursor runs
    // modulo the 72 steps and EVERY step refills its slot with the step 9 ahead, unconditionally: straight-line loads keep the counted vmcnt waits
    // (a conditional refill made every slab start with vmcnt(0), 7 us of exposed L2 latency per launch), conv1's last slab fetches conv2's first, and
    // conv2's last slab fetches conv1's first for the NEXT sample of this workgroup (weights do not depend on the sample).
    const char* wb1 = reinterpret_cast<const char*>(a.w1) + ((size_t)(cb * G::NS + kh * SPW) * TAPS) * WSTEP + (size_t)lane * 16;
    const char* wb2 = reinterpret_cast<const char*>(a.w2) + ((size_t)(cb * G::NS + kh * SPW) * TAPS) * WSTEP + (size_t)lane * 16;
    int pf = 0;  // prefetch cursor (wave-uniform)
    float4 wr[TAPS][NPL];
    auto ring_load = [&](int slot) {
        const char* p = (pf < NSTEP ? wb1 + (size_t)pf * WSTEP : wb2 + (size_t)(pf - NSTEP) * WSTEP);
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            if (ABL & 1) wr[slot][q] = make_float4(1e-3f * (float)q, 2e-3f, 3e-3f, 4e-3f);
            else wr[slot][q] = *reinterpret_cast<const float4*>(p + q * 1024);
        }
        pf = pf + 1 == 2 * NSTEP ? 0 : pf + 1;
    };
#pragma unroll
    for (int u = 0; u < TAPS; ++u) ring_load(u);

    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        // ================= (1) every independent load in one burst =================
        GnPartials gp;
        gn_load_partials(a.st, a.np, nullptr, 0, b, &gp);
        const float* xb = a.x + (size_t)b * NPIX * C;
        float4 sv[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int src, lds;
            s_geom(it, &src, &lds);
            sv[it] = *reinterpret_cast<const float4*>(xb + (size_t)src * C + c4 * 4);
        }
        const float4 g1 = *reinterpret_cast<const float4*>(a.g1 + c4 * 4), b1 = *reinterpret_cast<const float4*>(a.b1 + c4 * 4);
        float4 e_bt[2];  // bias + time bias of conv1 at this wave's epilogue couts
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
            const int co = e_co0 + 8 * (2 * kh + gg);
            const float4 bb = *reinterpret_cast<const float4*>(a.bias1 + co), tt = *reinterpret_cast<const float4*>(tbrow + (size_t)b * a.tbias_stride + co);
            e_bt[gg] = make_float4(bb.x + tt.x, bb.y + tt.y, bb.z + tt.z, bb.w + tt.w);
        }
        // ================= (2) GroupNorm 1 from the producer's partials; stage conv1's tile =================
        float mean, rstd;
        gn_reduce_partials(gp, a.st, a.np, nullptr, 0, b, (double)C * NPIX, &mean, &rstd);
        auto stage_write = [&](const float4& gq, const float4& bq, float mu, float rs) {
            float ga[4], gb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ga[i] = (&gq.x)[i] * rs;
                gb[i] = (&bq.x)[i] - mu * ga[i];
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                int src, lds;
                const bool ok = s_geom(it, &src, &lds);
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float x = dd_silu_scaled(fmaf((&sv[it].x)[i], ga[i], gb[i]), 1.0f / DDIF_F16_ASCALE);
                    v[i] = ok ? x : 0.f;  // zero padding comes AFTER the activation
                }
                if (lds >= 0) {
                    unsigned h01, l01, h23, l23;
                    dd_split2_pair(v[0], v[1], &h01, &l01);
                    dd_split2_pair(v[2], v[3], &h23, &l23);
                    float* d = &As[lds];
                    *reinterpret_cast<uint2*>(d) = make_uint2(h01, h23);
                    *reinterpret_cast<uint2*>(d + 8) = make_uint2(l01, l23);
                }
            }
        };
        stage_write(g1, b1, mean, rstd);
        __syncthreads();

        f32x16 acc[MB];
        // one conv's K half of this wave out of the staged tile: 4 slabs x 9 taps, ring refilled 9 steps ahead
        auto contract = [&]() {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
#pragma unroll 1
            for (int k = 0; k < SPW; ++k) {  // (not unrolled: one slab = 9 ring slots = the unit the code repeats; unrolled, hipcc hoists fragment reads across slabs and spills)
                const int sl = kh * SPW + k;
#pragma unroll
                for (int tap = 0; tap < TAPS; ++tap) {
                    const int aoff = (tap / 3) * AROW + (tap % 3) * APIX + sl * 16;
                    float4 xa[MB][NPL];
#pragma unroll
                    for (int q = 0; q < NPL; ++q)
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) xa[mb][q] = *reinterpret_cast<const float4*>(&As[abase[mb] + aoff + q * 8]);
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) {
                        f32x16 cacc = acc[mb];
                        if (ABL & 2) {
                            cacc[0] += wr[tap][0].x * xa[mb][0].x + wr[tap][1].y * xa[mb][1].y;
                            acc[mb] = cacc;
                            continue;
                        }
                        cacc = DDIF_MFMA_32x32x16_F16(wr[tap][1], xa[mb][0], cacc);  // lo * hi
                        cacc = DDIF_MFMA_32x32x16_F16(wr[tap][0], xa[mb][1], cacc);  // hi * lo
                        cacc = DDIF_MFMA_32x32x16_F16(wr[tap][0], xa[mb][0], cacc);  // hi * hi
                        acc[mb] = cacc;
                    }
                    ring_load(tap);  // the step 9 ahead, unconditionally (see the ring above)
#ifndef DDIF_EMU
                    asm volatile("" ::: "memory");  // pins the refill HERE: without it hipcc sinks the nine refills of a slab to the slab's end, and the next slab's first tap waits out a full L2 round trip
#endif
                    // (no sched_barrier around the MFMA group: kernels_lr.h records what that did on this part)
                }
            }
        };
        // K-partials -> LDS -> this wave's share of the sums: v[gg][mb] = (partial of kh 0) + (partial of kh 1), couts e_co0 + 8 (2 kh + gg) .. + 3 of pixel 32 mb + j
        auto reduce_k = [&](float4 (&v)[2][MB]) {
            __syncthreads();  // every wave is done reading the tile
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(&Red[(((wave * MB + mb) * 4 + g) * 64 + lane) * 4]) =
                        make_float4(acc[mb][4 * g + 0], acc[mb][4 * g + 1], acc[mb][4 * g + 2], acc[mb][4 * g + 3]);
            __syncthreads();
#pragma unroll
            for (int gg = 0; gg < 2; ++gg)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const int g = 2 * kh + gg;
                    const float4 p0v = *reinterpret_cast<const float4*>(&Red[((((cb * 2 + 0) * MB + mb) * 4 + g) * 64 + lane) * 4]);
                    const float4 p1v = *reinterpret_cast<const float4*>(&Red[((((cb * 2 + 1) * MB + mb) * 4 + g) * 64 + lane) * 4]);
                    v[gg][mb] = make_float4(p0v.x + p1v.x, p0v.y + p1v.y, p0v.z + p1v.z, p0v.w + p1v.w);
                }
        };

        // ================= (3) conv1 + bias + time bias -> h (LDS) and its statistics =================
        contract();
        float4 v[2][MB];
        reduce_k(v);
        // GroupNorm 2's affine (this thread's staging channels): from L2, in flight behind the epilogue below
        const float4 g2 = *reinterpret_cast<const float4*>(a.g2 + c4 * 4), b2 = *reinterpret_cast<const float4*>(a.b2 + c4 * 4);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int gg = 0; gg < 2; ++gg)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = fmaf((&v[gg][mb].x)[i], DDIF_F16_OSCALE, (&e_bt[gg].x)[i]);
                *reinterpret_cast<float4*>(&Hs[(mb * 32 + j) * HPIX + e_co0 + 8 * (2 * kh + gg)]) = make_float4(o[0], o[1], o[2], o[3]);
                s1 += (o[0] + o[1]) + (o[2] + o[3]);
                s2 += (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
            }
        {
            const float t1 = wave_sum_fast(s1), t2 = wave_sum_fast(s2);  // fp32 tree in-wave (total in lane 63), fp64 beyond
            if (lane == 63) {
                Sst[wave * 2 + 0] = t1;
                Sst[wave * 2 + 1] = t2;
            }
        }
        __syncthreads();  // h and the eight wave partials are complete; Red (= the tile region) is free again
        {
            const double q1 = (((double)Sst[0] + (double)Sst[2]) + ((double)Sst[4] + (double)Sst[6])) + (((double)Sst[8] + (double)Sst[10]) + ((double)Sst[12] + (double)Sst[14]));
            const double q2 = (((double)Sst[1] + (double)Sst[3]) + ((double)Sst[5] + (double)Sst[7])) + (((double)Sst[9] + (double)Sst[11]) + ((double)Sst[13] + (double)Sst[15]));
            const double mu = q1 / ((double)C * NPIX);
            double var = q2 / ((double)C * NPIX) - mu * mu;
            if (var < 0.0) var = 0.0;
            mean = (float)mu;
            rstd = (float)(1.0 / sqrt(var + DDIF_GN_EPS));
        }
        // ================= (4) GroupNorm 2 + SiLU -> conv2's tile =================
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int src, lds;
            s_geom(it, &src, &lds);
            sv[it] = *reinterpret_cast<const float4*>(&Hs[src * HPIX + c4 * 4]);
        }
        stage_write(g2, b2, mean, rstd);
        __syncthreads();
        // conv2's epilogue operands at this wave's positions: bias and the residual = x, re-read from L2, in flight behind the matrix phase
        float4 e_b2[2], e_res[2][MB];
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
            e_b2[gg] = *reinterpret_cast<const float4*>(a.bias2 + e_co0 + 8 * (2 * kh + gg));
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) e_res[gg][mb] = *reinterpret_cast<const float4*>(xb + (size_t)(mb * 32 + j) * C + e_co0 + 8 * (2 * kh + gg));
        }

        // ================= (5) conv2 + bias + residual -> out, one GroupNorm partial per sample =================
        contract();
        reduce_k(v);
        s1 = 0.f;
        s2 = 0.f;
        float* ob = a.out + (size_t)b * NPIX * C;
#pragma unroll
        for (int gg = 0; gg < 2; ++gg)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = fmaf((&v[gg][mb].x)[i], DDIF_F16_OSCALE, (&e_b2[gg].x)[i]) + (&e_res[gg][mb].x)[i];
                *reinterpret_cast<float4*>(ob + (size_t)(mb * 32 + j) * C + e_co0 + 8 * (2 * kh + gg)) = make_float4(o[0], o[1], o[2], o[3]);
                s1 += (o[0] + o[1]) + (o[2] + o[3]);
                s2 += (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
            }
        {
            const float t1 = wave_sum_fast(s1), t2 = wave_sum_fast(s2);
            if (lane == 63) {
                Sst[16 + wave * 2 + 0] = t1;
                Sst[16 + wave * 2 + 1] = t2;
            }
        }
        __syncthreads();  // also: Red is free for the next sample's tile
        if (a.st_out && tid == 0) {
            const float* S = Sst + 16;
            a.st_out[(size_t)b * 2 + 0] = (((double)S[0] + (double)S[2]) + ((double)S[4] + (double)S[6])) + (((double)S[8] + (double)S[10]) + ((double)S[12] + (double)S[14]));
            a.st_out[(size_t)b * 2 + 1] = (((double)S[1] + (double)S[3]) + ((double)S[5] + (double)S[7])) + (((double)S[9] + (double)S[11]) + ((double)S[13] + (double)S[15]));
        }
    }
}

}  // namespace ddif
