// Device-side primitives shared by every ddif kernel (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>

#ifdef DDIF_EMU  // tests/ only: host emulation build (tools/hipemu), never part of libddif.so
typedef hipemu_f32x16 f32x16;
typedef hipemu_f32x4 f32x4;
#define DDIF_MFMA_32x32x2(a, b, c) hipemu_mfma_32x32x2((a), (b), (c))
#define DDIF_MFMA_16x16x4(a, b, c) hipemu_mfma_16x16x4((a), (b), (c))
#define DDIF_DYN_SMEM(name) char* name = hipemu::tctx().dyn_smem
#define DDIF_SCHED_FENCE() ((void)0)
#else
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// exact-fp32 matrix FMA (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32): bitwise an fmaf chain over k
#define DDIF_MFMA_32x32x2(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define DDIF_MFMA_16x16x4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define DDIF_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
#define DDIF_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)  // instruction-scheduling fence (no code)
#endif

#define DDIF_WAVE 64
#define DDIF_GN_EPS 1e-5

#ifdef DDIF_EMU
static inline long long wall_clock64() { return 0; }
#endif

namespace ddif {

__device__ __forceinline__ float dd_exp(float x) { return expf(x); }  // accurate (softmax paths)
// SiLU on the hardware transcendental units: x * rcp(1 + exp2(-x*log2e)) = 2 quarter-rate + 3 full-rate VALU ops per
// element (v_exp_f32 / v_rcp_f32 are ~1 ulp; |error| of silu <~ 1e-7*|x|, the size of an fp32 rounding of the result).
__device__ __forceinline__ float dd_sigmoid(float x) {
#ifdef DDIF_EMU
    return 1.0f / (1.0f + exp2f(-1.4426950408889634f * x));
#else
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
#endif
}
__device__ __forceinline__ float dd_silu(float x) { return x * dd_sigmoid(x); }
__device__ __forceinline__ float dd_exp2_fast(float x) {
#ifdef DDIF_EMU
    return exp2f(x);
#else
    return __builtin_amdgcn_exp2f(x);
#endif
}
__device__ __forceinline__ float dd_rcp_fast(float x) {
#ifdef DDIF_EMU
    return 1.0f / x;
#else
    return __builtin_amdgcn_rcpf(x);
#endif
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
// Sum over the 64 lanes with DPP row operations (full-rate VALU, no LDS round trips); the total lands in LANE 63 only.
// Sequence as in rocPRIM's warp_reduce_dpp: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_ror:4, row_ror:8,
// row_bcast:15 (row_mask 0xa), row_bcast:31 (row_mask 0xc).
__device__ __forceinline__ float wave_sum_fast(float v) {
#ifdef DDIF_EMU
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
#else
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));  // row_bcast:15
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false));  // row_bcast:31
    return v;
#endif
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}

// GroupNorm(1 group) statistics travel with a tensor as per-producer-workgroup partials:
//   st[(b * np + i) * 2 + {0,1}] = {sum, sum of squares} over the elements workgroup i wrote for sample b.
// Called by ALL 64 lanes of a wavefront; returns mean / rstd for sample b over the concatenation of up to two
// tensors.  Deterministic (fixed summation order), fp64 combine.
__device__ __forceinline__ void gn_finalize_wave(const double* st0, int np0, const double* st1, int np1, int b,
                                                  double count, float* mean_out, float* rstd_out) {
    const int lane = threadIdx.x & 63;
    double s = 0.0, ss = 0.0;
    for (int i = lane; i < np0; i += 64) {
        s += st0[((size_t)b * np0 + i) * 2 + 0];
        ss += st0[((size_t)b * np0 + i) * 2 + 1];
    }
    if (st1 != nullptr) {
        for (int i = lane; i < np1; i += 64) {
            s += st1[((size_t)b * np1 + i) * 2 + 0];
            ss += st1[((size_t)b * np1 + i) * 2 + 1];
        }
    }
    s = wave_sum(s);
    ss = wave_sum(ss);
    const double mean = s / count;
    double var = ss / count - mean * mean;
    if (var < 0.0) var = 0.0;
    *mean_out = (float)mean;
    *rstd_out = (float)(1.0 / sqrt(var + DDIF_GN_EPS));
}

}  // namespace ddif
