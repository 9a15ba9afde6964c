// Device-side primitives shared by every ddif kernel (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>

#ifdef DDIF_EMU  // tests/ only: host emulation build (tools/hipemu), never part of libddif.so
typedef hipemu_f32x16 f32x16;
typedef hipemu_f32x4 f32x4;
#define DDIF_MFMA_32x32x2(a, b, c) hipemu_mfma_32x32x2((a), (b), (c))
#define DDIF_MFMA_16x16x4(a, b, c) hipemu_mfma_16x16x4((a), (b), (c))
static inline f32x16 ddif_mfma_bf16_emu(float4 a, float4 b, f32x16 c) {
    hipemu_u32x4 ua, ub;
    __builtin_memcpy(&ua, &a, 16);
    __builtin_memcpy(&ub, &b, 16);
    return hipemu_mfma_32x32x16_bf16(ua, ub, c);
}
#define DDIF_MFMA_32x32x16_BF16(a, b, c) ddif_mfma_bf16_emu((a), (b), (c))
static inline f32x16 ddif_mfma_f16_emu(float4 a, float4 b, f32x16 c) {
    hipemu_u32x4 ua, ub;
    __builtin_memcpy(&ua, &a, 16);
    __builtin_memcpy(&ub, &b, 16);
    return hipemu_mfma_32x32x16_f16(ua, ub, c);
}
#define DDIF_MFMA_32x32x16_F16(a, b, c) ddif_mfma_f16_emu((a), (b), (c))
#define DDIF_DYN_SMEM(name) char* name = hipemu::tctx().dyn_smem
#define DDIF_SCHED_FENCE() ((void)0)
#else
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// exact-fp32 matrix FMA (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32): bitwise an fmaf chain over k
#define DDIF_MFMA_32x32x2(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define DDIF_MFMA_16x16x4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// v_mfma_f32_32x32x16_bf16 on operands carried as float4 (8 bf16 = 16 bytes per lane: A[i][k = 8*(lane>>5) + t])
typedef __bf16 ddif_bf16x8 __attribute__((ext_vector_type(8)));
#define DDIF_MFMA_32x32x16_BF16(a, b, c) \
    __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ddif_bf16x8, (a)), __builtin_bit_cast(ddif_bf16x8, (b)), (c), 0, 0, 0)
// v_mfma_f32_32x32x16_f16: the same operand layout with IEEE halves (the f16x2 split path below)
typedef _Float16 ddif_f16x8 __attribute__((ext_vector_type(8)));
#define DDIF_MFMA_32x32x16_F16(a, b, c) \
    __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ddif_f16x8, (a)), __builtin_bit_cast(ddif_f16x8, (b)), (c), 0, 0, 0)
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

// ---- 3-way bf16 split of an fp32 value: x = hi + mid + lo exactly to ~2^-24 |x| (each part 8 significant bits, round
// to nearest even; the remainders x - hi and x - hi - mid are exact in fp32).  Six cross products hi*hi, hi*mid, mid*hi,
// mid*mid, hi*lo, lo*hi on the bf16 matrix instruction (exact products, fp32 accumulate) reproduce the fp32 product to
// ~2^-23 relative (measured: tools/probes/bf16x3.cpp, 1.6e-7 vs 1.0e-7 of the exact-fp32 MFMA, relative to sum |a b|).
__device__ __forceinline__ unsigned dd_bf16_bits(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float dd_bf16_val(unsigned b) { return __builtin_bit_cast(float, b << 16); }
__device__ __forceinline__ void dd_split3(float x, unsigned* hi, unsigned* mid, unsigned* lo) {
    const unsigned h = dd_bf16_bits(x);
    const float r1 = x - dd_bf16_val(h);
    const unsigned m = dd_bf16_bits(r1);
    const float r2 = r1 - dd_bf16_val(m);
    *hi = h;
    *mid = m;
    *lo = dd_bf16_bits(r2);
}

// the same split for two values at once, results packed (a in the low half): v_cvt_pk_bf16_f32 does the RNE rounding
// of both in one instruction (5.5 VALU ops per value instead of ~20 with integer rounding)
__device__ __forceinline__ void dd_split3_pair(float a, float b, unsigned* hi, unsigned* mid, unsigned* lo) {
#ifdef DDIF_EMU
    unsigned ha, ma, la, hb, mb, lb;
    dd_split3(a, &ha, &ma, &la);
    dd_split3(b, &hb, &mb, &lb);
    *hi = ha | (hb << 16);
    *mid = ma | (mb << 16);
    *lo = la | (lb << 16);
#else
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef __bf16 v2b __attribute__((ext_vector_type(2)));
    const v2f v = {a, b};
    const unsigned H = __builtin_bit_cast(unsigned, __builtin_convertvector(v, v2b));
    const v2f r1 = {a - __builtin_bit_cast(float, H << 16), b - __builtin_bit_cast(float, H & 0xffff0000u)};
    const unsigned M = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, v2b));
    const v2f r2 = {r1.x - __builtin_bit_cast(float, M << 16), r1.y - __builtin_bit_cast(float, M & 0xffff0000u)};
    *hi = H;
    *mid = M;
    *lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, v2b));
#endif
}

// one RNE rounding to bf16 of two values, packed (a in the low half): the throughput variant's only operand conversion (kernels_conv.h MATH = 4)
__device__ __forceinline__ unsigned dd_bf16_pair(float a, float b) {
#ifdef DDIF_EMU
    return dd_bf16_bits(a) | (dd_bf16_bits(b) << 16);
#else
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef __bf16 v2b __attribute__((ext_vector_type(2)));
    const v2f v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, v2b));  // v_cvt_pk_bf16_f32
#endif
}

// ---- 2-way fp16 split ("f16x2"): x * S = hi + lo, hi = half(x S) (round to nearest even, 11 significant bits), lo =
// half(x S - hi) (the remainder is exact in fp32): 22 significant bits, |x S - hi - lo| <= 2^-22 |x S|.  Three cross products
// lo*hi, hi*lo, hi*hi on v_mfma_f32_32x32x16_f16 (each exact in fp32, fp32 accumulate; the dropped lo*lo term is 2^-22 relative)
// reproduce the fp32 product as well as the exact-fp32 MFMA does (tools/probes/f16x2.cpp on MI355X, K = 288, error / sum|a b|:
// 1.1e-7 max / 1.6e-8 rms against 1.0e-7 / 2.3e-8 for v_mfma_f32_32x32x2_f32 and 1.6e-7 / 2.0e-8 for bf16x3) at HALF the matrix
// instructions of bf16x3 and two operand planes instead of three.  What fp16 does not have is bf16's exponent range, so both
// operands are pre-scaled by fixed powers of two (exact) and the accumulator is scaled back in the epilogue's fma:
//   activations x 2^4: finite for |x| < 4094 (a GroupNorm(1 group) output is bounded by sqrt(N) max|gamma| + max|beta|; the host
//     checks that bound against 4094 per conv and keeps the conv on bf16x3 otherwise; a raw tensor beyond it gives inf -> NaN,
//     never a silently wrong finite value); the lo part is a NORMAL half for |x| >= 2^-7, below that its quantum is the
//     subnormal 2^-24 / 2^4, i.e. an absolute error <= 1.9e-9 per element (gfx950 keeps half subnormals in v_cvt_pk_f16_f32 and
//     in the matrix core: probe);
//   weights x 2^10: finite for |w| < 64 (checked by the host at commit), lo normal for |w| >= 2^-13, absolute error <= 3e-11 below.
#define DDIF_F16_ASCALE 16.0f
#define DDIF_F16_WSCALE 1024.0f
#define DDIF_F16_OSCALE (1.0f / (DDIF_F16_ASCALE * DDIF_F16_WSCALE))
#define DDIF_F16_AMAX 4094.0f
#define DDIF_F16_WMAX 63.9f
#ifdef DDIF_EMU
static inline unsigned dd_f16_bits(float x) { return hipemu_f32_to_f16(x); }
static inline float dd_f16_val(unsigned b) { return hipemu_f16_to_f32((unsigned short)b); }
#endif
// a, b are ALREADY scaled; results packed (a in the low half)
__device__ __forceinline__ void dd_split2_pair(float a, float b, unsigned* hi, unsigned* lo) {
#ifdef DDIF_EMU
    const unsigned ha = dd_f16_bits(a), hb = dd_f16_bits(b);
    *hi = ha | (hb << 16);
    *lo = dd_f16_bits(a - dd_f16_val(ha)) | (dd_f16_bits(b - dd_f16_val(hb)) << 16);
#else
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef _Float16 v2h __attribute__((ext_vector_type(2)));
    const v2f v = {a, b};
    const v2h H = __builtin_convertvector(v, v2h);  // v_cvt_pk_f16_f32 (RNE)
    const v2f r = {a - (float)H.x, b - (float)H.y};
    *hi = __builtin_bit_cast(unsigned, H);
    *lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, v2h));
#endif
}
// SiLU(x) * S with the scale folded into the sigmoid's denominator: x * rcp((1 + e) / S)
__device__ __forceinline__ float dd_silu_scaled(float x, float inv_s) {
#ifdef DDIF_EMU
    return x * (1.0f / fmaf(exp2f(-1.4426950408889634f * x), inv_s, inv_s));
#else
    return x * __builtin_amdgcn_rcpf(fmaf(__builtin_amdgcn_exp2f(-1.4426950408889634f * x), inv_s, inv_s));
#endif
}

// The sampler's device step counter (round 6): the load goes out where the value is read, but hipcc scheduled the dependent multiply -- and with it an s_waitcnt vmcnt(0) --
// right behind the load, at the top of the kernel: one more serial L2 round trip in front of everything.  dd_late() hands the value through an opaque move at the place of its
// first real use, so the wait lands there, behind the other loads of the burst.
__device__ __forceinline__ int dd_late(int v) {
#if !defined(DDIF_EMU) && defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#endif
    return v;
}

// Kernel-argument lines touched at kernel entry (round 5).  The argument block of the big kernels is 300-400 bytes = 5-7 scalar-cache lines, the L2s (and the scalar
// caches) start COLD at every kernel boundary on this part, and hipcc loads the fields of a by-value struct lazily, where they are first used: every first touch of another
// line deep inside the prologue was one more Infinity-Cache round trip in the dependent chain in front of the first MFMA.  One s_load per 64-byte line, waited for inside the
// same statement (the results are dummies: nothing may still be in flight when the compiler reuses their registers), puts all lines into the scalar cache in ONE round trip.
template <int BYTES>
__device__ __forceinline__ void dd_touch_kernargs() {
#if !defined(DDIF_EMU) && defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long ka = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();  // (constant address space: as a 64-bit SGPR pair)
    static_assert(BYTES >= 4 && BYTES <= 448, "kernarg touch: up to eight loads");
    // offsets 0, 64, 128, ... and the block's last dword (the segment need not start on a line boundary): NL loads in ONE statement
    constexpr int NL = (BYTES + 63) / 64 + 1;
#define DD_KA_OFF(i) ((i) * 64 < BYTES - 4 ? (i) * 64 : BYTES - 4)
    [[maybe_unused]] unsigned d0, d1, d2, d3, d4, d5, d6, d7;
    if constexpr (NL == 2) asm volatile("s_load_dword %0, %2, %3\n\ts_load_dword %1, %2, %4\n\ts_waitcnt lgkmcnt(0)" : "=&s"(d0), "=&s"(d1) : "s"(ka), "n"(DD_KA_OFF(0)), "n"(DD_KA_OFF(1)) : "memory");
    else if constexpr (NL == 3) asm volatile("s_load_dword %0, %3, %4\n\ts_load_dword %1, %3, %5\n\ts_load_dword %2, %3, %6\n\ts_waitcnt lgkmcnt(0)" : "=&s"(d0), "=&s"(d1), "=&s"(d2) : "s"(ka), "n"(DD_KA_OFF(0)), "n"(DD_KA_OFF(1)), "n"(DD_KA_OFF(2)) : "memory");
    else if constexpr (NL == 4) asm volatile("s_load_dword %0, %4, %5\n\ts_load_dword %1, %4, %6\n\ts_load_dword %2, %4, %7\n\ts_load_dword %3, %4, %8\n\ts_waitcnt lgkmcnt(0)" : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3) : "s"(ka), "n"(DD_KA_OFF(0)), "n"(DD_KA_OFF(1)), "n"(DD_KA_OFF(2)), "n"(DD_KA_OFF(3)) : "memory");
    else if constexpr (NL == 5) asm volatile("s_load_dword %0, %5, %6\n\ts_load_dword %1, %5, %7\n\ts_load_dword %2, %5, %8\n\ts_load_dword %3, %5, %9\n\ts_load_dword %4, %5, %10\n\ts_waitcnt lgkmcnt(0)" : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4) : "s"(ka), "n"(DD_KA_OFF(0)), "n"(DD_KA_OFF(1)), "n"(DD_KA_OFF(2)), "n"(DD_KA_OFF(3)), "n"(DD_KA_OFF(4)) : "memory");
    else if constexpr (NL == 6) asm volatile("s_load_dword %0, %6, %7\n\ts_load_dword %1, %6, %8\n\ts_load_dword %2, %6, %9\n\ts_load_dword %3, %6, %10\n\ts_load_dword %4, %6, %11\n\ts_load_dword %5, %6, %12\n\ts_waitcnt lgkmcnt(0)" : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4), "=&s"(d5) : "s"(ka), "n"(DD_KA_OFF(0)), "n"(DD_KA_OFF(1)), "n"(DD_KA_OFF(2)), "n"(DD_KA_OFF(3)), "n"(DD_KA_OFF(4)), "n"(DD_KA_OFF(5)) : "memory");
    else if constexpr (NL == 7) asm volatile("s_load_dword %0, %7, %8\n\ts_load_dword %1, %7, %9\n\ts_load_dword %2, %7, %10\n\ts_load_dword %3, %7, %11\n\ts_load_dword %4, %7, %12\n\ts_load_dword %5, %7, %13\n\ts_load_dword %6, %7, %14\n\ts_waitcnt lgkmcnt(0)" : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4), "=&s"(d5), "=&s"(d6) : "s"(ka), "n"(DD_KA_OFF(0)), "n"(DD_KA_OFF(1)), "n"(DD_KA_OFF(2)), "n"(DD_KA_OFF(3)), "n"(DD_KA_OFF(4)), "n"(DD_KA_OFF(5)), "n"(DD_KA_OFF(6)) : "memory");
    else if constexpr (NL == 8) asm volatile("s_load_dword %0, %8, %9\n\ts_load_dword %1, %8, %10\n\ts_load_dword %2, %8, %11\n\ts_load_dword %3, %8, %12\n\ts_load_dword %4, %8, %13\n\ts_load_dword %5, %8, %14\n\ts_load_dword %6, %8, %15\n\ts_load_dword %7, %8, %16\n\ts_waitcnt lgkmcnt(0)" : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4), "=&s"(d5), "=&s"(d6), "=&s"(d7) : "s"(ka), "n"(DD_KA_OFF(0)), "n"(DD_KA_OFF(1)), "n"(DD_KA_OFF(2)), "n"(DD_KA_OFF(3)), "n"(DD_KA_OFF(4)), "n"(DD_KA_OFF(5)), "n"(DD_KA_OFF(6)), "n"(DD_KA_OFF(7)) : "memory");
#undef DD_KA_OFF
#endif
}

// Contiguous share of a persistent workgroup: [floor(b n / g), floor((b + 1) n / g)) for b = blockIdx.x, g = gridDim.x -- the same partition as the
// 64-bit expression, from three 32-bit divisions (n = q g + r  =>  floor(b n / g) = b q + floor(b r / g), and b r < g^2 < 2^32): the two emulated
// 64-bit divisions were several hundred instructions at the head of every conv launch's dependent prologue chain.
// xcd (round 5): the dispatcher places workgroup b on XCD b % 8 (observed, MI355X_MICROARCH: never relied on for correctness).  With xcd != 0 and 8 | grid the
// workgroups of ONE XCD take a contiguous eighth of the work items, so that neighbouring tiles (shared halos), the cout tiles of one pixel tile and -- the
// producer having used the same map -- the consumer's input all meet in that XCD's private L2 instead of in HBM / Infinity Cache.
__device__ __forceinline__ void wg_work_range(int nwork, int* w0, int* w1, int xcd = 0) {
    const unsigned g = gridDim.x, n = (unsigned)nwork;
    const unsigned b = (xcd && (g & 7u) == 0u) ? (blockIdx.x & 7u) * (g >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const unsigned q = n / g, r = n - q * g;
    *w0 = (int)(b * q + (b * r) / g);
    *w1 = (int)((b + 1) * q + ((b + 1) * r) / g);
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
// fp64 sum over the 64 lanes with the same DPP sequence as wave_sum_fast (two 32-bit DPP moves per step instead of
// two ds_bpermute round trips): every lane returns the total (read from lane 63).
__device__ __forceinline__ double wave_sum_fast_f64(double v) {
#ifdef DDIF_EMU
    return wave_sum(v);
#else
#define DDIF_DPP_ADD_F64(ctrl, rmask)                                                                              \
    {                                                                                                              \
        const unsigned long long u = __builtin_bit_cast(unsigned long long, v);                                    \
        const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, ctrl, rmask, 0xf, false);                   \
        const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), ctrl, rmask, 0xf, false);           \
        v += __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo); \
    }
    DDIF_DPP_ADD_F64(0xB1, 0xf)   // quad_perm [1,0,3,2]
    DDIF_DPP_ADD_F64(0x4E, 0xf)   // quad_perm [2,3,0,1]
    DDIF_DPP_ADD_F64(0x124, 0xf)  // row_ror:4
    DDIF_DPP_ADD_F64(0x128, 0xf)  // row_ror:8
    DDIF_DPP_ADD_F64(0x142, 0xa)  // row_bcast:15
    DDIF_DPP_ADD_F64(0x143, 0xc)  // row_bcast:31
#undef DDIF_DPP_ADD_F64
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 63);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
#endif
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}

// GroupNorm(1 group) statistics travel with a tensor as per-producer-workgroup partials:
//   st[(b * np + i) * 2 + {0,1}] = {sum, sum of squares} over the elements workgroup i wrote for sample b.
// Called by ALL 64 lanes of a wavefront; returns mean / rstd for sample b over the concatenation of up to two
// tensors.  Deterministic (fixed summation order), fp64 combine.  Split in two so that a kernel prologue can put the
// loads (gn_load_partials: up to 4 pairs per lane and tensor in registers, no arithmetic on them) in flight together
// with its other prologue loads and pay ONE memory latency before gn_reduce_partials.
// one GroupNorm backward of a training iteration, for the batched dgamma / dbeta launch at the end of the reverse program (kernels_bwd.h gnb_bwd_reduce_all_kernel)
struct GnRedRec { const double* cpart; float* dgamma; float* dbeta; int C, nchunk, blk0, pad; };
struct GnPartials {
    double v[2][4][2];  // [tensor][k: partial lane + 64 k][sum | sum of squares]
};
__device__ __forceinline__ void gn_load_partials(const double* st0, int np0, const double* st1, int np1, int b, GnPartials* p) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = lane + 64 * k;
        const int i0 = i < np0 ? i : np0 - 1;
        p->v[0][k][0] = st0[((size_t)b * np0 + i0) * 2 + 0];
        p->v[0][k][1] = st0[((size_t)b * np0 + i0) * 2 + 1];
        if (st1 != nullptr) {
            const int i1 = i < np1 ? i : np1 - 1;
            p->v[1][k][0] = st1[((size_t)b * np1 + i1) * 2 + 0];
            p->v[1][k][1] = st1[((size_t)b * np1 + i1) * 2 + 1];
        }
    }
}
__device__ __forceinline__ void gn_reduce_partials(const GnPartials& p, const double* st0, int np0, const double* st1, int np1, int b,
                                                    double count, float* mean_out, float* rstd_out) {
    const int lane = threadIdx.x & 63;
    double s = 0.0, ss = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (lane + 64 * k < np0) {
            s += p.v[0][k][0];
            ss += p.v[0][k][1];
        }
    for (int i = lane + 256; i < np0; i += 64) {  // more than 256 producer workgroups per sample (images beyond 128x128)
        s += st0[((size_t)b * np0 + i) * 2 + 0];
        ss += st0[((size_t)b * np0 + i) * 2 + 1];
    }
    if (st1 != nullptr) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (lane + 64 * k < np1) {
                s += p.v[1][k][0];
                ss += p.v[1][k][1];
            }
        for (int i = lane + 256; i < np1; i += 64) {
            s += st1[((size_t)b * np1 + i) * 2 + 0];
            ss += st1[((size_t)b * np1 + i) * 2 + 1];
        }
    }
    s = wave_sum_fast_f64(s);
    ss = wave_sum_fast_f64(ss);
    const double mean = s / count;
    double var = ss / count - mean * mean;
    if (var < 0.0) var = 0.0;
    *mean_out = (float)mean;
    *rstd_out = (float)(1.0 / sqrt(var + DDIF_GN_EPS));
}
__device__ __forceinline__ void gn_finalize_wave(const double* st0, int np0, const double* st1, int np1, int b,
                                                  double count, float* mean_out, float* rstd_out) {
    GnPartials p;
    gn_load_partials(st0, np0, st1, np1, b, &p);
    gn_reduce_partials(p, st0, np0, st1, np1, b, count, mean_out, rstd_out);
}

}  // namespace ddif
