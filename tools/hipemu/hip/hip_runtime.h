// hipemu: a TEST-ONLY host emulation of the subset of the HIP programming model the ddif kernels use.
//
// The product is built by hipcc for gfx950 and never sees this file.  tests/ build the same csrc/ sources
// with the host clang++ and `-I tools/hipemu -DDDIF_EMU` so that kernel logic (indexing, LDS tiling, barrier
// placement, MFMA lane maps, host launch sequences) can be checked against the oracle in a container without
// a GPU, and under UBSan.  Threads of a workgroup run as cooperatively scheduled fibers; a wavefront is 64
// consecutive fibers; __syncthreads / shuffles / MFMA are rendezvous points.  Blocks are spread over OS
// threads.  It models semantics, not timing.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#ifndef __restrict__
#define __restrict__ __restrict
#endif

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(16) double2 { double x, y; };
struct uint2 { unsigned x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline double2 make_double2(double x, double y) { return {x, y}; }
static inline int2 make_int2(int x, int y) { return {x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }

namespace hipemu {
struct ThreadCtx {
    dim3 tid, bid, bdim, gdim;
    char* dyn_smem;
    int lane, wave, linear;
};
ThreadCtx& tctx();
void block_barrier();
// wave-wide exchange: every lane deposits `bytes` at src, gets a pointer to the 64 x bytes staging area
// (lane-major) valid until wave_release().
const char* wave_gather(const void* src, size_t bytes);
void wave_release();
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
}  // namespace hipemu

#define threadIdx (hipemu::tctx().tid)
#define blockIdx (hipemu::tctx().bid)
#define blockDim (hipemu::tctx().bdim)
#define gridDim (hipemu::tctx().gdim)
#define warpSize 64

static inline void __syncthreads() { hipemu::block_barrier(); }

template <typename T>
static inline T __hipemu_shfl_idx(T v, int src_lane) {
    const char* buf = hipemu::wave_gather(&v, sizeof(T));
    T out;
    std::memcpy(&out, buf + (size_t)(src_lane & 63) * sizeof(T), sizeof(T));
    hipemu::wave_release();
    return out;
}
template <typename T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
    (void)width;
    return __hipemu_shfl_idx(v, hipemu::tctx().lane ^ mask);
}
template <typename T>
static inline T __shfl_down(T v, unsigned delta, int width = 64) {
    int l = hipemu::tctx().lane;
    int src = l + (int)delta;
    if ((src / width) != (l / width) || src > 63) src = l;
    return __hipemu_shfl_idx(v, src);
}
template <typename T>
static inline T __shfl(T v, int src, int width = 64) {
    int l = hipemu::tctx().lane;
    return __hipemu_shfl_idx(v, (l / width) * width + (src % width));
}

// ---- MFMA f32 (lane maps per cdna_hip_programming.md section 3) -------------------------------------------
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));

// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; D col=l&31, row=(r&3)+8*(r>>2)+4*(l>>5)
static inline hipemu_f32x16 hipemu_mfma_32x32x2(float a, float b, hipemu_f32x16 c) {
    float ab[2] = {a, b};
    const float* g = (const float*)hipemu::wave_gather(ab, sizeof(ab));
    int l = hipemu::tctx().lane;
    int col = l & 31, hi = l >> 5;
    hipemu_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = d[r];
        for (int k = 0; k < 2; ++k) acc = fmaf(g[(row + 32 * k) * 2 + 0], g[(col + 32 * k) * 2 + 1], acc);
        d[r] = acc;
    }
    hipemu::wave_release();
    return d;
}
// v_mfma_f32_32x32x16_bf16: A[i=l&31][k=8*(l>>5)+t], B[k=8*(l>>5)+t][j=l&31], t = 0..7 (8 bf16 = 4 dwords per lane);
// same D layout as 32x32x2.  Products are exact in fp32; the accumulation order of the hardware is not documented, so
// this model (k ascending, fp32 adds) agrees with the GPU to fp32 rounding, not bitwise.
struct hipemu_u32x4 { unsigned v[4]; };
static inline float hipemu_bf16_to_f32(unsigned short b) { unsigned u = (unsigned)b << 16; float f; __builtin_memcpy(&f, &u, 4); return f; }
static inline hipemu_f32x16 hipemu_mfma_32x32x16_bf16(hipemu_u32x4 a, hipemu_u32x4 b, hipemu_f32x16 c) {
    float ab[16];  // this lane's 8 A values then its 8 B values, already widened (exactly) to fp32
    for (int t = 0; t < 8; ++t) {
        const unsigned aw = a.v[t >> 1], bw = b.v[t >> 1];
        ab[t] = hipemu_bf16_to_f32((unsigned short)((t & 1) ? (aw >> 16) : (aw & 0xffffu)));
        ab[8 + t] = hipemu_bf16_to_f32((unsigned short)((t & 1) ? (bw >> 16) : (bw & 0xffffu)));
    }
    const float* g = (const float*)hipemu::wave_gather(ab, sizeof(ab));
    int l = hipemu::tctx().lane;
    int col = l & 31, hi = l >> 5;
    hipemu_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = d[r];
        for (int half = 0; half < 2; ++half) {
            const float* ar = g + (row + 32 * half) * 16;
            const float* bc = g + (col + 32 * half) * 16 + 8;
            for (int t = 0; t < 8; ++t) acc += ar[t] * bc[t];  // exact product (8 x 8 significant bits), one fp32 rounding per add
        }
        d[r] = acc;
    }
    hipemu::wave_release();
    return d;
}
// v_mfma_f32_32x32x16_f16: the same lane layout with IEEE half operands (subnormals kept, as measured on gfx950:
// tools/probes/f16x2.cpp).  Products of two halves (11 x 11 significant bits) are exact in fp32.
static inline float hipemu_f16_to_f32(unsigned short b) { _Float16 hv; __builtin_memcpy(&hv, &b, 2); return (float)hv; }
static inline unsigned short hipemu_f32_to_f16(float f) { _Float16 hv = (_Float16)f; unsigned short b; __builtin_memcpy(&b, &hv, 2); return b; }  // round to nearest even
static inline hipemu_f32x16 hipemu_mfma_32x32x16_f16(hipemu_u32x4 a, hipemu_u32x4 b, hipemu_f32x16 c) {
    float ab[16];
    for (int t = 0; t < 8; ++t) {
        const unsigned aw = a.v[t >> 1], bw = b.v[t >> 1];
        ab[t] = hipemu_f16_to_f32((unsigned short)((t & 1) ? (aw >> 16) : (aw & 0xffffu)));
        ab[8 + t] = hipemu_f16_to_f32((unsigned short)((t & 1) ? (bw >> 16) : (bw & 0xffffu)));
    }
    const float* g = (const float*)hipemu::wave_gather(ab, sizeof(ab));
    int l = hipemu::tctx().lane;
    int col = l & 31, hi = l >> 5;
    hipemu_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = d[r];
        for (int half = 0; half < 2; ++half) {
            const float* ar = g + (row + 32 * half) * 16;
            const float* bc = g + (col + 32 * half) * 16 + 8;
            for (int t = 0; t < 8; ++t) acc += ar[t] * bc[t];
        }
        d[r] = acc;
    }
    hipemu::wave_release();
    return d;
}
// v_mfma_f32_16x16x4_f32: A[l&15][k=l>>4], B[k=l>>4][l&15]; D col=l&15, row=(l>>4)*4+r
static inline hipemu_f32x4 hipemu_mfma_16x16x4(float a, float b, hipemu_f32x4 c) {
    float ab[2] = {a, b};
    const float* g = (const float*)hipemu::wave_gather(ab, sizeof(ab));
    int l = hipemu::tctx().lane;
    int col = l & 15, q = l >> 4;
    hipemu_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        int row = q * 4 + r;
        float acc = d[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(g[(row + 16 * k) * 2 + 0], g[(col + 16 * k) * 2 + 1], acc);
        d[r] = acc;
    }
    hipemu::wave_release();
    return d;
}

// ---- math the kernels use ------------------------------------------------------------------------------
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline float __fdividef(float a, float b) { return a / b; }
static inline void sincosf_(float x, float* s, float* c) { *s = sinf(x); *c = cosf(x); }

// ---- runtime API subset -------------------------------------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
typedef struct hipemu_stream* hipStream_t;
typedef struct hipemu_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
const char* hipGetErrorString(hipError_t);
hipError_t hipGetLastError();
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipDeviceSynchronize();
hipError_t hipStreamSynchronize(hipStream_t);
hipError_t hipSetDevice(int);
hipError_t hipGetDevice(int*);
hipError_t hipGetDeviceCount(int*);
hipError_t hipEventCreate(hipEvent_t*);
hipError_t hipEventDestroy(hipEvent_t);
hipError_t hipEventRecord(hipEvent_t, hipStream_t);
hipError_t hipEventSynchronize(hipEvent_t);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
template <typename T>
static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
static inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 4; return hipSuccess; }

namespace hipemu {
template <typename K, typename... A>
static inline void launch_kernel(K kern, dim3 grid, dim3 block, size_t shmem, A... args) {
    launch(grid, block, shmem, [=]() { kern(args...); });
}
}  // namespace hipemu
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    hipemu::launch_kernel(kern, (grid), (block), (shmem), __VA_ARGS__)
