// Probe: can f32 VALU work hide under v_mfma_f32_32x32x2_f32 on gfx950?
//  (a) one wave per SIMD: N independent v_fma_f32 between consecutive dependent MFMAs -> cycles per MFMA vs N
//  (b) two waves per SIMD: wave A = MFMA stream, wave B = VALU stream; each one's cycle count alone vs together
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_coexec.cpp -o tools/probes/mfma_coexec.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int N, int CHAINS>
__global__ __launch_bounds__(256) void k_interleave(float* out, long long* cyc, int iters) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = a + i;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < N; ++n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[n % 12]) : "v"(b), "v"(a));
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
    for (int i = 0; i < 12; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// block of 512 threads = 8 waves = 2 per SIMD; waves 0-3 role A, 4-7 role B
template <int ROLE_A, int ROLE_B>  // 0 = idle, 1 = fp32 mfma stream, 2 = valu stream, 3 = transcendental stream, 4 = lds stream, 5 = bf16 mfma stream (32x32x16)
__global__ __launch_bounds__(512) void k_pair(float* out, long long* cyc, int iters) {
    __shared__ float lds[4096];
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? ROLE_A : ROLE_B;
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = a + i;
    lds[threadIdx.x] = a;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    if (role == 1) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    } else if (role == 5) {
        typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
        const float4 fa = make_float4(a, b, a, b);
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 8; ++u)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fa), acc, 0, 0, 0);
    } else if (role == 2) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 96; ++u) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[u % 12]) : "v"(b), "v"(a));
    } else if (role == 3) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 96; ++u) asm volatile("v_exp_f32 %0, %0" : "+v"(v[u % 12]));
    } else if (role == 4) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 24; ++u) {
                float q0, q1, q2, q3;
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(*reinterpret_cast<float4*>(&v[0])) : "v"((unsigned)((((threadIdx.x & 63) * 4 + u * 16) & 4092) * 4)));
                (void)q0; (void)q1; (void)q2; (void)q3;
            }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i];
    for (int i = 0; i < 12; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[1 + wave] = t1 - t0;
}

#define CK_(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int N, int CH> int run_i(float* out, long long* cyc, int iters) {
    hipLaunchKernelGGL((k_interleave<N, CH>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    CK_(hipDeviceSynchronize());
    long long h; CK_(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("one wave/SIMD, %d chain(s), %2d v_fma between MFMAs: %7.1f cycles per MFMA\n", CH, N, (double)h / (iters * 8.0 * CH));
    return 0;
}
template <int A, int B> int run_p(float* out, long long* cyc, int iters) {
    hipLaunchKernelGGL((k_pair<A, B>), dim3(256), dim3(512), 0, 0, out, cyc, iters);
    CK_(hipDeviceSynchronize());
    long long h[9]; CK_(hipMemcpy(h, cyc, 72, hipMemcpyDeviceToHost));
    const char* nm[] = {"idle", "mfma f32 x8/iter", "v_fma x96/iter", "v_exp x96/iter", "ds_read_b128 x24/iter", "mfma bf16 x8/iter"};
    printf("pair A=%-22s B=%-22s  A: %8.1f cycles/iter   B: %8.1f cycles/iter\n", nm[A], nm[B], (double)h[1] / iters, (double)h[5] / iters);
    return 0;
}
int main() {
    float* out; long long* cyc;
    CK_(hipMalloc(&out, 256 * 512 * 4)); CK_(hipMalloc(&cyc, 128)); CK_(hipMemset(cyc, 0, 128));
    const int iters = 2000;
    run_i<0, 1>(out, cyc, iters); run_i<4, 1>(out, cyc, iters); run_i<8, 1>(out, cyc, iters); run_i<12, 1>(out, cyc, iters); run_i<16, 1>(out, cyc, iters); run_i<24, 1>(out, cyc, iters);
    run_i<0, 2>(out, cyc, iters); run_i<8, 2>(out, cyc, iters); run_i<16, 2>(out, cyc, iters); run_i<32, 2>(out, cyc, iters);
    run_p<1, 0>(out, cyc, iters); run_p<0, 2>(out, cyc, iters); run_p<1, 2>(out, cyc, iters); run_p<1, 1>(out, cyc, iters);
    run_p<0, 3>(out, cyc, iters); run_p<1, 3>(out, cyc, iters); run_p<0, 4>(out, cyc, iters); run_p<1, 4>(out, cyc, iters); run_p<2, 2>(out, cyc, iters);
    // round 2: the bf16 matrix instruction of the bf16x3 path beside another wave's VALU / transcendental / LDS work
    run_p<5, 0>(out, cyc, iters); run_p<5, 2>(out, cyc, iters); run_p<5, 3>(out, cyc, iters); run_p<5, 4>(out, cyc, iters); run_p<5, 5>(out, cyc, iters);
    return 0;
}
