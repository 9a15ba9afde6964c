// Probe for the 3-way bf16 split ("bf16x3") on gfx950: lane layout of v_mfma_f32_32x32x16_bf16, accuracy of the 6-term
// product against fp64 and against the exact-fp32 MFMA, dependent-chain issue rate.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/bf16x3.cpp -o tools/probes/bf16x3.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

__device__ __forceinline__ unsigned bf16_rne(float x) {  // bits of bf16(x), round to nearest even
    unsigned u = __builtin_bit_cast(unsigned, x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_f(unsigned b) { return __builtin_bit_cast(float, b << 16); }

// D[32 x 32] = A[32 x K] * B[K x 32], K multiple of 16; one wave.  A row-major [i][k], B row-major [k][j].
__global__ void k_x3(const float* A, const float* B, int K, float* D3, float* D1, float* Dx) {
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    f32x16 acc3, acc1, accx;
    for (int r = 0; r < 16; ++r) acc3[r] = acc1[r] = accx[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        bf16x8 a[3], b[3];
        for (int t = 0; t < 8; ++t) {
            const int k = k0 + 8 * h + t;
            float av = A[i * K + k], bv = B[k * 32 + i];
            unsigned a0 = bf16_rne(av); float ar = av - bf16_f(a0); unsigned a1 = bf16_rne(ar); ar -= bf16_f(a1); unsigned a2 = bf16_rne(ar);
            unsigned b0 = bf16_rne(bv); float br = bv - bf16_f(b0); unsigned b1 = bf16_rne(br); br -= bf16_f(b1); unsigned b2 = bf16_rne(br);
            a[0][t] = __builtin_bit_cast(__bf16, (u16)a0); a[1][t] = __builtin_bit_cast(__bf16, (u16)a1); a[2][t] = __builtin_bit_cast(__bf16, (u16)a2);
            b[0][t] = __builtin_bit_cast(__bf16, (u16)b0); b[1][t] = __builtin_bit_cast(__bf16, (u16)b1); b[2][t] = __builtin_bit_cast(__bf16, (u16)b2);
        }
        // small terms first
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc3, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc3, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc3, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc3, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc3, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc3, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc1, 0, 0, 0);
        for (int t = 0; t < 8; ++t) {  // exact f32 MFMA: k pairs (k, k + 1) per instruction, lane half picks which
            // 32x32x2: A[i][k = h], B[k = h][j]
        }
    }
    for (int k0 = 0; k0 < K; k0 += 2) accx = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k0 + h], B[(k0 + h) * 32 + i], accx, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        D3[row * 32 + i] = acc3[r]; D1[row * 32 + i] = acc1[r]; Dx[row * 32 + i] = accx[r];
    }
}

template <int CH>
__global__ void k_rate(float* out, long long* cyc, int iters) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    bf16x8 a, b;
    for (int t = 0; t < 8; ++t) { a[t] = (__bf16)(threadIdx.x * 0.001f + t); b[t] = (__bf16)(1.0f + t * 0.01f); }
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

#define CK_(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
    const int K = 288;
    std::vector<float> A(32 * K), B(K * 32);
    srand(1);
    for (auto& v : A) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
    for (auto& v : B) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
    float *dA, *dB, *d3, *d1, *dx; long long* cyc;
    CK_(hipMalloc(&dA, A.size() * 4)); CK_(hipMalloc(&dB, B.size() * 4)); CK_(hipMalloc(&d3, 4096)); CK_(hipMalloc(&d1, 4096)); CK_(hipMalloc(&dx, 4096)); CK_(hipMalloc(&cyc, 64));
    CK_(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK_(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_x3, dim3(1), dim3(64), 0, 0, dA, dB, K, d3, d1, dx);
    CK_(hipDeviceSynchronize());
    std::vector<float> h3(1024), h1(1024), hx(1024);
    CK_(hipMemcpy(h3.data(), d3, 4096, hipMemcpyDeviceToHost)); CK_(hipMemcpy(h1.data(), d1, 4096, hipMemcpyDeviceToHost)); CK_(hipMemcpy(hx.data(), dx, 4096, hipMemcpyDeviceToHost));
    double e3 = 0, e1 = 0, ex = 0, mag = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double r = 0, m = 0;
        for (int k = 0; k < K; ++k) { r += (double)A[i * K + k] * B[k * 32 + j]; m += fabs((double)A[i * K + k] * B[k * 32 + j]); }
        e3 = fmax(e3, fabs(h3[i * 32 + j] - r) / m); e1 = fmax(e1, fabs(h1[i * 32 + j] - r) / m); ex = fmax(ex, fabs(hx[i * 32 + j] - r) / m);
        mag = fmax(mag, fabs(r));
    }
    printf("K=%d  max |err| / sum|a*b| vs fp64:   bf16x3 (6 terms) %.3e    plain bf16 %.3e    exact-f32 MFMA %.3e   (max |d| %.3f)\n", K, e3, e1, ex, mag);
    float* out; CK_(hipMalloc(&out, 256 * 256 * 4));
    const int iters = 2000;
    hipLaunchKernelGGL((k_rate<1>), dim3(256), dim3(256), 0, 0, out, cyc, iters); CK_(hipDeviceSynchronize());
    long long c1; CK_(hipMemcpy(&c1, cyc, 8, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL((k_rate<2>), dim3(256), dim3(256), 0, 0, out, cyc, iters); CK_(hipDeviceSynchronize());
    long long c2; CK_(hipMemcpy(&c2, cyc, 8, hipMemcpyDeviceToHost));
    printf("v_mfma_f32_32x32x16_bf16: %.1f cycles per MFMA (1 dependent chain), %.1f (2 chains), one wave per SIMD\n", c1 / (iters * 8.0), c2 / (iters * 16.0));
    return 0;
}
