// Probe for the 2-way fp16 split ("f16x2") on gfx950: accuracy of the 3-term product hi*hi + hi*lo + lo*hi on
// v_mfma_f32_32x32x16_f16 against fp64, next to the 6-term bf16x3 product and the exact-fp32 MFMA; what the matrix core and
// v_cvt_pk_f16_f32 do with fp16 SUBNORMALS (the lo parts of small values are subnormal); effect of power-of-two pre-scaling.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/f16x2.cpp -o tools/probes/f16x2.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;

__device__ __forceinline__ unsigned bf16_rne(float x) {
    unsigned u = __builtin_bit_cast(unsigned, x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_f(unsigned b) { return __builtin_bit_cast(float, b << 16); }

// the split the kernels would use: two values at once through the packed conversion
__device__ __forceinline__ void split2_pair(float a, float b, f16x2* hi, f16x2* lo) {
    const f32x2 v = {a, b};
    const f16x2 H = __builtin_convertvector(v, f16x2);
    const f32x2 r = {a - (float)H.x, b - (float)H.y};
    *hi = H;
    *lo = __builtin_convertvector(r, f16x2);
}

// D[32 x 32] = A[32 x K] * B[K x 32]; one wave.  sa / sb: power-of-two pre-scales of A / B (undone on the result).
__global__ void k_acc(const float* A, const float* B, int K, float sa, float sb, float* D2, float* D3, float* Dx) {
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    f32x16 acc2, acc3, accx;
    for (int r = 0; r < 16; ++r) acc2[r] = acc3[r] = accx[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        bf16x8 a[3], b[3];
        f16x8 ah, al, bh, bl;
        for (int t = 0; t < 8; t += 2) {
            const int k = k0 + 8 * h + t;
            f16x2 h2, l2;
            split2_pair(A[i * K + k] * sa, A[i * K + k + 1] * sa, &h2, &l2);
            ah[t] = h2.x; ah[t + 1] = h2.y; al[t] = l2.x; al[t + 1] = l2.y;
            split2_pair(B[k * 32 + i] * sb, B[(k + 1) * 32 + i] * sb, &h2, &l2);
            bh[t] = h2.x; bh[t + 1] = h2.y; bl[t] = l2.x; bl[t + 1] = l2.y;
        }
        for (int t = 0; t < 8; ++t) {
            const int k = k0 + 8 * h + t;
            float av = A[i * K + k], bv = B[k * 32 + i];
            unsigned a0 = bf16_rne(av); float ar = av - bf16_f(a0); unsigned a1 = bf16_rne(ar); ar -= bf16_f(a1); unsigned a2 = bf16_rne(ar);
            unsigned b0 = bf16_rne(bv); float br = bv - bf16_f(b0); unsigned b1 = bf16_rne(br); br -= bf16_f(b1); unsigned b2 = bf16_rne(br);
            a[0][t] = __builtin_bit_cast(__bf16, (u16)a0); a[1][t] = __builtin_bit_cast(__bf16, (u16)a1); a[2][t] = __builtin_bit_cast(__bf16, (u16)a2);
            b[0][t] = __builtin_bit_cast(__bf16, (u16)b0); b[1][t] = __builtin_bit_cast(__bf16, (u16)b1); b[2][t] = __builtin_bit_cast(__bf16, (u16)b2);
        }
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc3, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc3, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc3, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc3, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc3, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc3, 0, 0, 0);
    }
    for (int k0 = 0; k0 < K; k0 += 2) accx = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k0 + h], B[(k0 + h) * 32 + i], accx, 0, 0, 0);
    const float inv = 1.0f / (sa * sb);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        D2[row * 32 + i] = acc2[r] * inv; D3[row * 32 + i] = acc3[r]; Dx[row * 32 + i] = accx[r];
    }
}

// subnormal behaviour: out[0] = bits of f16(x) for a value in the fp16 subnormal range (conversion flushes?),
// out[1..] = D[0][0] of an MFMA whose A operand is the subnormal 2^-20 and B = 2^10 (expected 16 * 2^-10 if not flushed)
__global__ void k_sub(float* out) {
    const float tiny = 9.5367431640625e-07f;  // 2^-20: subnormal in fp16 (min normal 2^-14)
    const f32x2 v = {tiny, 3.0f * tiny};
    const f16x2 c = __builtin_convertvector(v, f16x2);
    f16x8 a, b;
    for (int t = 0; t < 8; ++t) { a[t] = c.x; b[t] = (_Float16)1024.f; }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) {
        out[0] = (float)c.x;
        out[1] = (float)c.y;
        out[2] = acc[0];
        out[3] = 16.f * tiny * 1024.f;
    }
}

template <int CH>
__global__ void k_rate(float* out, long long* cyc, int iters) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    f16x8 a, b;
    for (int t = 0; t < 8; ++t) { a[t] = (_Float16)(threadIdx.x * 0.001f + t); b[t] = (_Float16)(1.0f + t * 0.01f); }
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

#define CK_(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
static float urand() { return rand() / (float)RAND_MAX - 0.5f; }
static float nrand() { float s = 0; for (int i = 0; i < 12; ++i) s += rand() / (float)RAND_MAX; return s - 6.f; }

int main() {
    const int K = 288;
    float *dA, *dB, *d2, *d3, *dx; long long* cyc;
    CK_(hipMalloc(&dA, 32 * K * 4)); CK_(hipMalloc(&dB, K * 32 * 4)); CK_(hipMalloc(&d2, 4096)); CK_(hipMalloc(&d3, 4096)); CK_(hipMalloc(&dx, 4096)); CK_(hipMalloc(&cyc, 64));
    struct Case { const char* name; float amag, bmag; int gauss; float sa, sb; };
    const Case cases[] = {
        {"A~U(-2,2)      B~U(-.1,.1)    unscaled      ", 4.f, 0.2f, 0, 1.f, 1.f},
        {"A~U(-2,2)      B~U(-.1,.1)    sa=16  sb=2^14", 4.f, 0.2f, 0, 16.f, 16384.f},
        {"A~N(0,1)       B~N(0,.05)     unscaled      ", 1.f, 0.05f, 1, 1.f, 1.f},
        {"A~N(0,1)       B~N(0,.05)     sa=16  sb=2^12", 1.f, 0.05f, 1, 16.f, 4096.f},
        {"A~N(0,1e-3)    B~N(0,.05)     unscaled      ", 1e-3f, 0.05f, 1, 1.f, 1.f},
        {"A~N(0,1e-3)    B~N(0,.05)     sa=16  sb=2^12", 1e-3f, 0.05f, 1, 16.f, 4096.f},
        {"A~N(0,30)      B~N(0,.05)     sa=1   sb=2^12", 30.f, 0.05f, 1, 1.f, 4096.f},
    };
    for (const Case& c : cases) {
        std::vector<float> A(32 * K), B(K * 32);
        srand(1);
        for (auto& v : A) v = (c.gauss ? nrand() : urand()) * c.amag;
        for (auto& v : B) v = (c.gauss ? nrand() : urand()) * c.bmag;
        CK_(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK_(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_acc, dim3(1), dim3(64), 0, 0, dA, dB, K, c.sa, c.sb, d2, d3, dx);
        CK_(hipDeviceSynchronize());
        std::vector<float> h2(1024), h3(1024), hx(1024);
        CK_(hipMemcpy(h2.data(), d2, 4096, hipMemcpyDeviceToHost)); CK_(hipMemcpy(h3.data(), d3, 4096, hipMemcpyDeviceToHost)); CK_(hipMemcpy(hx.data(), dx, 4096, hipMemcpyDeviceToHost));
        double e2 = 0, e3 = 0, ex = 0, r2 = 0, r3 = 0, rx = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double r = 0, m = 0;
            for (int k = 0; k < K; ++k) { r += (double)A[i * K + k] * B[k * 32 + j]; m += fabs((double)A[i * K + k] * B[k * 32 + j]); }
            const double d2e = fabs(h2[i * 32 + j] - r) / m, d3e = fabs(h3[i * 32 + j] - r) / m, dxe = fabs(hx[i * 32 + j] - r) / m;
            e2 = fmax(e2, d2e); e3 = fmax(e3, d3e); ex = fmax(ex, dxe);
            r2 += d2e * d2e; r3 += d3e * d3e; rx += dxe * dxe;
        }
        printf("%s  max (rms) |err|/sum|ab|:  f16x2 %.2e (%.2e)   bf16x3 %.2e (%.2e)   exact-f32 %.2e (%.2e)\n", c.name, e2, sqrt(r2 / 1024), e3, sqrt(r3 / 1024), ex, sqrt(rx / 1024));
    }
    float* out; CK_(hipMalloc(&out, 256 * 256 * 4));
    hipLaunchKernelGGL(k_sub, dim3(1), dim3(64), 0, 0, out); CK_(hipDeviceSynchronize());
    float hs[4]; CK_(hipMemcpy(hs, out, 16, hipMemcpyDeviceToHost));
    printf("subnormals: f16(2^-20) = %.6e (%s), f16(3*2^-20) = %.6e; MFMA sum_k 2^-20 * 2^10 = %.6e, expected %.6e (%s)\n", hs[0], hs[0] != 0.f ? "kept" : "FLUSHED", hs[1], hs[2], hs[3],
           hs[2] == hs[3] ? "kept" : "FLUSHED");
    const int iters = 2000;
    hipLaunchKernelGGL((k_rate<1>), dim3(256), dim3(256), 0, 0, out, cyc, iters); CK_(hipDeviceSynchronize());
    long long c1; CK_(hipMemcpy(&c1, cyc, 8, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL((k_rate<2>), dim3(256), dim3(256), 0, 0, out, cyc, iters); CK_(hipDeviceSynchronize());
    long long c2; CK_(hipMemcpy(&c2, cyc, 8, hipMemcpyDeviceToHost));
    printf("v_mfma_f32_32x32x16_f16: %.1f cycles per MFMA (1 dependent chain), %.1f (2 chains), one wave per SIMD\n", c1 / (iters * 8.0), c2 / (iters * 16.0));
    return 0;
}
