// Sampler state shared by the update kernels (kernels_misc.h) and the final conv's sampler epilogue (kernels_conv.h EPI_SAMP):
// the device-resident description of a run and the counter-based normal generator.
#pragma once
#include "ddif_dev.h"

namespace ddif {

struct SamplerRun {
    const float* noise;  // (n_steps, B, C, H, W) NCHW standard normals in execution order, or null -> Philox
    unsigned long long seed, tile0;
    float lo, hi;
    int do_clamp, n_steps;
    const float* tab[6];  // per-step coefficient tables (device), meaning depends on the sampler
};

// ----------------------------------------------------------------------------------------------------------------
// counter-based normal generator (Philox4x32-10 + Box-Muller), keyed by (seed, draw index, global element index) so
// results do not depend on the batch split across GPUs.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned* o) {
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
__device__ __forceinline__ float philox_normal(unsigned long long seed, unsigned draw, unsigned long long elem) {
    unsigned o[4];
    philox4x32_10((unsigned)elem, (unsigned)(elem >> 32), draw, 0x5DD1Fu, (unsigned)seed, (unsigned)(seed >> 32), o);
    const float u1 = ((float)(o[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(o[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    // Box-Muller on the hardware transcendentals: v_log_f32 (log2), v_sqrt_f32, v_cos_f32 (argument in revolutions: cos(2 pi u2) directly).
    // ~1e-6 absolute on a standard normal -- and no libm call: ocml's cosf keeps a private array for its large-argument reduction, which put
    // 2 KB of scratch per lane into every kernel that inlines this (the final conv with the sampler epilogue ran 533 us instead of 30)
#ifdef DDIF_EMU
    return sqrtf(-2.0f * 0.69314718055994530942f * log2f(u1)) * cosf(6.28318530717958647692f * u2);
#else
    return __builtin_amdgcn_sqrtf(-2.0f * 0.69314718055994530942f * __builtin_amdgcn_logf(u1)) * __builtin_amdgcn_cosf(u2);
#endif
}


}  // namespace ddif
