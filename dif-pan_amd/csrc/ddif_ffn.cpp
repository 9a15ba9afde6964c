// Instantiation + launcher of the fused feed-forward kernel (kernels_ffn.h) -- a translation unit of its own so that it builds in
// parallel with the conv kernel families.
#include "ddif_plan.h"
#include "kernels_ffn.h"

namespace ddif {

// shapes the fused kernel carries: C = 32 -> 64 -> 32 (the 64-pixel-tile level of the engine network), any H x W (partial tiles are masked)
bool ffnfuse_supported(int C, int CM) { return C == 32 && CM == 64; }
size_t ffnfuse_smem() { return FfnFuseGeom<32, 64>::smem; }
int ffnfuse_launch(const FfnFuseArgs& a, int grid, hipStream_t s, bool prepare_only) {
    auto fn = ffn_fused_kernel<32, 64>;
    if (prepare_only) {
        DDIF_HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FfnFuseGeom<32, 64>::smem));
        return 0;
    }
    const size_t smem = FfnFuseGeom<32, 64>::smem;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(512), smem, s, a);
    return 0;
}

}  // namespace ddif
