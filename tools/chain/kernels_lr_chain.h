// MEASURED AND NOT ADOPTED (round 5; profiles/r05/r_lr_chain.txt): consecutive low-resolution conv launches as ONE kernel behind XCD-local barriers.
// Development code for tools/mbench_chain.cpp -- nothing in the product includes this file.
// The layer body is the product's conv_lr_kernel with its __global__ head replaced (tools/chain/make_body.sh generates tools/chain/_gen/kernels_lr_body.h from
// dif-pan_amd/csrc/kernels_lr.h: the product's kernel text stays the single source).
#pragma once
#include <cstddef>
#include "kernels_lr.h"
#include "_gen/kernels_lr_body.h"

namespace ddif {

// XCD-local barrier (round 5).  All participants sit behind ONE L2 (the workgroups of an XCD: blockIdx % 8, checked by the
// caller against XCC_ID), so nothing has to leave it: stores drained (vmcnt 0 = acknowledged by the L2), one returning atomic add executed IN that L2
// (workgroup scope: no sc1 -- an agent-scope atomic is resolved at the fabric, > 1 us), the generation derived from the value it returns (the counter is
// monotonic: gridDim / 8 arrivals per barrier), polling loads with sc1 (an sc0 load hits the vector L1 for ever: tools/probes/xcdsync.cpp).  The vector L1 is
// not invalidated -- see below for the contract that makes that correct.  1.0 us against 2.8 us for a kernel boundary; bounded spin: *fault = 1.
__device__ __forceinline__ unsigned dd_xcc_id() {
#ifdef DDIF_EMU
    return blockIdx.x & 7u;
#else
    return __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | ((4 - 1) << 11));
#endif
}
__device__ __forceinline__ void xcd_barrier(unsigned* ctr, int* fault) {
#ifndef DDIF_EMU
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned per = gridDim.x >> 3;
        const unsigned old = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const unsigned target = (old / per + 1u) * per;
        int spins = 0;
        while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            if (++spins > (1 << 20)) {
                *fault = 1;
                break;
            }
        }
    }
    __syncthreads();
#endif
}

// ------------------------------------------------------------------------------------------------------------------------------------------------------------
// A CHAIN of low-resolution launches in one kernel (round 5).  Every launch starts cold on this part (the L2s are written back and invalidated at each kernel
// boundary) and a boundary costs 2.8 us even between empty kernels; with the XCD-contiguous work partition (ddif_dev.h wg_work_range, xcd = 1) and 8 | B the
// workgroups of ONE XCD produce everything the same workgroups consume in the next layer -- samples never cross XCDs -- so consecutive layers need a barrier
// among the 32 workgroups of an XCD only, and their hand-off never leaves that XCD's L2:  xcd_barrier (below), 1.0 us (tools/probes/xcdsync.cpp,
// profiles/r05/r_xcdsync_probe.txt), the consumer's input is L2-hot and its weight ring / argument lines can be in flight under the wait.
//   * grid = 256 workgroups of 256 threads, all co-resident (one per CU); every spin is bounded and raises the plan's fault flag instead of hanging;
//   * the vector L1 is NOT invalidated (buffer_inv sc1, the only form that does it, costs 7.5 us: measured): the host only chains layers whose inputs are fresh in
//     the kernel -- nothing a layer writes may have been read or written by an earlier layer of the same chain (a plan would have to check that) -- so no L1 can
//     hold a stale line of it;
//   * arithmetic, tiling and summation order are those of the separate launches: results are bit-identical (tools/mbench_chain.cpp checks it).
// the next layer's argument lines in ONE round trip before the barrier (the product's dd_touch_kernargs, at an offset inside the argument block)
__device__ __forceinline__ void chain_touch_next([[maybe_unused]] unsigned off) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(sizeof(ConvArgs) > 320 && sizeof(ConvArgs) <= 384, "six lines + the last dword");
    const unsigned long long ka = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr() + off;
    unsigned d0, d1, d2, d3, d4, d5, d6;
    asm volatile("s_load_dword %0, %7, 0\n\ts_load_dword %1, %7, 64\n\ts_load_dword %2, %7, 128\n\ts_load_dword %3, %7, 192\n\ts_load_dword %4, %7, 256\n\t"
                 "s_load_dword %5, %7, 320\n\ts_load_dword %6, %7, %8\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4), "=&s"(d5), "=&s"(d6) : "s"(ka), "n"((int)sizeof(ConvArgs) - 4) : "memory");
#endif
}
constexpr int LR_CHAIN_MAX = 6;
enum { LRK_GNSILU = 0, LRK_GNSILU_RES = 1, LRK_SILU = 2, LRK_RES = 3, LRK_FILM = 4, LRK_COLST = 5, LRK_COLSM = 6, LRK_COUNT = 7 };
struct LrChainArgs {       // (the header first: it shares the first argument lines with op[0], which are touched at kernel entry)
    int kind[LR_CHAIN_MAX];  // LRK_*
    int n;
    unsigned* bar;           // 8 counters (one per XCD), 128 bytes apart; monotonic, shared by every chain launch of the plan
    int* fault;              // sticky: 1 = a barrier timed out, 2 = a workgroup is not on XCD blockIdx % 8
    ConvArgs op[LR_CHAIN_MAX];
};
template <int MB>
struct LrChainSmem {
    static constexpr size_t mx(size_t a, size_t b) { return a > b ? a : b; }
    static constexpr size_t smem = mx(mx(mx(LrGeom<3, MB, PRO_GN_SILU, false, true>::smem, LrGeom<3, MB, PRO_NONE, false, true>::smem),
                                         mx(LrGeom<1, MB, PRO_NONE, false, true>::smem, LrGeom<1, MB, PRO_NONE, true, true>::smem)),
                                      LrGeom<1, MB, PRO_COLSM, false, false>::smem);
};
template <int MB>
__global__ __launch_bounds__(256) void conv_lr_chain_kernel(LrChainArgs c) {
    dd_touch_kernargs<(offsetof(LrChainArgs, op) + sizeof(ConvArgs) <= 448 ? offsetof(LrChainArgs, op) + sizeof(ConvArgs) : 448)>();  // the header's and the first layer's lines
    DDIF_DYN_SMEM(smem);
    const unsigned xcc = dd_xcc_id();
    if (xcc != (blockIdx.x & 7u) && threadIdx.x == 0) *c.fault = 2;
#pragma unroll 1
    for (int i = 0; i < (c.n & 255); ++i) {
        const ConvArgs& a = c.op[i];
        switch (c.kind[i]) {
        case LRK_GNSILU: conv_lr_body<3, MB, PRO_GN_SILU, 0, 0, true>(a, smem); break;
        case LRK_GNSILU_RES: conv_lr_body<3, MB, PRO_GN_SILU, EPI_RES, 0, true>(a, smem); break;
        case LRK_SILU: conv_lr_body<3, MB, PRO_NONE, EPI_SILU, 0, true>(a, smem); break;
        case LRK_RES: conv_lr_body<3, MB, PRO_NONE, EPI_RES, 0, true>(a, smem); break;
        case LRK_FILM: conv_lr_body<1, MB, PRO_NONE, EPI_FILM, 0, true>(a, smem); break;
        case LRK_COLST: conv_lr_body<1, MB, PRO_NONE, EPI_COLST, 0, true>(a, smem); break;
        default: conv_lr_body<1, MB, PRO_COLSM, 0, 0, false>(a, smem); break;
        }
        if (i + 1 < (c.n & 255)) {
            chain_touch_next((unsigned)(offsetof(LrChainArgs, op) + (unsigned)(i + 1) * sizeof(ConvArgs)));
#ifdef LR_CHAIN_DBG
            if (c.n & 256) continue;  // (microbenchmark: no barrier -- timing only, the results are wrong)
#endif
            xcd_barrier(c.bar + xcc * 32, c.fault);
        }
    }
}

}  // namespace ddif
