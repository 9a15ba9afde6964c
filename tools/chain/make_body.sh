#!/bin/bash
# tools/chain/_gen/kernels_lr_body.h: the product's conv_lr_kernel (dif-pan_amd/csrc/kernels_lr.h) with its __global__ head turned into a device function
# conv_lr_body(const ConvArgs&, char* smem), so that tools/chain/kernels_lr_chain.h can run several layers inside one kernel.  Generated, never committed
# (the product's kernel text stays the single source); run from the repository root.
set -e
mkdir -p tools/chain/_gen
sed -e 's/^#pragma once$/#pragma once\n#define DDIF_LR_BODY_GENERATED 1/' \
    -e 's/^__global__ __launch_bounds__(256) void conv_lr_kernel(ConvArgs a) {$/__device__ __forceinline__ void conv_lr_body(const ConvArgs\& a, char* smem) {/' \
    -e '/^    dd_touch_kernargs<sizeof(ConvArgs)>();/d' \
    -e '/^    DDIF_DYN_SMEM(smem);$/d' \
    -e 's/struct LrGeom/struct LrGeomB/; s/LrGeom</LrGeomB</g' \
    dif-pan_amd/csrc/kernels_lr.h > tools/chain/_gen/kernels_lr_body.h
grep -q "void conv_lr_body(const ConvArgs& a, char\* smem) {" tools/chain/_gen/kernels_lr_body.h
! grep -q "DDIF_DYN_SMEM\|dd_touch_kernargs" tools/chain/_gen/kernels_lr_body.h
