#!/bin/bash
# experiment builds of libgvd_diffusion.so with parts of the GEMM kernel compiled out (GVD_GEMM_DBG bits: 1 no DMA in the K loop,
# 2 no MFMAs, 4 no epilogue); select one at run time with GVD_DIFFUSION_LIB=guidedvd-3dgs_amd/lib/libgvd_diffusion_dbgN.so
cd "$(dirname "$0")/../../guidedvd-3dgs_amd" || exit 1
for d in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function -Wno-pass-failed -fno-honor-nans \
    -DGVD_GEMM_DBG=$d -o lib/libgvd_diffusion_dbg$d.so csrc/diffusion_kernels.hip csrc/attention_backward.hip csrc/conv_mfma.hip csrc/gemm_mfma.hip &
done
wait
