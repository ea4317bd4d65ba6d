#!/bin/bash
# experiment build of libgvd_diffusion.so with the convolution kernel's timeline stamps (GVD_CONV_TRACE); select it with
# GVD_DIFFUSION_LIB=guidedvd-3dgs_amd/lib/libgvd_diffusion_ctrace.so   (tests/scripts/r4_conv_trace.py)
cd "$(dirname "$0")/../../guidedvd-3dgs_amd" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function -Wno-pass-failed -fno-honor-nans \
  -DGVD_CONV_TRACE "$@" -o lib/libgvd_diffusion_ctrace.so csrc/diffusion_kernels.hip csrc/attention_backward.hip csrc/conv_mfma.hip csrc/gemm_mfma.hip
