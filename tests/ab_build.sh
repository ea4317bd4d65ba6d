#!/bin/bash
# usage: tests/ab_build.sh "<extra hipcc flags>"  -- rebuild libgvd_raster.so with extra flags (A/B experiments on the GPU box)
cd "$(dirname "$0")/../guidedvd-3dgs_amd"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-slp-vectorize -Wno-unused-function $1 -o lib/libgvd_raster.so csrc/capi.hip csrc/raster_forward.hip csrc/raster_backward.hip
