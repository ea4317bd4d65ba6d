# Builds guidedvd-3dgs_amd/lib/libgvd_raster_trace.so with the per-workgroup stamps of the blend kernels compiled in
# (tests/scripts/r5_bwd_trace.py, r5_fwd_trace.py; run them with GVD_RASTER_LIB=<that file>).  Not part of the product build.
cd "$(dirname "$0")/../../guidedvd-3dgs_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off \
  -fno-slp-vectorize -Wall -Wno-unused-function -DGVD_RBWD_TRACE -DGVD_RFWD_TRACE -o ../lib/libgvd_raster_trace.so capi.hip raster_forward.hip raster_backward.hip
