# Device ISA of every kernel source with the flags of __graft_entry__.build() (build container, no GPU): -> $1 (default /tmp/isa)/*.s,
# then tests/scripts/r6_kernel_resources.py summarises registers / LDS / scratch / spills / instruction mix into profiles/r06_kernel_resources.txt
O=${1:-/tmp/isa}; mkdir -p $O
C=guidedvd-3dgs_amd/csrc
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function --cuda-device-only -S"
for f in gemm_mfma attention_backward diffusion_kernels conv_mfma; do $H -Wno-pass-failed -fno-honor-nans -o $O/$f.s $C/$f.hip 2>/dev/null; done
for f in raster_forward raster_backward knn; do $H -ffp-contract=off -fno-slp-vectorize -o $O/$f.s $C/$f.hip 2>/dev/null; done
$H -o $O/ssim.s $C/ssim.hip 2>/dev/null
python tests/scripts/r6_kernel_resources.py $O
