# experiment: does starting the persistent GEMM workgroups out of phase (so that their epilogue store bursts do not coincide) shorten a launch?
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids\|Gloo\]'
O=gpurun_out/r06_gemm_stagger.txt
: > $O
for cfg in "0 1" "8000 2" "16000 2" "6000 4" "12000 4" "3000 8" "0 1"; do
  set -- $cfg
  echo "== stagger $1 cycles x phases $2" >> $O
  GVD_GEMM_STAGGER=$1 GVD_GEMM_PHASES=$2 python tests/bench_gemm.py 2>&1 | grep -v "$F" | cut -c1-50 >> $O
done
