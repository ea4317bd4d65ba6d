# A/B of libgvd_diffusion.so against a saved library on tests/bench_gemm.py, alternating passes on one box.  usage: r6_gemm_ab.sh <other.so> <tag>
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids\|Gloo\]'
O=gpurun_out/r06_gemm_ab_$2.txt
: > $O
for pass in 1 2; do
  echo "== new (pass $pass)" >> $O; python tests/bench_gemm.py 2>&1 | grep -v "$F" >> $O
  echo "== $1 (pass $pass)" >> $O; GVD_DIFFUSION_LIB=$PWD/$1 python tests/bench_gemm.py 2>&1 | grep -v "$F" >> $O
done
