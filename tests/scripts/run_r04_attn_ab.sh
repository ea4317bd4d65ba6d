# A/B of an attention-kernel change: lib/libgvd_diffusion_base.so (before) against lib/libgvd_diffusion.so; attention tests on the new one
mkdir -p gpurun_out
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids\|Gloo\]'
python -m pytest tests/test_diffusion_gpu.py -m gpu -q -x 2>&1 | grep -v "$F" | tail -3 > gpurun_out/r04_attn_ab_tests.log
for r in 1 2; do
python tests/scripts/bench_attn.py 2>/dev/null > gpurun_out/r04_attn_ab_new$r.txt
GVD_DIFFUSION_LIB=$PWD/guidedvd-3dgs_amd/lib/libgvd_diffusion_base.so python tests/scripts/bench_attn.py 2>/dev/null > gpurun_out/r04_attn_ab_base$r.txt
done
tail -2 gpurun_out/r04_attn_ab_tests.log
for r in 1 2; do echo "-- base $r"; cat gpurun_out/r04_attn_ab_base$r.txt; echo "-- new $r"; cat gpurun_out/r04_attn_ab_new$r.txt; done
