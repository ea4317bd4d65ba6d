echo "=== ku2 (default)" > gpurun_out/r3_gemm_ku.txt; python tests/bench_gemm.py 2>&1 | grep -v amdgpu >> gpurun_out/r3_gemm_ku.txt
echo "=== ku4" >> gpurun_out/r3_gemm_ku.txt; GVD_DIFFUSION_LIB=$PWD/guidedvd-3dgs_amd/lib/libgvd_diffusion_ku4.so python tests/bench_gemm.py 2>&1 | grep -v amdgpu >> gpurun_out/r3_gemm_ku.txt
python bench.py --workload ddim_guided --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r3_guided4.json 2> gpurun_out/r3_guided4.err
python bench.py --workload config4 > gpurun_out/r3_config4.json 2> gpurun_out/r3_config4.err
