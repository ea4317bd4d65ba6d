for b in 0 4 5; do echo "=== blocks $b"; GVD_GEMM_BLOCKS=$b python tests/bench_gemm.py 2>&1 | grep -v amdgpu; done > gpurun_out/r3_gemm_bench6.txt
python -m pytest tests/test_gemm_gpu.py tests/test_diffusion_gpu.py -m gpu -q 2>&1 | tail -3 > gpurun_out/r3_t12.log
