python -m pytest tests/test_gemm_gpu.py -m gpu -q 2>&1 | tail -3 > gpurun_out/r3_t7a.log
for v in 1 0; do for l2 in 1572864 1 ; do echo "=== variant $v L2 budget $l2"; GVD_GEMM_L2_BYTES=$l2 GVD_GEMM_VARIANT=$v python tests/bench_gemm.py 2>&1 | grep -v amdgpu; done; done > gpurun_out/r3_gemm_bench4.txt
