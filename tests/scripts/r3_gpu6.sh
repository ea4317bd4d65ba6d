python -m pytest tests/test_gemm_gpu.py tests/test_wide_attention_gpu.py -m gpu -q 2>&1 | tail -12 > gpurun_out/r3_t6a.log
GVD_GEMM_VARIANT=1 python -m pytest tests/test_gemm_gpu.py -m gpu -q 2>&1 | tail -5 >> gpurun_out/r3_t6a.log
for v in 0 1; do echo "=== variant $v"; GVD_GEMM_VARIANT=$v python tests/bench_gemm.py 2>&1 | grep -v amdgpu; done > gpurun_out/r3_gemm_bench3.txt
python -m pytest tests/test_diffusion_parity_bars_gpu.py -m gpu -q -s 2>&1 | grep "ratio\|passed\|failed\|Error\|assert" > gpurun_out/r3_t6c.log
