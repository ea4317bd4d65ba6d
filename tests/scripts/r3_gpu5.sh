for d in 0 1 2 3 4 5 7; do echo "=== GVD_GEMM_DBG=$d"; GVD_GEMM_DBG=$d python tests/bench_gemm.py 2>&1 | grep -v amdgpu | awk '{print $1,$2,$3,$4,$5}' ; done > gpurun_out/r3_gemm_dbg.txt
python -m pytest tests/test_gemm_gpu.py tests/test_diffusion_parity_bars_gpu.py -m gpu -q -s 2>&1 | grep "ratio\|passed\|failed\|Error" > gpurun_out/r3_t5.log
