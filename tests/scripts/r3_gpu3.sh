python -m pytest tests/test_gemm_gpu.py -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r3_t3a.log
python tests/bench_gemm.py > gpurun_out/r3_gemm_bench.txt 2>&1
