python -m pytest tests/test_gemm_gpu.py tests/test_wide_attention_gpu.py -m gpu -q 2>&1 | tail -3 > gpurun_out/r3_t13.log
python tests/bench_gemm.py 2>&1 | grep -v amdgpu > gpurun_out/r3_gemm_bench7.txt
python tests/scripts/r3_gemm_trace.py > gpurun_out/r3_gemm_trace2.txt 2>&1
python bench.py --workload ddim --steps 20 --warmup 3 > gpurun_out/r3_ddim_2.json 2> gpurun_out/r3_ddim_2.err
