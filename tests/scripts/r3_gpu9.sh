python -m pytest tests/ -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3_t9.log
python bench.py > gpurun_out/r3_bench_1.json 2> gpurun_out/r3_bench_1.err
python tests/bench_gemm.py 2>&1 | grep -v amdgpu > gpurun_out/r3_gemm_bench5.txt
