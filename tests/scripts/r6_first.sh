# Round 6, first call of a session: full GPU suite + GEMM table vs hipBLASLt + default bench line + guided 320x448 line at HEAD.
mkdir -p gpurun_out
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids\|Gloo\]'
python tests/bench_gemm.py 2>&1 | grep -v "$F" > gpurun_out/r06_gemm_vs_hipblaslt.txt
GVD_BENCH_SHAPE_TABLE=gpurun_out/r06_ddim_by_shape.json python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_first.err
GVD_BENCH_SHAPE_TABLE=gpurun_out/r06_guided_by_shape.json python bench.py --workload ddim_guided --ddim-height 320 --ddim-width 448 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r06_bench_guided_320x448.json 2>> gpurun_out/r06_first.err
python -m pytest tests/ -m gpu -q -x 2>&1 | grep -v "$F" | tail -15 > gpurun_out/r06_tests_first.log
cat gpurun_out/r06_gemm_vs_hipblaslt.txt; cut -c1-600 gpurun_out/r06_bench_default.json; cut -c1-300 gpurun_out/r06_bench_guided_320x448.json; tail -5 gpurun_out/r06_tests_first.log; tail -5 gpurun_out/r06_first.err
