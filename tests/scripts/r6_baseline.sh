# Round 6 baseline at the round's first commit: GEMM table against hipBLASLt with the library's kernel names (macro tiles) from a kernel trace,
# the default bench line, the guided 320x448 line.
mkdir -p gpurun_out
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids\|Gloo\]'
python tests/bench_gemm.py 2>&1 | grep -v "$F" > gpurun_out/r06_base_gemm.txt
R=$PWD
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_gemm -- python $R/tests/bench_gemm.py > /dev/null 2>&1 )
S=$(ls gpurun_out/prof_gemm/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$S" ] && cp $S gpurun_out/r06_base_gemm_kernel_stats.csv
rm -rf gpurun_out/prof_gemm
GVD_BENCH_SHAPE_TABLE=gpurun_out/r06_base_ddim_by_shape.json python bench.py > gpurun_out/r06_base_bench_default.json 2> gpurun_out/r06_base.err
GVD_BENCH_SHAPE_TABLE=gpurun_out/r06_base_guided_by_shape.json python bench.py --workload ddim_guided --ddim-height 320 --ddim-width 448 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r06_base_bench_guided_320x448.json 2>> gpurun_out/r06_base.err
cat gpurun_out/r06_base_gemm.txt; cut -c1-400 gpurun_out/r06_base_bench_default.json; cut -c1-300 gpurun_out/r06_base_bench_guided_320x448.json
