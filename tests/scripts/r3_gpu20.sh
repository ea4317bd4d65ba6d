python -m pytest tests/ -m gpu -q 2>&1 | tail -6 > gpurun_out/r3_t20.log
GVD_BENCH_SHAPE_TABLE=gpurun_out/r03_ddim_by_shape.json python bench.py > gpurun_out/r3_bench_final.json 2> gpurun_out/r3_bench_final.err
python tests/bench_conv.py > gpurun_out/r03_conv_microbench.txt 2>&1
TAG=r03_guided STEPS=2 WARMUP=1 bash tests/scripts/run_ddim_prof.sh --workload ddim_guided --no-cpu-baseline > gpurun_out/r3_prof_guided.log 2>&1
F=$(ls gpurun_out/prof_r03_guided/*/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$F" ] && python tests/scripts/prof_summary.py $F 70 > gpurun_out/r03_guided_576x1024_summary.txt
S=$(ls gpurun_out/prof_r03_guided/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$S" ] && cp $S gpurun_out/r03_guided_576x1024_kernel_stats.csv
rm -rf gpurun_out/prof_r03_guided
TAG=r03_ddim STEPS=3 WARMUP=1 bash tests/scripts/run_ddim_prof.sh --workload ddim --no-cpu-baseline > gpurun_out/r3_prof_ddim.log 2>&1
F=$(ls gpurun_out/prof_r03_ddim/*/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$F" ] && python tests/scripts/prof_summary.py $F 60 > gpurun_out/r03_ddim_576x1024_summary.txt
S=$(ls gpurun_out/prof_r03_ddim/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$S" ] && cp $S gpurun_out/r03_ddim_576x1024_kernel_stats.csv
rm -rf gpurun_out/prof_r03_ddim
