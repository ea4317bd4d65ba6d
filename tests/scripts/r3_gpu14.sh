python -m pytest tests/ -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r3_t14.log
python bench.py > gpurun_out/r3_bench_2.json 2> gpurun_out/r3_bench_2.err
TAG=r03_ddim STEPS=3 WARMUP=1 bash tests/scripts/run_ddim_prof.sh --workload ddim --no-cpu-baseline > gpurun_out/r3_prof_ddim.log 2>&1
F=$(ls gpurun_out/prof_r03_ddim/*/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$F" ] && python tests/scripts/prof_summary.py $F 60 > gpurun_out/r03_ddim_576x1024_summary.txt
S=$(ls gpurun_out/prof_r03_ddim/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$S" ] && cp $S gpurun_out/r03_ddim_576x1024_kernel_stats.csv
rm -rf gpurun_out/prof_r03_ddim
