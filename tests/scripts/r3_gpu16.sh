TAG=r03_guided STEPS=2 WARMUP=1 bash tests/scripts/run_ddim_prof.sh --workload ddim_guided --no-cpu-baseline > gpurun_out/r3_prof_guided.log 2>&1
F=$(ls gpurun_out/prof_r03_guided/*/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$F" ] && python tests/scripts/prof_summary.py $F 70 > gpurun_out/r03_guided_576x1024_summary.txt
S=$(ls gpurun_out/prof_r03_guided/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$S" ] && cp $S gpurun_out/r03_guided_576x1024_kernel_stats.csv
rm -rf gpurun_out/prof_r03_guided
python bench.py --steps 20 --warmup 5 --ddim-steps 5 --no-cpu-baseline > gpurun_out/r3_bench_3.json 2> gpurun_out/r3_bench_3.err
