TAG=r03_guided_320 STEPS=3 WARMUP=1 bash tests/scripts/run_ddim_prof.sh --workload ddim_guided --ddim-height 320 --ddim-width 448 --no-cpu-baseline > gpurun_out/r3_prof_guided_320.log 2>&1
F=$(ls gpurun_out/prof_r03_guided_320/*/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$F" ] && python tests/scripts/prof_summary.py $F 60 > gpurun_out/r03_guided_320x448_summary.txt
S=$(ls gpurun_out/prof_r03_guided_320/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$S" ] && cp $S gpurun_out/r03_guided_320x448_kernel_stats.csv
rm -rf gpurun_out/prof_r03_guided_320
