# everything the round's committed numbers come from, at HEAD: full GPU suite, default bench line, guided / configs[3] / pipeline lines,
# kernel summaries of the unguided and guided steps (rocprofv3 --kernel-trace --stats)
python -m pytest tests/ -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids" | tail -4 > gpurun_out/round_end_tests.log
GVD_BENCH_SHAPE_TABLE=gpurun_out/r03_ddim_by_shape.json python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/round_end.err
python bench.py --workload ddim_guided --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_guided.json 2>> gpurun_out/round_end.err
python bench.py --workload ddim_guided --ddim-height 320 --ddim-width 448 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03_bench_guided_320x448.json 2>> gpurun_out/round_end.err
python bench.py --workload config4 --no-cpu-baseline > gpurun_out/r03_config4.json 2>> gpurun_out/round_end.err
python bench.py --workload pipeline --no-cpu-baseline > gpurun_out/r03_pipeline.json 2>> gpurun_out/round_end.err
for w in ddim ddim_guided; do
  tag=r03_${w/ddim_guided/guided}; tag=${tag/r03_ddim/r03_ddim}
  TAG=$tag STEPS=2 WARMUP=1 bash tests/scripts/run_ddim_prof.sh --workload $w --no-cpu-baseline > gpurun_out/prof_$w.log 2>&1
  F=$(ls gpurun_out/prof_$tag/*/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$F" ] && python tests/scripts/prof_summary.py $F 60 > gpurun_out/${tag}_576x1024_summary.txt
  S=$(ls gpurun_out/prof_$tag/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$S" ] && cp $S gpurun_out/${tag}_576x1024_kernel_stats.csv
  rm -rf gpurun_out/prof_$tag
done
