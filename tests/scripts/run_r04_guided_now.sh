# Round 4 (second session): the guided 320x448 step at HEAD -- torch-op table, bench line with per-shape table, rocprofv3 kernel summary
mkdir -p gpurun_out
GVD_BENCH_TORCH_PROFILE=gpurun_out/r04b_guided_320_torch_ops.txt python bench.py --workload ddim_guided --ddim-height 320 --ddim-width 448 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r4b_tp.json 2> gpurun_out/r4b_tp.err
GVD_BENCH_SHAPE_TABLE=gpurun_out/r04b_guided_by_shape.json python bench.py --workload ddim_guided --ddim-height 320 --ddim-width 448 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r04b_bench_guided_320x448.json 2>> gpurun_out/r4b_tp.err
tag=r04b_guided_320x448
TAG=$tag STEPS=3 WARMUP=1 bash tests/scripts/run_ddim_prof.sh --workload ddim_guided --ddim-height 320 --ddim-width 448 --no-cpu-baseline > gpurun_out/prof_g.log 2>&1
T=$(ls gpurun_out/prof_$tag/*/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$T" ] && python tests/scripts/prof_summary.py $T 90 > gpurun_out/${tag}_summary.txt
rm -rf gpurun_out/prof_$tag
cut -c1-400 gpurun_out/r04b_bench_guided_320x448.json; head -40 gpurun_out/${tag}_summary.txt | cut -c1-160
