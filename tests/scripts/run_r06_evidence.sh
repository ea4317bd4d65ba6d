# Round 6 evidence at HEAD (everything the committed numbers come from): full GPU suite, the default bench line with the per-shape table,
# guided / configs[3] / pipeline lines, kernel summaries of the unguided and guided steps (rocprofv3 --kernel-trace --stats), the MFMA /
# traffic counter passes of the diffusion kernels, the raster kernel stats + three PMC passes, lane statistics of the blend kernels.
mkdir -p gpurun_out
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids\|Gloo\]'
python -m pytest tests/ -m gpu -q 2>&1 | grep -v "$F" | tail -6 > gpurun_out/r06_round_end_tests.log
python tests/bench_gemm.py 2>&1 | grep -v "$F" > gpurun_out/r06_gemm_vs_hipblaslt.txt
GVD_BENCH_SHAPE_TABLE=gpurun_out/r06_ddim_by_shape.json python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_evidence.err
GVD_BENCH_SHAPE_TABLE=gpurun_out/r06_guided_by_shape.json python bench.py --workload ddim_guided --ddim-height 320 --ddim-width 448 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r06_bench_guided_320x448.json 2>> gpurun_out/r06_evidence.err
python bench.py --workload ddim_guided --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r06_bench_guided.json 2>> gpurun_out/r06_evidence.err
python bench.py --workload config4 --no-cpu-baseline > gpurun_out/r06_config4.json 2>> gpurun_out/r06_evidence.err
python bench.py --workload pipeline --no-cpu-baseline > gpurun_out/r06_pipeline.json 2>> gpurun_out/r06_evidence.err
for w in ddim guided320 guided; do
  case $w in
    ddim) A="--workload ddim"; tag=r06_ddim_576x1024;;
    guided320) A="--workload ddim_guided --ddim-height 320 --ddim-width 448"; tag=r06_guided_320x448;;
    guided) A="--workload ddim_guided"; tag=r06_guided_576x1024;;
  esac
  TAG=$tag STEPS=3 WARMUP=1 bash tests/scripts/run_ddim_prof.sh $A --no-cpu-baseline > gpurun_out/prof_$w.log 2>&1
  T=$(ls gpurun_out/prof_$tag/*/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$T" ] && python tests/scripts/prof_summary.py $T 70 > gpurun_out/${tag}_summary.txt
  S=$(ls gpurun_out/prof_$tag/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$S" ] && cp $S gpurun_out/${tag}_kernel_stats.csv
  rm -rf gpurun_out/prof_$tag
done
TAG=r06 bash tests/scripts/run_diff_pmc.sh > gpurun_out/r06_diff_pmc.log 2>&1
bash tests/scripts/run_raster_prof_all.sh r06 > gpurun_out/r06_raster_prof.log 2>&1
bash tests/scripts/build_raster_trace_lib.sh > /dev/null 2>&1 && { GVD_RASTER_LIB=$PWD/guidedvd-3dgs_amd/lib/libgvd_raster_trace.so python tests/scripts/r5_bwd_trace.py 2>&1 | grep -v "$F" > gpurun_out/r06_bwd_trace_final.txt; GVD_RASTER_LIB=$PWD/guidedvd-3dgs_amd/lib/libgvd_raster_trace.so python tests/scripts/r5_fwd_trace.py 2>&1 | grep -v "$F" > gpurun_out/r06_fwd_trace_final.txt; }
bash tests/scripts/r6_guard_all.sh > /dev/null 2>&1
python tests/scripts/lane_stats.py > gpurun_out/r06_lane_stats.txt 2>> gpurun_out/r06_evidence.err
tail -3 gpurun_out/r06_round_end_tests.log; cut -c1-300 gpurun_out/r06_bench_default.json; cut -c1-200 gpurun_out/r06_bench_guided_320x448.json; cat gpurun_out/r06_lane_stats.txt
