# rocprofv3 kernel summaries of the bench's TIMED REGION only (bench.py brackets it with k_profile_marker launches under GVD_BENCH_MARKERS;
# run_ddim_prof.sh sets it): unguided 576x1024, guided 320x448, guided 576x1024.  usage: r5_ddim_prof_steps_only.sh [tag-prefix]
P=${1:-r05b}
for w in ddim guided320 guided; do
  case $w in
    ddim) A="--workload ddim"; tag=${P}_ddim_576x1024;;
    guided320) A="--workload ddim_guided --ddim-height 320 --ddim-width 448"; tag=${P}_guided_320x448;;
    guided) A="--workload ddim_guided"; tag=${P}_guided_576x1024;;
  esac
  TAG=$tag STEPS=3 WARMUP=1 bash tests/scripts/run_ddim_prof.sh $A --no-cpu-baseline > gpurun_out/prof_$w.log 2>&1
  T=$(ls gpurun_out/prof_$tag/*/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$T" ] && python tests/scripts/prof_summary.py $T 70 > gpurun_out/${tag}_summary.txt
  rm -rf gpurun_out/prof_$tag
  head -2 gpurun_out/${tag}_summary.txt; grep -A14 "not this package" gpurun_out/${tag}_summary.txt
done
