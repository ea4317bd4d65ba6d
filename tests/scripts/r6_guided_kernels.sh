# kernel-time summary of the guided 320x448 step (timed region only) with an environment switch on / off.  usage: r6_guided_kernels.sh VAR
R=$PWD
for v in off on; do
  if [ $v = on ]; then export $1=1; else unset $1; fi
  ( cd /tmp && export TMPDIR=/tmp && GVD_BENCH_MARKERS=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_g_$v -- python $R/bench.py --workload ddim_guided --ddim-height 320 --ddim-width 448 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 )
  T=$(ls gpurun_out/prof_g_$v/*/*kernel_trace.csv | head -1)
  python tests/scripts/prof_summary.py $T 40 > gpurun_out/r06_guided_kernels_$1_$v.txt
  rm -rf gpurun_out/prof_g_$v
done
