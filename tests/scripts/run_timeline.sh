R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_fused -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline > /dev/null 2>&1
f=$(ls $R/gpurun_out/tl_fused/*/*kernel_trace.csv | head -1); python $R/tests/scripts/timeline_gaps.py $f 400
GVD_RASTER_FUSED_SORT=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_sep -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline > /dev/null 2>&1
f=$(ls $R/gpurun_out/tl_sep/*/*kernel_trace.csv | head -1); python $R/tests/scripts/timeline_gaps.py $f 400
