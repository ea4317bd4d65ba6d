# rocprofv3 kernel trace of a bench.py run -> per-kernel summary (the multi-MB trace itself is not kept).
# usage: TAG=name bash tests/scripts/run_prof_csv.sh <bench.py args>      output: gpurun_out/<TAG>_summary.txt, <TAG>_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=/tmp/prof_${TAG:-run}
rm -rf $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $R/bench.py "$@" > $R/gpurun_out/${TAG:-run}_bench.json 2> /dev/null
T=$(find $OUT -name '*kernel_trace.csv' | head -1)
S=$(find $OUT -name '*kernel_stats.csv' | head -1)
python $R/tests/scripts/prof_summary.py $T 60 > $R/gpurun_out/${TAG:-run}_summary.txt 2>&1
cp $S $R/gpurun_out/${TAG:-run}_kernel_stats.csv
head -30 $R/gpurun_out/${TAG:-run}_summary.txt
