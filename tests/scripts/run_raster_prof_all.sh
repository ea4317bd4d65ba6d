# Raster evidence in one go: kernel trace + stats of the raster bench command, then three separate PMC passes
# (FETCH_SIZE / WRITE_SIZE cannot share a pass on gfx950; SQ counters; never combined with trace domains).
# The multi-MB traces stay in /tmp on the box; only the per-kernel summaries come back.
# usage: bash tests/scripts/run_raster_prof_all.sh <tag>     -> gpurun_out/<tag>_*  (copy the summaries into profiles/)
TAG=${1:-r02_v1}
R=${GRAFT_REPO_ROOT:-$PWD}
O=/tmp/rprof_$TAG
rm -rf $O; mkdir -p $O $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --workload raster --steps 200 --warmup 20 > $R/gpurun_out/${TAG}_raster_bench_under_rocprof.json 2>/dev/null
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $R/gpurun_out/${TAG}_raster_kernel_stats.csv
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/tests/profile_raster.py 12 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/tests/profile_raster.py 12 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $O/pmc_sq -- python $R/tests/profile_raster.py 12 > /dev/null 2>&1
cd $R
mkdir -p gpurun_out/profiles_new
python tests/scripts/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_sq $TAG
cp profiles/pmc_traffic.json profiles/${TAG}_raster_pmc_*.csv gpurun_out/profiles_new/
tail -1 gpurun_out/${TAG}_raster_bench_under_rocprof.json | cut -c1-200
head -12 gpurun_out/${TAG}_raster_kernel_stats.csv | cut -c1-150
