# rocprofv3 evidence for the raster path: (1) kernel trace + stats of the default bench command,
# (2)+(3) separate PMC passes (FETCH_SIZE, WRITE_SIZE cannot share a pass on gfx950; no trace domains with --pmc).
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_raster -- python $R/bench.py --steps 200 --warmup 20 > $R/gpurun_out/raster_bench_under_rocprof.json 2>/dev/null
tail -1 $R/gpurun_out/raster_bench_under_rocprof.json | cut -c1-300
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python $R/tests/profile_raster.py 12 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python $R/tests/profile_raster.py 12 > /dev/null 2>&1
ls $R/gpurun_out/pmc_fetch/* $R/gpurun_out/pmc_write/* | head
