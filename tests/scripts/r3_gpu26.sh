TAG=r03 bash tests/scripts/run_diff_pmc.sh > gpurun_out/r3_pmc.log 2>&1
