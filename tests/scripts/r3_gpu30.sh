bash tests/scripts/run_conv_lds_hunt.sh > gpurun_out/r3_hunt.log 2>&1
