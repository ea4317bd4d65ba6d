python tests/scripts/r3_pair_probe.py > gpurun_out/r3_pair_probe.txt 2>&1
