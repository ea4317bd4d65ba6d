python tests/scripts/autograd_node_census.py > gpurun_out/r03_autograd_census.txt 2>&1
