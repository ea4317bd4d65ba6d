python bench.py --workload ddim --steps 20 --warmup 3 --no-cpu-baseline --batch-cfg > gpurun_out/r3_ddim_batchcfg.json 2> gpurun_out/r3_ddim_batchcfg.err
python bench.py --workload ddim --steps 20 --warmup 3 --no-cpu-baseline --graph > gpurun_out/r3_ddim_graph.json 2> gpurun_out/r3_ddim_graph.err
python bench.py --workload ddim_guided --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r3_guided.json 2> gpurun_out/r3_guided.err
