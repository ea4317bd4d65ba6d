python bench.py --workload ddim --batch-cfg --steps 10 --warmup 2 --no-cpu-baseline 2>> gpurun_out/r04_evidence.err | cut -c1-220 > gpurun_out/r04_ddim_batchcfg.json
cat gpurun_out/r04_ddim_batchcfg.json
