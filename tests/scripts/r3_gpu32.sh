python bench.py --workload ddim --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/r3_ddim_d.json 2> gpurun_out/r3_d.err
python bench.py --workload ddim_guided --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3_guided_576_d.json 2>> gpurun_out/r3_d.err
python bench.py --workload ddim_guided --ddim-height 320 --ddim-width 448 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3_guided_320_d.json 2>> gpurun_out/r3_d.err
