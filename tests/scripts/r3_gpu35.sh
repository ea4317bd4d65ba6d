python bench.py --workload ddim --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/r3_ddim_e.json 2> gpurun_out/r3_e.err
python bench.py --workload ddim --steps 20 --warmup 2 --no-cpu-baseline --batch-cfg > gpurun_out/r3_ddim_e_bcfg.json 2>> gpurun_out/r3_e.err
python bench.py --workload ddim --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/r3_ddim_e2.json 2>> gpurun_out/r3_e.err
python bench.py --workload ddim --steps 20 --warmup 2 --no-cpu-baseline --batch-cfg --ddim-height 320 --ddim-width 448 > gpurun_out/r3_ddim_e320_bcfg.json 2>> gpurun_out/r3_e.err
python bench.py --workload ddim --steps 20 --warmup 2 --no-cpu-baseline --ddim-height 320 --ddim-width 448 > gpurun_out/r3_ddim_e320.json 2>> gpurun_out/r3_e.err
