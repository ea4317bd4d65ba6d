for af in 5 13 25; do
python bench.py --workload ddim_guided --ddim-height 320 --ddim-width 448 --steps 5 --warmup 2 --no-cpu-baseline --ae-frames $af > gpurun_out/r3_guided_320_af$af.json 2>> gpurun_out/r3_guided_320.err
done
python bench.py --workload ddim --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/r3_ddim_graph.json 2>> gpurun_out/r3_guided_320.err
python bench.py --workload ddim --steps 20 --warmup 2 --no-cpu-baseline --eager > gpurun_out/r3_ddim_eager.json 2>> gpurun_out/r3_guided_320.err
python bench.py --workload ddim --ddim-height 320 --ddim-width 448 --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/r3_ddim_320_graph.json 2>> gpurun_out/r3_guided_320.err
python bench.py --workload ddim --ddim-height 320 --ddim-width 448 --steps 20 --warmup 2 --no-cpu-baseline --eager > gpurun_out/r3_ddim_320_eager.json 2>> gpurun_out/r3_guided_320.err
