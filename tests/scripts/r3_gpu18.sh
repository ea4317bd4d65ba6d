python -m pytest tests/ -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r3_t18.log
python bench.py --workload ddim_guided --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r3_guided3.json 2> gpurun_out/r3_guided3.err
