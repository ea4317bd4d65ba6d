python -m pytest "tests/test_guided_schedule.py::test_raster_rank_and_diffusion_rank_on_one_gpu_match_the_single_process_run" -m gpu -x -q 2>&1 | grep -v "Gloo\|^$" | tail -70 > gpurun_out/r3_t2a.log
python -m pytest tests/test_raster_gpu.py tests/test_raster_fuzz_gpu.py -m gpu -q 2>&1 | tail -15 > gpurun_out/r3_t2b.log
python -m pytest tests/test_diffusion_parity_bars_gpu.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r3_t2c.log
