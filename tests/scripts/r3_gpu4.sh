python -m pytest tests/test_gemm_gpu.py -m gpu -q 2>&1 | tail -25 > gpurun_out/r3_t4a.log
python tests/bench_gemm.py > gpurun_out/r3_gemm_bench2.txt 2>&1
python -m pytest "tests/test_guided_schedule.py::test_raster_rank_and_diffusion_rank_on_one_gpu_match_the_single_process_run" -m gpu -x -q 2>&1 | grep -v "Gloo\|^$" | tail -30 > gpurun_out/r3_t4b.log
python -m pytest tests/test_diffusion_parity_bars_gpu.py -m gpu -q -s 2>&1 | grep "ratio\|passed\|failed" > gpurun_out/r3_t4c.log
python -m pytest tests/test_diffusion_gpu.py tests/test_diffusion_goldens_gpu.py tests/test_lvdm_dropin.py -m gpu -q 2>&1 | tail -25 > gpurun_out/r3_t4d.log
