python -m pytest tests/test_gemm_gpu.py tests/test_diffusion_goldens_gpu.py tests/test_diffusion_parity_bars_gpu.py tests/test_lvdm_dropin.py tests/test_clip_golden.py -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r3_t17.log
python bench.py --workload ddim_guided --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r3_guided2.json 2> gpurun_out/r3_guided2.err
