set -x
python -m pytest tests/test_guided_schedule.py tests/test_raster_gpu.py tests/test_raster_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3_t1.log
python -m pytest tests/test_diffusion_gpu.py tests/test_conv_gpu.py tests/test_diffusion_goldens_gpu.py -m gpu -x -q 2>&1 | tail -8 >> gpurun_out/r3_t1.log
python bench.py > gpurun_out/r3_bench_0.json 2> gpurun_out/r3_bench_0.err
GVD_DIST_BACKEND=gloo timeout 900 python bench.py --workload config4 --gpus 2 --c4-iters 60 --c4-rounds 2 --c4-ddim-steps 2 --c4-deliver-after 60 > gpurun_out/r3_c4_2rank_dry.json 2> gpurun_out/r3_c4_2rank_dry.err
tail -3 gpurun_out/r3_t1.log
