# Round 4, third GPU call: batched temporal launches + distance-2 weight prefetch (A/B against the -DGVD_CONV_WPF=1 build), configs[4]
# test with the single-step probe, per-shape table of the guided step.
mkdir -p gpurun_out
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids\|Gloo\]'
python -m pytest tests/test_conv_gpu.py tests/test_diffusion_goldens_gpu.py tests/test_diffusion_parity_bars_gpu.py tests/test_diffusion_gpu.py tests/test_lvdm_dropin.py tests/test_diffusion_trajectory_gpu.py tests/test_ddim_parallel_gloo.py -m gpu -q -x 2>&1 | grep -v "$F" | tail -25 > gpurun_out/r04_third_tests.log
python tests/scripts/r4_small_conv.py > gpurun_out/r04_small_conv_wpf2.txt 2>> gpurun_out/r04_third.err
GVD_DIFFUSION_LIB=$PWD/guidedvd-3dgs_amd/lib/libgvd_diffusion_wpf1.so python tests/scripts/r4_small_conv.py > gpurun_out/r04_small_conv_wpf1.txt 2>> gpurun_out/r04_third.err
python bench.py --workload ddim_guided --ddim-height 320 --ddim-width 448 --steps 6 --warmup 2 --no-cpu-baseline 2>> gpurun_out/r04_third.err | cut -c1-200 > gpurun_out/r04_guided_v3.json
GVD_DIFFUSION_LIB=$PWD/guidedvd-3dgs_amd/lib/libgvd_diffusion_wpf1.so python bench.py --workload ddim_guided --ddim-height 320 --ddim-width 448 --steps 6 --warmup 2 --no-cpu-baseline 2>> gpurun_out/r04_third.err | cut -c1-200 > gpurun_out/r04_guided_v3_wpf1.json
GVD_BENCH_SHAPE_TABLE=gpurun_out/r04_guided_by_shape.json python bench.py --workload ddim_guided --ddim-height 320 --ddim-width 448 --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>> gpurun_out/r04_third.err
python bench.py --workload ddim --steps 10 --warmup 2 --no-cpu-baseline 2>> gpurun_out/r04_third.err | cut -c1-200 > gpurun_out/r04_ddim_v3.json
python -m pytest "tests/test_guided_schedule.py::test_config5_eight_ranks_on_one_gpu_with_hip_kernels" -m gpu -q -s 2>&1 | grep -v "$F" | tail -12 > gpurun_out/r04_config5.log
tail -4 gpurun_out/r04_third_tests.log; cat gpurun_out/r04_small_conv_wpf2.txt gpurun_out/r04_small_conv_wpf1.txt; cat gpurun_out/r04_guided_v3.json gpurun_out/r04_guided_v3_wpf1.json gpurun_out/r04_ddim_v3.json; tail -4 gpurun_out/r04_config5.log
