# Round 4, second GPU call: configs[4] test with the fp32 anchor; the gradient hand-over cells (ops.GradCell) against the goldens / parity
# bars and A/B on the guided step; torch-op table of the guided step; convolution ablation on the VAE shapes.
mkdir -p gpurun_out
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids\|Gloo\]'
python -m pytest "tests/test_guided_schedule.py::test_config5_eight_ranks_on_one_gpu_with_hip_kernels" -m gpu -q -s 2>&1 | grep -v "$F" | tail -30 > gpurun_out/r04_config5.log
python -m pytest tests/test_diffusion_goldens_gpu.py tests/test_diffusion_parity_bars_gpu.py tests/test_diffusion_gpu.py tests/test_conv_gpu.py tests/test_gemm_gpu.py tests/test_lvdm_dropin.py tests/test_diffusion_trajectory_gpu.py -m gpu -q -s 2>&1 | grep -v "$F" | grep "ratio\|passed\|failed\|Error\|error" | tail -40 > gpurun_out/r04_cells_tests.log
for c in 0 1; do
  GVD_GRAD_CELLS=$c python bench.py --workload ddim_guided --ddim-height 320 --ddim-width 448 --steps 6 --warmup 2 --no-cpu-baseline 2>> gpurun_out/r04_second.err | cut -c1-160 > gpurun_out/r04_guided_cells$c.json
done
GVD_BENCH_TORCH_PROFILE=gpurun_out/r04_guided_320_torch_ops.txt python bench.py --workload ddim_guided --ddim-height 320 --ddim-width 448 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>> gpurun_out/r04_second.err
python tests/scripts/r4_conv_ablate.py > gpurun_out/r04_conv_ablate.txt 2>> gpurun_out/r04_second.err
tail -4 gpurun_out/r04_config5.log; tail -5 gpurun_out/r04_cells_tests.log; cat gpurun_out/r04_guided_cells0.json gpurun_out/r04_guided_cells1.json; cat gpurun_out/r04_conv_ablate.txt
