# Round 4: evaluation of a convolution-kernel change (tests, VAE ablation, small-level microbench, timeline, tile sweep, both bench lines)
mkdir -p gpurun_out
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids\|Gloo\]'
python -m pytest tests/test_conv_gpu.py tests/test_diffusion_goldens_gpu.py tests/test_diffusion_parity_bars_gpu.py tests/test_diffusion_gpu.py -m gpu -q -x 2>&1 | grep -v "$F" | tail -8 > gpurun_out/r04_seventh_tests.log
python tests/scripts/r4_conv_ablate.py > gpurun_out/r04_conv_ablate_v4.txt 2>> gpurun_out/r04_seventh.err
python tests/scripts/r4_small_conv.py > gpurun_out/r04_small_conv_v4.txt 2>> gpurun_out/r04_seventh.err
GVD_DIFFUSION_LIB=$PWD/guidedvd-3dgs_amd/lib/libgvd_diffusion_ctrace.so python tests/scripts/r4_conv_trace.py > gpurun_out/r04_conv_trace_v4.txt 2>> gpurun_out/r04_seventh.err
rm -f gpurun_out/r04_tile_sweep_v4.txt
for c in rule 0 1 2 4; do
  if [ $c = rule ]; then python tests/scripts/r4_tile_sweep.py 2>> gpurun_out/r04_seventh.err >> gpurun_out/r04_tile_sweep_v4.txt
  else GVD_CONV_FORCE_CFG=$c python tests/scripts/r4_tile_sweep.py 2>> gpurun_out/r04_seventh.err >> gpurun_out/r04_tile_sweep_v4.txt; fi
done
python bench.py --workload ddim_guided --ddim-height 320 --ddim-width 448 --steps 6 --warmup 2 --no-cpu-baseline 2>> gpurun_out/r04_seventh.err | cut -c1-200 > gpurun_out/r04_guided_v7.json
python bench.py --workload ddim --steps 10 --warmup 2 --no-cpu-baseline 2>> gpurun_out/r04_seventh.err | cut -c1-200 > gpurun_out/r04_ddim_v7.json
tail -3 gpurun_out/r04_seventh_tests.log; cat gpurun_out/r04_conv_ablate_v4.txt gpurun_out/r04_small_conv_v4.txt gpurun_out/r04_conv_trace_v4.txt gpurun_out/r04_tile_sweep_v4.txt gpurun_out/r04_guided_v7.json gpurun_out/r04_ddim_v7.json
