# Round 4, fourth GPU call: epilogue prefetch (tests + ablation), convolution workgroup timeline, tile-configuration sweep
mkdir -p gpurun_out
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids\|Gloo\]'
python -m pytest tests/test_conv_gpu.py tests/test_diffusion_goldens_gpu.py tests/test_diffusion_parity_bars_gpu.py tests/test_diffusion_gpu.py -m gpu -q -x 2>&1 | grep -v "$F" | tail -8 > gpurun_out/r04_fourth_tests.log
python tests/scripts/r4_conv_ablate.py > gpurun_out/r04_conv_ablate_v2.txt 2>> gpurun_out/r04_fourth.err
GVD_DIFFUSION_LIB=$PWD/guidedvd-3dgs_amd/lib/libgvd_diffusion_ctrace.so python tests/scripts/r4_conv_trace.py > gpurun_out/r04_conv_trace.txt 2>> gpurun_out/r04_fourth.err
for c in rule 0 1 2 4; do
  if [ $c = rule ]; then python tests/scripts/r4_tile_sweep.py >> gpurun_out/r04_tile_sweep.txt 2>> gpurun_out/r04_fourth.err
  else GVD_CONV_FORCE_CFG=$c python tests/scripts/r4_tile_sweep.py >> gpurun_out/r04_tile_sweep.txt 2>> gpurun_out/r04_fourth.err; fi
done
python bench.py --workload ddim_guided --ddim-height 320 --ddim-width 448 --steps 6 --warmup 2 --no-cpu-baseline 2>> gpurun_out/r04_fourth.err | cut -c1-200 > gpurun_out/r04_guided_v4.json
tail -3 gpurun_out/r04_fourth_tests.log; cat gpurun_out/r04_conv_ablate_v2.txt gpurun_out/r04_conv_trace.txt gpurun_out/r04_tile_sweep.txt gpurun_out/r04_guided_v4.json
