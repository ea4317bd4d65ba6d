mkdir -p gpurun_out
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids\|Gloo\]'
python -m pytest tests/test_diffusion_gpu.py tests/test_diffusion_goldens_gpu.py tests/test_diffusion_parity_bars_gpu.py tests/test_diffusion_trajectory_gpu.py tests/test_wide_attention_gpu.py -m gpu -q -x 2>&1 | grep -v "$F" | tail -3 > gpurun_out/r04_attn_tests.log
tail -2 gpurun_out/r04_attn_tests.log
for r in 1 2; do
for t in new base; do
  L=$PWD/guidedvd-3dgs_amd/lib/libgvd_diffusion.so; [ $t = base ] && L=$PWD/guidedvd-3dgs_amd/lib/libgvd_diffusion_base.so
  GVD_DIFFUSION_LIB=$L python bench.py --workload ddim --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t ddim', d['ms_per_step'], d['roofline_attention']['achieved'], d['roofline_attention']['ms_per_step'])"
  GVD_DIFFUSION_LIB=$L python bench.py --workload ddim_guided --ddim-height 320 --ddim-width 448 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t guided320', d['ms_per_step'])"
done; done
