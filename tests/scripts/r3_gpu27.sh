python -m pytest tests/test_gemm_gpu.py tests/test_diffusion_parity_bars_gpu.py tests/test_diffusion_goldens_gpu.py -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids" | tail -8 > gpurun_out/r3_t27.log
python tests/bench_gemm.py > gpurun_out/r3_gemm_bench27.txt 2>&1
python bench.py --workload ddim --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/r3_ddim_c.json 2> gpurun_out/r3_c.err
