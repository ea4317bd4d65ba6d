python -m pytest tests/ -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids" | tail -8 > gpurun_out/r3_t25.log
python bench.py --workload ddim --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/r3_ddim_b.json 2> gpurun_out/r3_b.err
python bench.py --workload ddim_guided --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3_guided_576_b.json 2>> gpurun_out/r3_b.err
