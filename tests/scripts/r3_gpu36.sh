python -m pytest tests/ -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids" | tail -6 > gpurun_out/r3_t36.log
python bench.py --workload ddim_guided --ddim-height 320 --ddim-width 448 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3_guided_320_f.json 2> gpurun_out/r3_f.err
python bench.py --workload ddim_guided --ddim-height 320 --ddim-width 448 --steps 5 --warmup 2 --no-cpu-baseline --no-batch-cfg > gpurun_out/r3_guided_320_f_seq.json 2>> gpurun_out/r3_f.err
python bench.py --workload config4 --no-cpu-baseline > gpurun_out/r3_config4_f.json 2>> gpurun_out/r3_f.err
