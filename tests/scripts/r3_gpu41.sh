python -m pytest tests/ -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids" | tail -6 > gpurun_out/r3_t41.log
python bench.py > gpurun_out/r3_bench_final2.json 2> gpurun_out/r3_final2.err
python bench.py --workload ddim_guided --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3_guided_final2.json 2>> gpurun_out/r3_final2.err
python bench.py --workload config4 --no-cpu-baseline > gpurun_out/r3_config4_final2.json 2>> gpurun_out/r3_final2.err
python bench.py --workload pipeline --no-cpu-baseline > gpurun_out/r3_pipeline_final2.json 2>> gpurun_out/r3_final2.err
