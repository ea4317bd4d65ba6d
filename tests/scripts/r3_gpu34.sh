python -m pytest tests/test_raster_gpu.py tests/test_raster_fuzz_gpu.py -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids" | tail -6 > gpurun_out/r3_t34.log
python bench.py --workload raster --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r3_raster_q.json 2> gpurun_out/r3_q.err
