# The shortest useful call (~8-10 min on the box): the GPU tests added in the round's last session, smoke(), the driver-flag bench line.
mkdir -p gpurun_out
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids\|Gloo\]'
python -m pytest -q -m gpu tests/test_raster_gpu.py -k "compact_chunk or no_grad_renders or sync_free" tests/test_diffusion_gpu.py::test_fp32_device_tensors_raise_unless_the_caller_opts_in tests/test_diffusion_gpu.py::test_multicond_sampler_on_device_uses_fused_step 2>&1 | grep -v "$F" | tail -8 > gpurun_out/r06_new_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" | tail -3 >> gpurun_out/r06_new_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_flags.json 2> gpurun_out/r06_last_minute.err
GVD_GUARD_TESTS=1 python -m pytest -q -m gpu tests/test_guard_allocator_gpu.py 2>&1 | grep -v "$F" | tail -6 >> gpurun_out/r06_new_tests.log
cat gpurun_out/r06_new_tests.log; cut -c1-500 gpurun_out/r06_bench_driver_flags.json; tail -3 gpurun_out/r06_last_minute.err
