# Memory-safety pass (verdict r5 item 6): the randomised raster / diffusion scripts under the red-zone guard allocator (tests/guard/), then the
# address-sanitizer attempt (gfx950:xnack+ builds of the libraries; what the toolchain / the box says is logged verbatim).  -> gpurun_out/r06_guard.log
mkdir -p gpurun_out
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids\|Gloo\]'
O=gpurun_out/r06_guard.log
: > $O
bash tests/guard/build.sh >> $O 2>&1
run() { echo "== $*" >> $O; timeout 1500 python tests/scripts/r6_guard_run.py "$@" 2>&1 | grep -v "$F" | tail -${TAILN:-6} >> $O; echo "exit ${PIPESTATUS[0]}" >> $O; }
run --selftest
run tests/scripts/r5_raster_stress.py
run tests/scripts/r5_raster_stress.py 203
run tests/scripts/r5_raster_threads.py
run tests/scripts/r5_diffusion_fuzz.py 11
run tests/scripts/r5_diffusion_fuzz.py 23
run tests/scripts/r5_unet_shape_fuzz.py 3 16
run tests/scripts/r5_vae_shape_fuzz.py 5 12
echo "== pytest tests/test_raster_gpu.py tests/test_raster_fuzz_gpu.py tests/test_gemm_gpu.py tests/test_conv_gpu.py under the guard (conftest: GVD_GUARD_ALLOC=1)" >> $O
GVD_GUARD_ALLOC=1 timeout 2400 python -m pytest tests/test_raster_gpu.py tests/test_raster_fuzz_gpu.py tests/test_gemm_gpu.py tests/test_conv_gpu.py -m gpu -q -x 2>&1 | grep -v "$F" | tail -8 >> $O
bash tests/scripts/r6_asan.sh >> $O 2>&1
cat $O
