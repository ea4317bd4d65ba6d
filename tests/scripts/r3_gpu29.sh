export GVD_GEMM_BLOCKS=4
echo "== 8-wave, 256-wide tiles" > gpurun_out/r3_gemm_w4.txt
GVD_GEMM_VARIANT=1 python tests/bench_gemm.py 2>/dev/null | head -16 >> gpurun_out/r3_gemm_w4.txt
for lib in libgvd_diffusion libgvd_diffusion_w4ku2 libgvd_diffusion_w4ku4; do
echo "== 4 waves x (128 x 128), $lib" >> gpurun_out/r3_gemm_w4.txt
GVD_DIFFUSION_LIB=guidedvd-3dgs_amd/lib/$lib.so GVD_GEMM_VARIANT=2 python tests/bench_gemm.py 2>/dev/null | head -16 >> gpurun_out/r3_gemm_w4.txt
done
GVD_GEMM_VARIANT=2 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x 2>&1 | tail -3 >> gpurun_out/r3_gemm_w4.txt
