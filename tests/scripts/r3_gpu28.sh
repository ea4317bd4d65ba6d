echo "== default (8-wave for big shapes)" > gpurun_out/r3_gemm_variant.txt
python tests/bench_gemm.py 2>/dev/null | head -16 >> gpurun_out/r3_gemm_variant.txt
echo "== GVD_GEMM_VARIANT=0 (4 waves, two workgroups per CU)" >> gpurun_out/r3_gemm_variant.txt
GVD_GEMM_VARIANT=0 python tests/bench_gemm.py 2>/dev/null | head -16 >> gpurun_out/r3_gemm_variant.txt
