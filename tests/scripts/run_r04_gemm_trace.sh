mkdir -p gpurun_out
for l in libgvd_diffusion_trace.so libgvd_diffusion_trace_nostore.so; do python tests/scripts/r4_gemm_trace.py $l 2>/dev/null; done > gpurun_out/r04_gemm_trace2.txt
cat gpurun_out/r04_gemm_trace2.txt
