# Round 4: evaluation of a GEMM-kernel change -- tests, the micro-benchmark under both K pipelines (GVD_GEMM_STAGES), both bench lines
mkdir -p gpurun_out
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids\|Gloo\]'
python -m pytest tests/test_gemm_gpu.py tests/test_wide_attention_gpu.py tests/test_diffusion_goldens_gpu.py tests/test_diffusion_parity_bars_gpu.py -m gpu -q -x 2>&1 | grep -v "$F" | tail -5 > gpurun_out/r04_gemm_tests.log
python tests/bench_gemm.py 2>/dev/null | cut -c1-120 > gpurun_out/r04_gemm_microbench_st4.txt
GVD_GEMM_STAGES=2 python tests/bench_gemm.py 2>/dev/null | cut -c1-120 > gpurun_out/r04_gemm_microbench_st2.txt
for st in 4 2; do
  GVD_GEMM_STAGES=$st python bench.py --workload ddim --steps 10 --warmup 2 --no-cpu-baseline 2>> gpurun_out/r04_gemm.err | cut -c1-200 > gpurun_out/r04_ddim_gemm_st$st.json
  GVD_GEMM_STAGES=$st python bench.py --workload ddim_guided --ddim-height 320 --ddim-width 448 --steps 6 --warmup 2 --no-cpu-baseline 2>> gpurun_out/r04_gemm.err | cut -c1-200 > gpurun_out/r04_guided_gemm_st$st.json
done
tail -3 gpurun_out/r04_gemm_tests.log; paste -d'|' gpurun_out/r04_gemm_microbench_st4.txt gpurun_out/r04_gemm_microbench_st2.txt | cut -c1-60,120-170; cat gpurun_out/r04_ddim_gemm_st*.json gpurun_out/r04_guided_gemm_st*.json
