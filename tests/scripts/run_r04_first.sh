# Round 4, first GPU call: the new tests first (fast feedback), then the whole GPU suite, the raster evidence at HEAD
# (kernel stats + three PMC passes), and the guided 320x448 step (bench line + kernel summary) as the round's baseline.
mkdir -p gpurun_out
python -m pytest tests/test_diffusion_trajectory_gpu.py tests/test_wide_attention_gpu.py "tests/test_guided_schedule.py::test_config5_eight_ranks_on_one_gpu_with_hip_kernels" -m gpu -q -s -x 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids" | tail -60 > gpurun_out/r04_new_tests.log
python -m pytest tests/ -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids" | tail -15 > gpurun_out/r04_all_tests.log
bash tests/scripts/run_raster_prof_all.sh r04 > gpurun_out/r04_raster_prof.log 2>&1
python bench.py --workload ddim_guided --ddim-height 320 --ddim-width 448 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r04_bench_guided_320x448_base.json 2> gpurun_out/r04_first.err
TAG=r04_guided_base STEPS=5 WARMUP=2 bash tests/scripts/run_ddim_prof.sh --workload ddim_guided --ddim-height 320 --ddim-width 448 --no-cpu-baseline > gpurun_out/prof_guided_base.log 2>&1
F=$(ls gpurun_out/prof_r04_guided_base/*/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$F" ] && python tests/scripts/prof_summary.py $F 70 > gpurun_out/r04_guided_320x448_base_summary.txt
S=$(ls gpurun_out/prof_r04_guided_base/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$S" ] && cp $S gpurun_out/r04_guided_320x448_base_kernel_stats.csv
rm -rf gpurun_out/prof_r04_guided_base
tail -3 gpurun_out/r04_new_tests.log; tail -3 gpurun_out/r04_all_tests.log; cut -c1-400 gpurun_out/r04_bench_guided_320x448_base.json
