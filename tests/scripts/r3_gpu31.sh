python -m pytest tests/test_conv_gpu.py tests/test_diffusion_goldens_gpu.py -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids" | tail -4 > gpurun_out/r3_t31.log
python tests/bench_conv.py --no-miopen > gpurun_out/r3_conv_bench31.txt 2>&1
sed -i 's/for d in 0 1 2 4 8 12 16 32 48/for d in 0/g' tests/scripts/run_conv_lds_hunt.sh
bash tests/scripts/run_conv_lds_hunt.sh > gpurun_out/r3_hunt2.log 2>&1
cp gpurun_out/r03_conv_lds_hunt.txt gpurun_out/r03_conv_lds_hunt_after.txt
