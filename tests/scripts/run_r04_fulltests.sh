# the whole GPU suite + the two bench lines (a checkpoint run)
mkdir -p gpurun_out
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids\|Gloo\]'
python -m pytest tests/ -m gpu -q 2>&1 | grep -v "$F" | tail -12 > gpurun_out/r04_checkpoint_tests.log
python bench.py > gpurun_out/r04_checkpoint_bench.json 2> gpurun_out/r04_checkpoint.err
tail -6 gpurun_out/r04_checkpoint_tests.log; cut -c1-400 gpurun_out/r04_checkpoint_bench.json
