# the -m gpu suite several times over (catches order- / timing-dependent failures: the persistent GEMM's staging race showed up once in ~3 runs)
for i in 1 2 3; do
python -m pytest tests/ -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids" | tail -3 | sed "s/^/run $i: /" >> gpurun_out/soak.log
done
