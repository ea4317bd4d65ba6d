# Round 4: A/B of a host-side switch on a bench line.  usage: [WL="ddim_guided --ddim-height 320 --ddim-width 448"] bash tests/scripts/run_r04_ab.sh ENV_NAME "pytest node ids"
mkdir -p gpurun_out
V=$1; shift
WL=${WL:-ddim}
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids\|Gloo\]'
[ -n "$1" ] && python -m pytest $@ -m gpu -q -x 2>&1 | grep -v "$F" | tail -4 > gpurun_out/r04_ab_tests.log
rm -f gpurun_out/r04_ab.txt
for v in 1 0 1 0; do
  env $V=$v python bench.py --workload $WL --steps 10 --warmup 2 --no-cpu-baseline 2>> gpurun_out/r04_ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$V=$v $WL', d['ms_per_step'], d.get('roofline_conv', {}).get('ms_per_step'))" >> gpurun_out/r04_ab.txt
done
tail -3 gpurun_out/r04_ab_tests.log; cat gpurun_out/r04_ab.txt
