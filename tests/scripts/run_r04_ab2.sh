# Round 4: A/B of TWO host switches together on a bench line: usage [WL=...] bash tests/scripts/run_r04_ab2.sh ENV_A ENV_B
mkdir -p gpurun_out
A=$1; B=$2
WL=${WL:-ddim}
rm -f gpurun_out/r04_ab2.txt
for v in "1 1" "0 0" "1 0" "1 1" "0 0"; do
  set -- $v
  env $A=$1 $B=$2 python bench.py --workload $WL --steps 10 --warmup 2 --no-cpu-baseline 2>> gpurun_out/r04_ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$A=$1 $B=$2 $WL', d['ms_per_step'], d.get('roofline_conv', {}).get('ms_per_step'))" >> gpurun_out/r04_ab2.txt
done
cat gpurun_out/r04_ab2.txt
