# guided 320x448 step, alternating runs on one box: an environment switch on / off.  usage: r6_guided_ab.sh VAR [tag]
mkdir -p gpurun_out
O=gpurun_out/r06_guided_ab_${2:-$1}.txt
: > $O
for pass in 1 2; do
  for v in off on; do
    if [ $v = on ]; then export $1=1; else unset $1; fi
    python bench.py --workload ddim_guided --ddim-height 320 --ddim-width 448 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', '$v', 'ms_per_step', d['ms_per_step'], 'clock', (d.get('sustained_clock') or {}).get('sclk_mhz_mean'))" >> $O
  done
done
cat $O
