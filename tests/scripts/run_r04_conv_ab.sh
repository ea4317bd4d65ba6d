# A/B of the XCD-aware convolution grid readings (GVD_CONV_XCD_MAP=0: plain, 1: inputs first, 2: weights first, unset: xcd_rule()); convolution tests under the rule
mkdir -p gpurun_out
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids\|Gloo\]'
python -m pytest tests/test_conv_gpu.py -m gpu -q -x 2>&1 | grep -v "$F" | tail -3 > gpurun_out/r04_conv_ab_tests.log
GVD_CONV_XCD_MAP=2 python -m pytest tests/test_conv_gpu.py -m gpu -q -x 2>&1 | grep -v "$F" | tail -1 >> gpurun_out/r04_conv_ab_tests.log
tail -2 gpurun_out/r04_conv_ab_tests.log
TAGS=${TAGS:-plain m1 m2 rule}
for r in 1 2; do for t in $TAGS; do
  case $t in plain) export GVD_CONV_XCD_MAP=0;; m1) export GVD_CONV_XCD_MAP=1;; m2) export GVD_CONV_XCD_MAP=2;; rule) unset GVD_CONV_XCD_MAP;; esac
  python tests/bench_conv.py --no-miopen 2>/dev/null | cut -c1-110 > gpurun_out/r04_conv_ab_${t}_$r.txt
  python bench.py --workload ddim --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t ddim', d['ms_per_step'], d['roofline_conv']['achieved'], d['roofline_conv']['ms_per_step'])"
  python bench.py --workload ddim_guided --ddim-height 320 --ddim-width 448 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t guided320', d['ms_per_step'], d['roofline_conv']['achieved'], d['roofline_conv']['ms_per_step'])"
done; done
echo $TAGS "(pass 1) |" $TAGS "(pass 2)"
F2=""; for r in 1 2; do for t in $TAGS; do F2="$F2 gpurun_out/r04_conv_ab_${t}_$r.txt"; done; done
paste -d'|' $F2 | awk -F'|' '{printf "%s |", substr($1,1,34); for(i=1;i<=NF;i+=2) printf " %s", substr($i,35,12); printf "\n"}'
