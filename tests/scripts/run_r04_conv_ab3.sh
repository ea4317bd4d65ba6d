for r in 1 2 3; do for t in rule r3 m1; do
  case $t in m1) export GVD_CONV_XCD_MAP=1;; r3) export GVD_CONV_XCD_MAP=3;; rule) unset GVD_CONV_XCD_MAP;; esac
  python bench.py --workload ddim_guided --ddim-height 320 --ddim-width 448 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t guided320', d['ms_per_step'], d['roofline_conv']['ms_per_step'])"
done; done
for t in rule r3; do
  case $t in r3) export GVD_CONV_XCD_MAP=3;; rule) unset GVD_CONV_XCD_MAP;; esac
  python bench.py --workload ddim --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t ddim', d['ms_per_step'], d['roofline_conv']['ms_per_step'])"
done
