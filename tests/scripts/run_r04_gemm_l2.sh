# sweep of the GEMM's L2 budget for the weight rows of a channel-tile group (GVD_GEMM_L2_BYTES; default 3 << 19); BUDGETS overrides the list
mkdir -p gpurun_out
BUDGETS=${BUDGETS:-1572864 2097152 2621440 3145728 1048576}
for r in 1 2; do for b in $BUDGETS; do
  GVD_GEMM_L2_BYTES=$b python tests/bench_gemm.py 2>/dev/null | cut -c1-60 > gpurun_out/r04_gemm_l2_${b}_$r.txt
  GVD_GEMM_L2_BYTES=$b python bench.py --workload ddim --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$b ddim', d['ms_per_step'], d['roofline_gemm']['achieved'], d['roofline_gemm']['ms_per_step'])"
done; done
echo $BUDGETS "(pass 1) |" $BUDGETS "(pass 2)"
F=""; for r in 1 2; do for b in $BUDGETS; do F="$F gpurun_out/r04_gemm_l2_${b}_$r.txt"; done; done
paste -d'|' $F | awk -F'|' '{ printf "%s", substr($1, 1, 32); for (i = 1; i <= NF; i++) printf " %s", substr($i, 36, 7); printf "\n" }'
