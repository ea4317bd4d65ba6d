# A/B of GEMM-kernel builds: lib/libgvd_diffusion_<tag>.so for each tag in $TAGS (default "base exp3"), two alternating passes of tests/bench_gemm.py
mkdir -p gpurun_out
TAGS=${TAGS:-base exp3}
for r in 1 2; do for t in $TAGS; do
GVD_DIFFUSION_LIB=$PWD/guidedvd-3dgs_amd/lib/libgvd_diffusion_$t.so python tests/bench_gemm.py 2>/dev/null | cut -c1-60 > gpurun_out/r04_gemm_ab_${t}_$r.txt
done; done
for t in $TAGS; do printf "%-16s" $t; done; echo
F=""; for r in 1 2; do for t in $TAGS; do F="$F gpurun_out/r04_gemm_ab_${t}_$r.txt"; done; done
paste -d'|' $F | awk -F'|' '{ printf "%s", substr($1, 1, 32); for (i = 1; i <= NF; i++) printf " %s", substr($i, 33, 16); printf "\n" }'
