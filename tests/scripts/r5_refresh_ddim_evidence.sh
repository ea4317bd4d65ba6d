GVD_BENCH_SHAPE_TABLE=gpurun_out/r05_ddim_by_shape.json python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_evidence.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_flags.json 2>> gpurun_out/r05_evidence.err
tag=r05_ddim_576x1024
TAG=$tag STEPS=3 WARMUP=1 bash tests/scripts/run_ddim_prof.sh --workload ddim --no-cpu-baseline > gpurun_out/prof_ddim.log 2>&1
T=$(ls gpurun_out/prof_$tag/*/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$T" ] && python tests/scripts/prof_summary.py $T 70 > gpurun_out/${tag}_summary.txt
S=$(ls gpurun_out/prof_$tag/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$S" ] && cp $S gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/prof_$tag
head -3 gpurun_out/${tag}_summary.txt; grep "not this package" gpurun_out/${tag}_summary.txt; grep -i "cat\|gn_stats" gpurun_out/${tag}_summary.txt | head -8
python - <<PY
import json
for f in ("r05_bench_default","r05_bench_driver_flags"):
    d=json.loads(open(f"gpurun_out/{f}.json").readline()); print(f, d["value"], d["ms_per_step"], d["sustained"]["value"], "ddim", d["ddim"]["value"], d["ddim"]["ms_per_step"])
PY
