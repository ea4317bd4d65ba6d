# second-session checkpoint: default bench line (raster + ddim with the sustained-clock object) and the guided 320x448 line
mkdir -p gpurun_out
python bench.py > gpurun_out/r04b_bench_default.json 2> gpurun_out/r04b_bench.err
python bench.py --workload ddim_guided --ddim-height 320 --ddim-width 448 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r04b_bench_guided_320x448.json 2>> gpurun_out/r04b_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04b_bench_default.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d.get("sustained"), d["roofline"]["frac"])
dd = d["ddim"]
print({k: dd[k] for k in ("value", "ms_per_step", "sustained_clock")})
for k in ("roofline_conv", "roofline_attention", "roofline_gemm"):
    print(k, dd[k]["achieved"], dd[k]["frac"], dd[k]["ms_per_step"])
g = json.loads(open("gpurun_out/r04b_bench_guided_320x448.json").read().strip().splitlines()[-1])
print({k: g[k] for k in ("value", "ms_per_step", "sustained_clock")})
PY
tail -3 gpurun_out/r04b_bench.err
