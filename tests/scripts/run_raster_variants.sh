# Raster bench variants quoted in DESIGN.md / README.md.
python bench.py 2>/dev/null | tail -1 > gpurun_out/final_raster.json
python -c "import json; d=json.load(open('gpurun_out/final_raster.json')); print('default', d['value'], d['roofline'], d['kernels_us'], d['variants'], d['cpu_baseline'])"
for v in "--with-loss fused1" "--with-loss fused" "--instance-capacity 700000" "--instance-capacity 700000 --with-loss fused1"; do
  python bench.py --no-cpu-baseline $v 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"
done
