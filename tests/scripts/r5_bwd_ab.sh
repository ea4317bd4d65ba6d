# Round 5: A/B of the backward blend kernel forms (GVD_BWD_VARIANT: 0 = per-pixel DPP reduction (round 4), 1 = transposed 128x16,
# 2 = transposed 256x8, 3 = transposed 128x8).  Parity tests first, then the raster bench line per variant.
mkdir -p gpurun_out
VARIANTS=${VARIANTS:-"0 1 2"}
for v in $VARIANTS; do
  echo "=== variant $v: parity" 
  GVD_BWD_VARIANT=$v timeout 900 python -m pytest tests/test_raster_gpu.py tests/test_raster_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -3
done
for rep in 1 2; do
for v in $VARIANTS; do
  GVD_BWD_VARIANT=$v python bench.py --workload raster --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | tail -1 > gpurun_out/r5_bwd_v$v.json
  python -c "import json; d=json.load(open('gpurun_out/r5_bwd_v$v.json')); print('variant $v', d['value'], d['sustained']['value'], d['kernels_us'])"
done
done
