for m in 0 1 0 1; do
  echo "fused=$m"
  GVD_RASTER_FUSED_SORT=$m python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['kernels_us'])"
done
