# Round 4: evaluation of a rasterizer change -- raster tests, then the raster bench line (kernels_us) as shipped and with the forward's
# LDS padded back to two workgroups per CU (GVD_FWD_LDS_PAD) for the occupancy A/B
mkdir -p gpurun_out
python -m pytest tests/test_raster_gpu.py tests/test_raster_fuzz_gpu.py tests/test_capi_host.py tests/test_oracle_golden.py -m gpu -q -x 2>&1 | tail -4 > gpurun_out/r04_raster_tests.log
for pad in 0 32768; do
  GVD_FWD_LDS_PAD=$pad python bench.py --workload raster --steps 400 --warmup 50 --no-cpu-baseline 2>> gpurun_out/r04_raster.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('pad $pad', d['value'], d['sustained']['value'], d['kernels_us'])" >> gpurun_out/r04_raster_ab.txt
done
tail -3 gpurun_out/r04_raster_tests.log; cat gpurun_out/r04_raster_ab.txt
