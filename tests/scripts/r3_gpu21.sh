hipcc --offload-arch=gfx950 -O3 -o /tmp/pingpong tests/scripts/mfma_valu_pingpong.hip 2>/dev/null
/tmp/pingpong > gpurun_out/r03_mfma_valu_pingpong.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -DNV=64 -DNE=16 -o /tmp/pingpong2 tests/scripts/mfma_valu_pingpong.hip 2>/dev/null
/tmp/pingpong2 >> gpurun_out/r03_mfma_valu_pingpong.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_valu tests/scripts/mfma_valu_overlap.hip 2>/dev/null
/tmp/mfma_valu >> gpurun_out/r03_mfma_valu_pingpong.txt 2>&1
python tests/bench_conv.py --temporal-only > gpurun_out/r3_conv_temporal.txt 2>&1
