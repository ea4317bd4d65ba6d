# (Re)record hipBLASLt/rocBLAS solution choices for the U-Net GEMM shapes with PyTorch TunableOp; results accumulate in
# lvdm_amd/tunableop/tunableop0.csv (copied to gpurun_out/ so they come back from the GPU box).  args -> bench.py
export MIOPEN_USER_DB_PATH=$PWD/guidedvd-3dgs_amd/lvdm_amd/miopen_db
export GVD_CONV_FIND=1
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=$PWD/guidedvd-3dgs_amd/lvdm_amd/tunableop/tunableop.csv
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=20 PYTORCH_TUNABLEOP_MAX_TUNING_ITERATIONS=5 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5
mkdir -p gpurun_out
timeout 1700 python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" 2> gpurun_out/tunable.err | tail -1 | cut -c1-200
cp guidedvd-3dgs_amd/lvdm_amd/tunableop/tunableop0.csv gpurun_out/tunableop0.csv; wc -l gpurun_out/tunableop0.csv
cp $MIOPEN_USER_DB_PATH/*.txt gpurun_out/miopen_db/ 2>/dev/null
