# DDIM bench with the MIOpen find-db kept under gpurun_out/miopen_db (merged back, reused on the next call)
export MIOPEN_USER_DB_PATH=$PWD/guidedvd-3dgs_amd/lvdm_amd/miopen_db
mkdir -p $MIOPEN_USER_DB_PATH gpurun_out
export GVD_CONV_FIND=1
timeout 1500 python bench.py --workload ddim --steps ${STEPS:-3} --warmup ${WARMUP:-2} "$@" 2> gpurun_out/ddim_last.err | tail -1 | tee gpurun_out/ddim_last.json
mkdir -p gpurun_out/miopen_db; cp $MIOPEN_USER_DB_PATH/*.txt gpurun_out/miopen_db/ 2>/dev/null
