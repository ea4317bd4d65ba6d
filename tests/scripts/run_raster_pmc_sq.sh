R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmc_sq -- python $R/tests/profile_raster.py 12 > /dev/null 2>&1
ls $R/gpurun_out/pmc_sq/*/
