R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_attn1 -- python $R/tests/scripts/attn_one.py $1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_attn2 -- python $R/tests/scripts/attn_one.py $1 > /dev/null 2>&1
ls $R/gpurun_out/pmc_attn1/*/ $R/gpurun_out/pmc_attn2/*/
