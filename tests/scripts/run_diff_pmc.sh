# rocprofv3 counter pass (SQ + GRBM only; no trace domains) over tests/scripts/diff_kernels_one.py -> gpurun_out/<TAG>_mfma_pmc.{csv,json}
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${TAG:-r02}
OUT=/tmp/pmc_${TAG}
rm -rf $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python $R/tests/scripts/diff_kernels_one.py > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES --output-format csv -d $OUT/b -- python $R/tests/scripts/diff_kernels_one.py > /dev/null 2>&1
# memory-side traffic: FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950 (MI355X_MICROARCH.md, TCC: 3 + 2 of 4 slots)
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c -- python $R/tests/scripts/diff_kernels_one.py > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/d -- python $R/tests/scripts/diff_kernels_one.py > /dev/null 2>&1
python $R/tests/scripts/diff_pmc_summary.py $OUT/a $OUT/b $R/gpurun_out/${TAG}_mfma_pmc $OUT/c $OUT/d
