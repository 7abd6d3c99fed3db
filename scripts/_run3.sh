cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2c
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/r2c/sq -- python3 scripts/one_conv.py > gpurun_out/r2c/sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d gpurun_out/r2c/sq2 -- python3 scripts/one_conv.py > gpurun_out/r2c/sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r2c/fetch -- python3 scripts/one_conv.py > gpurun_out/r2c/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r2c/write -- python3 scripts/one_conv.py > gpurun_out/r2c/write.log 2>&1
find gpurun_out/r2c -name "*.csv" | head -20
