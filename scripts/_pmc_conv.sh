# gpurun: PMC passes over scripts/one_conv.py (conv forward 256->256, its weight gradient, the panel GEMMs) -> gpurun_out/$1/mfma_busy.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=${1:-r4b}; mkdir -p gpurun_out/$T
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/$T/sq -- python3 scripts/one_conv.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/$T/sq2 -- python3 scripts/one_conv.py > /dev/null 2>&1
python scripts/pmc_summary.py busy $(find gpurun_out/$T/sq -name "*counter_collection.csv") $(find gpurun_out/$T/sq2 -name "*counter_collection.csv") > gpurun_out/$T/mfma_busy.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/trace -- python3 scripts/one_conv.py > /dev/null 2>&1
cp $(find gpurun_out/$T/trace -name "*kernel_stats.csv" | head -1) gpurun_out/$T/one_conv_kernel_stats.csv
rm -rf gpurun_out/$T/sq gpurun_out/$T/sq2 gpurun_out/$T/trace
head -60 gpurun_out/$T/mfma_busy.txt; head -12 gpurun_out/$T/one_conv_kernel_stats.csv
