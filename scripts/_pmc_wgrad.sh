# PMC passes (MFMA busy, LDS) over the kernels of scripts/one_conv.py with the row-pair weight-gradient kernel
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
export VOCR_WGRAD_WINO_DMA=${1:-3}
T=gpurun_out/pmcw
rm -rf $T; mkdir -p $T
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $T/sq -- python3 scripts/one_conv.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $T/sq2 -- python3 scripts/one_conv.py > /dev/null 2>&1
python scripts/pmc_summary.py busy $(find $T/sq -name "*counter_collection.csv") $(find $T/sq2 -name "*counter_collection.csv") 2>&1 | sed -n '/weight gradient/,/^$/p'
