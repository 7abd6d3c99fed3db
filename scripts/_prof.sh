cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-r02b}
mkdir -p gpurun_out/$T
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/$T/bench_profiled.json 2> gpurun_out/$T/bench_profiled.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/$T/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/$T/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/$T/sq -- python3 scripts/one_conv.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/$T/sq2 -- python3 scripts/one_conv.py > /dev/null 2>&1
python bench.py > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$T/smoke.txt 2>&1; tail -2 gpurun_out/$T/smoke.txt
tail -c 1200 gpurun_out/$T/bench.json
