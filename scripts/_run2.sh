cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_ops_gpu.py tests/test_round2_gpu.py -m gpu -x -q -k "conv or ctc or bn or config5 or reproducible or fit" 2>&1 | tail -15 > gpurun_out/r2b_pytest.log
python scripts/conv_bench.py > gpurun_out/r2b_conv.log 2>&1
VOCR_CONV_TILE=2 SWEEP=0 python scripts/conv_bench.py > gpurun_out/r2b_conv_full.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2b_bench.json 2> gpurun_out/r2b_bench.err
cat gpurun_out/r2b_pytest.log gpurun_out/r2b_conv.log gpurun_out/r2b_conv_full.log; cut -c1-1800 gpurun_out/r2b_bench.json
