cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv3x3_fwd_dgrad" 2>&1 | tail -5
SWEEP=0 python scripts/conv_bench.py
echo ---- no tail; VOCR_CONV_TAIL=0 SWEEP=0 python scripts/conv_bench.py
