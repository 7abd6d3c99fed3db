cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py -q -m gpu -k "conv" 2>&1 | grep -E "passed|failed|Error" | tail -3
SWEEP=0 python scripts/conv_bench.py
echo "--- separate tail launch"; VOCR_CONV_TAIL=3 SWEEP=0 python scripts/conv_bench.py | tail -6
