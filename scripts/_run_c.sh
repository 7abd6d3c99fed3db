cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py -q -m gpu -k "conv" 2>&1 | grep -E "passed|failed|Error|mismatch" | tail -8
SWEEP=0 python scripts/conv_bench.py
