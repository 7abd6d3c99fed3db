cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py -q -m gpu -k "conv3x3_fwd_dgrad" 2>&1 | grep -E "passed|failed|Error|mismatch" | tail -5
