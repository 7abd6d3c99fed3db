cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python scripts/bias_probe.py 2>&1 | grep -v amdgpu
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_model_golden_gpu.py -x -q -k "lstm or golden or bilstm" 2>&1 | tail -3
