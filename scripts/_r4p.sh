cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "conv3x3_fwd_dgrad_wgrad and transform" 2>&1 | tail -3
SWEEP=0 python scripts/conv_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-110
