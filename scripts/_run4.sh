cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
VOCR_CONV_TILE=3 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv3x3_fwd" 2>&1 | tail -3
SWEEP=0 python scripts/conv_bench.py 2>&1 | grep -v amdgpu.ids
echo "--- tiny"
VOCR_CONV_TILE=3 SWEEP=0 python scripts/conv_bench.py 2>&1 | grep -v amdgpu.ids
