cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
VOCR_CONV_WINO4=5 timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "conv3x3_fwd_dgrad_wgrad and transform" 2>&1 | tail -3
for m in 1 5; do echo "VOCR_CONV_WINO4=$m"; SWEEP=0 VOCR_CONV_WINO4=$m python scripts/conv_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-100; done
