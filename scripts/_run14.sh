cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv" 2>&1 | tail -3
echo "--- dma"; SWEEP=0 python scripts/conv_bench.py 2>&1 | grep -v amdgpu
python scripts/conv_occ.py 2>&1 | grep -v amdgpu | tail -8
echo "--- old"; VOCR_CONV_DMA=0 SWEEP=0 python scripts/conv_bench.py 2>&1 | grep -v amdgpu
