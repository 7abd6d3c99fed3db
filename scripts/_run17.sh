cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
VOCR_CONV_DMA=2 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv" 2>&1 | tail -2
echo "--- pc"; VOCR_CONV_DMA=2 SWEEP=0 python scripts/conv_bench.py 2>&1 | grep -v amdgpu
VOCR_CONV_DMA=2 python scripts/conv_occ.py 2>&1 | grep -v amdgpu | tail -8
echo "--- dma"; SWEEP=0 python scripts/conv_bench.py 2>&1 | grep -v amdgpu | tail -3
