cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_round2_gpu.py -m gpu -x -q -k "torch_optim" 2>&1 | tail -40
