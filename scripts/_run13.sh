cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv" 2>&1 | tail -2
SWEEP=0 python scripts/conv_bench.py 2>&1 | grep -v amdgpu
python scripts/conv_occ.py 2>&1 | grep -v amdgpu
run() { python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['h2d_inclusive']['ms_per_step'], d['roofline']['achieved'], d['ms_per_step_by_entry_point'])"; }
run base
run base
