cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_ops_gpu.py tests/test_model_golden_gpu.py -m gpu -x -q 2>&1 | tail -15
for i in 1 2; do python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['h2d_inclusive'], d['roofline']['achieved'], d['ms_per_step_by_entry_point'])"; done
