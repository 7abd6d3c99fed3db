cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_ops_gpu.py -m gpu -x -q 2>&1 | tail -4
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2d -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2d_bench.json 2> gpurun_out/r2d_bench.err
cut -c1-600 gpurun_out/r2d_bench.json
python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | cut -c1-200
python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['h2d_inclusive'], d['roofline']['achieved'])"
