cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -s 2>&1 | tail -60 > gpurun_out/r2a_pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err
tail -c 3000 gpurun_out/r2a_bench.json
