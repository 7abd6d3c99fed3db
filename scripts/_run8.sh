cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2f -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r2f_bench.json 2> gpurun_out/r2f_bench.err
for f in 1 0; do VOCR_POOL_BWD_FUSED=$f python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused=$f', d['value'], d['ms_per_step'], d['h2d_inclusive']['ms_per_step'], d['roofline']['achieved'])"; done
