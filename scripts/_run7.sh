cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "pool or bn" 2>&1 | tail -4
python scripts/bn_bench.py 2>&1 | grep -v amdgpu
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2e -- python3 scripts/bn_bench.py > /dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r2e/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r['Name'][:70], r['Calls'], r['AverageNs'])
PY
