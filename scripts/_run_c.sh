cd $GRAFT_REPO_ROOT
python scripts/gemm_bench.py 2>&1 | tail -11
python -m pytest tests/test_ops_gpu.py -q -m gpu -k "gemm" 2>&1 | grep -E "passed|failed|Error" | tail -3
for i in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], 'h2d', j['h2d_inclusive']['ms_per_step'], 'conv', j['roofline']['achieved'], j['roofline']['frac'])"; done
