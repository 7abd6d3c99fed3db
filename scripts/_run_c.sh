cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5
for i in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], 'h2d', j['h2d_inclusive']['ms_per_step'], 'conv', j['roofline']['achieved'], j['roofline']['frac'])"; done
