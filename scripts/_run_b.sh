cd $GRAFT_REPO_ROOT
for a in 1 2 ; do
  for t in 1 0; do
  echo "== tail=$t"
  date +%s.%N
  VOCR_CONV_TAIL=$t timeout 200 python bench.py --no-cpu-baseline --steps 100 --warmup 20 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], 'h2d', j['h2d_inclusive']['ms_per_step'], 'conv', j['roofline']['achieved'], j['roofline']['frac'])"
  done
done
date +%s.%N
