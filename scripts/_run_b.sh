cd $GRAFT_REPO_ROOT
for a in "" "--warmup 60" "--event-every 0" "--event-every 1" "--steps 200 --warmup 20" ""; do
  echo "== $a"
  python bench.py --no-cpu-baseline $a 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], 'h2d', j['h2d_inclusive']['ms_per_step'], 'conv', j['roofline']['achieved'])"
done
