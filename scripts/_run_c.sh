cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --hidden 256 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('H=256:', j['value'], j['ms_per_step'])"
python bench.py --no-cpu-baseline --conv-dtype fp16 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp16 conv:', j['value'], j['ms_per_step'])"
python scripts/decode_bench.py 2>&1 | tail -3
python scripts/sync_step_bench.py 2>&1 | tail -3
