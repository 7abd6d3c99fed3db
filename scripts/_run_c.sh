cd $GRAFT_REPO_ROOT
VOCR_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29531 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('default env, nccl world 1:', j['value'], j['ms_per_step'], 'gemm', j['ms_per_step_by_entry_point']['vocr_gemm'])"
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('no dist:', j['value'], j['ms_per_step'])"
python -m pytest tests/test_round2_gpu.py -q -m gpu -k "rccl or allreduce or bench_launches or fit" 2>&1 | grep -E "passed|failed|Error" | tail -3
