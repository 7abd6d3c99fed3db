cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { python bench.py --no-cpu-baseline --no-gemm-alone --event-every 0 --steps 60 --warmup 15 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1  %.3f ms  %.1f img/s  loss %s' % (d['ms_per_step'], d['value'], d['config'].get('final_loss')))"; }
run fast_bwd_tanh; run fast_bwd_tanh; run fast_bwd_tanh
timeout 1500 python -m pytest tests/test_configs_gpu.py tests/test_round2_gpu.py -x -q 2>&1 | tail -2
