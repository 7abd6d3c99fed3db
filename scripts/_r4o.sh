cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
VOCR_WGRAD_WINO_DMA=3 timeout 1200 python -m pytest tests/test_ops_gpu.py -x -q -k "conv or wgrad" 2>&1 | tail -3
run() { python bench.py --no-cpu-baseline --no-gemm-alone --event-every 0 --steps 60 --warmup 15 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1  %.3f ms  %.1f img/s  loss %s' % (d['ms_per_step'], d['value'], d['config'].get('final_loss')))"; }
run mode2
export VOCR_WGRAD_WINO_DMA=3; run mode3; unset VOCR_WGRAD_WINO_DMA
run mode2
export VOCR_WGRAD_WINO_DMA=3; run mode3; unset VOCR_WGRAD_WINO_DMA
