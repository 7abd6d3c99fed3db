cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for m in 128 64; do echo MINCO=$m; VOCR_CONV_WINO4_MINCO=$m SWEEP=0 python scripts/conv_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-80; done
run() { python bench.py --no-cpu-baseline --no-gemm-alone --event-every 0 --steps 60 --warmup 15 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1  %.3f ms  %.1f img/s  loss %s' % (d['ms_per_step'], d['value'], d['config'].get('final_loss')))"; }
run minco128
export VOCR_CONV_WINO4_MINCO=64; run minco64; unset VOCR_CONV_WINO4_MINCO
run minco128
export VOCR_CONV_WINO4_MINCO=64; run minco64; unset VOCR_CONV_WINO4_MINCO
