cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['h2d_inclusive']['ms_per_step'], d['ms_per_step_by_entry_point']['vocr_conv3x3_wgrad'])"; }
run base
VOCR_SIDE_LOWPRIO=0 run side_normal_prio
VOCR_WGRAD_OCC2=1 run wgrad_occ2
VOCR_WGRAD_OCC2=1 VOCR_SIDE_LOWPRIO=0 run occ2_normal_prio
run base
VOCR_WGRAD_OCC2=1 SWEEP=0 python scripts/conv_bench.py 2>&1 | grep "wgrad" | cut -c1-120
