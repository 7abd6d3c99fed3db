cd $GRAFT_REPO_ROOT
SWEEP_FIRST=1 python scripts/sweep_corun.py 2>&1 | tail -16
