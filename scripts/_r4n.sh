cd $GRAFT_REPO_ROOT
python scripts/sweep_corun.py 2>&1 | tail -30
