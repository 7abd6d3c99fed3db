cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python scripts/wgrad_ab.py 3 2>&1 | grep -v amdgpu.ids | tail -7
bash scripts/_pmc_wgrad.sh 3 | tail -22
