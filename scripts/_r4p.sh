cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python scripts/wgrad_ab.py 3 2>&1 | grep -v amdgpu.ids
for m in 3 2; do echo mode $m; VOCR_WGRAD_WINO_DMA=$m python scripts/_w2d_probe.py 2>&1 | grep -v amdgpu.ids; done
