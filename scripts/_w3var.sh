cd $GRAFT_REPO_ROOT
for c in 0 1 3 7; do
  hipcc --offload-arch=gfx950 -O3 -w -std=c++17 -DW3_CUT=$c -o /tmp/w3v_$c scripts/wgrad3_var.hip || echo FAIL
done
for c in 0 1 3 7 0; do /tmp/w3v_$c; done
/tmp/w3v_0 64 64 30 600; /tmp/w3v_0 128 128 15 420
python scripts/wgrad_ab.py 2
