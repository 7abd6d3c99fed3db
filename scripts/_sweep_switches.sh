# same-box A/B of experiment switches against the base, on the experiments build (python -m vistaocr_amd.build --experiments):
#   bash scripts/_sweep_switches.sh "VOCR_CONV_WINO4=5 VOCR_SIDE_LOWPRIO=0" [reps]
cd $GRAFT_REPO_ROOT
run() { python scripts/_lib_ab.py cut ../bench.py --no-cpu-baseline --no-gemm-alone --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
export VOCR_EXPERIMENTS=1
for rep in $(seq 1 ${2:-1}); do
  for kv in $1; do
    echo "base: $(run)"
    echo "$kv: $(env $kv bash -c "$(declare -f run); run")"
  done
done
echo "base: $(run)"
