cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-r3e}
mkdir -p gpurun_out/$T
python bench.py --no-cpu-baseline > gpurun_out/$T/bench_default.json 2> gpurun_out/$T/bench_default.err
env ${2:-VOCR_DW_TILES=1} python bench.py --no-cpu-baseline > gpurun_out/$T/bench_b.json 2> /dev/null
python bench.py --no-cpu-baseline > gpurun_out/$T/bench_default2.json 2> /dev/null
env ${2:-VOCR_DW_TILES=1} python bench.py --no-cpu-baseline > gpurun_out/$T/bench_b2.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/$T/bench_profiled.json 2> gpurun_out/$T/bench_profiled.err
for f in default b default2 b2; do python - <<PY
import json
d=json.load(open("gpurun_out/$T/bench_$f.json"))
print("$f", d["value"], d["ms_per_step"], d["resident_input"]["ms_per_step"], {k: v for k, v in d["ms_per_step_by_entry_point"].items() if "pool" in k or "bn_relu_bwd" in k or "gemm" in k or "lstm" in k})
PY
done
