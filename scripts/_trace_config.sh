cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; C=${1:-c4}; T=${2:-r05c}_$C; mkdir -p gpurun_out/$T
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/trace -- python3 bench.py --config $C --steps 8 --warmup 3 --no-cpu-baseline --no-gemm-alone > gpurun_out/$T/bench_profiled.json 2> gpurun_out/$T/bench_profiled.err
python scripts/timeline.py $(find gpurun_out/$T/trace -name "*kernel_trace.csv" | head -1) 5 > gpurun_out/$T/timeline.txt 2>&1
cp $(find gpurun_out/$T/trace -name "*kernel_stats.csv" | head -1) gpurun_out/$T/kernel_stats.csv
rm -rf gpurun_out/$T/trace
timeout 900 python bench.py --config $C > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err; echo "rc $?"
python - <<PY
import json
d=json.load(open('gpurun_out/$T/bench.json'))
print(d['value'], d['ms_per_step'], json.dumps(d['ms_per_step_by_entry_point']))
for k,v in d['roofline']['families_in_step'].items(): print(k[:40], v['ms_per_step'], v['frac'], v['executed_frac'])
PY
head -25 gpurun_out/$T/kernel_stats.csv | cut -c1-150
