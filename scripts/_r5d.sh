cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05d
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "ctc" > gpurun_out/r05d/t_ctc.txt 2>&1; echo "ctc rc $?"; tail -5 gpurun_out/r05d/t_ctc.txt
bash scripts/_r5c.sh c5
timeout 600 python bench.py --config c4 --no-cpu-baseline > gpurun_out/r05d/bench_c4.json 2> gpurun_out/r05d/bench_c4.err; echo "c4 rc $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05d/bench_c4.json'))
print('c4', d['value'], d['ms_per_step'], json.dumps(d['ms_per_step_by_entry_point']))
PY
