cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05f
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "f16" > gpurun_out/r05f/t_f16.txt 2>&1; echo "f16 tests rc $?"; tail -12 gpurun_out/r05f/t_f16.txt
timeout 300 python scripts/f16_conv_bench.py > gpurun_out/r05f/f16_conv_bench.txt 2>&1; cat gpurun_out/r05f/f16_conv_bench.txt
timeout 600 python bench.py --config c5 --no-cpu-baseline > gpurun_out/r05f/bench_c5.json 2> gpurun_out/r05f/bench_c5.err; echo "c5 rc $?"; tail -3 gpurun_out/r05f/bench_c5.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05f/bench_c5.json'))
print('c5', d['value'], d['ms_per_step'], json.dumps(d['ms_per_step_by_entry_point']))
PY
