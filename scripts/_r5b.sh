cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05b
timeout 900 python -m pytest tests/test_packed_gpu.py tests/test_abi.py -x -q > gpurun_out/r05b/t_packed.txt 2>&1; echo "packed rc $?"; tail -15 gpurun_out/r05b/t_packed.txt
timeout 1200 python -m pytest tests/test_ops_gpu.py -x -q -k "lstm or sweep or bilstm" > gpurun_out/r05b/t_lstm.txt 2>&1; echo "lstm rc $?"; tail -5 gpurun_out/r05b/t_lstm.txt
python scripts/lstm_ab.py "" "T=576" > gpurun_out/r05b/lstm_ab.txt 2>&1; cat gpurun_out/r05b/lstm_ab.txt
for c in c1 c4; do timeout 600 python bench.py --config $c --no-cpu-baseline > gpurun_out/r05b/bench_$c.json 2> gpurun_out/r05b/bench_$c.err; echo "$c rc $?"; tail -3 gpurun_out/r05b/bench_$c.err; done
python - <<'PY'
import json
for c in ('c1','c4'):
    try:
        d=json.load(open('gpurun_out/r05b/bench_%s.json'%c)); p=d.get('parity',{})
        print(c, d['value'], d['ms_per_step'], p.get('label_mismatches'), p.get('loss_rel_err'), json.dumps(d['ms_per_step_by_entry_point']))
    except Exception as e: print(c, 'ERR', e)
PY
