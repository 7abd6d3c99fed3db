cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05a
python scripts/lstm_ab.py "" "T=576" > gpurun_out/r05a/lstm_ab.txt 2>&1
for c in c1 c4 c5; do timeout 600 python bench.py --config $c --no-cpu-baseline > gpurun_out/r05a/bench_$c.json 2> gpurun_out/r05a/bench_$c.err; echo "$c rc $?"; done
cat gpurun_out/r05a/lstm_ab.txt
python - <<'PY'
import json
for c in ('c1','c4','c5'):
    d=json.load(open('gpurun_out/r05a/bench_%s.json'%c)); p=d.get('parity',{})
    print(c, d['value'], d['ms_per_step'], p.get('label_mismatches'), p.get('loss_rel_err'), json.dumps(d['ms_per_step_by_entry_point']))
PY
