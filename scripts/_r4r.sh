cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r04h
for c in c4 c5; do timeout 600 python bench.py --config $c > gpurun_out/r04h/bench_$c.json 2> gpurun_out/r04h/bench_$c.err; echo "$c rc $?"; done
python - <<'PY'
import json
for c in ('c4','c5'):
    d=json.load(open('gpurun_out/r04h/bench_%s.json'%c)); p=d.get('parity',{})
    print(c, d['value'], d['ms_per_step'], p.get('label_mismatches'), p.get('loss_rel_err'), p.get('cer'), d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
