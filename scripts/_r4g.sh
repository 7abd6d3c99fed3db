cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4g
timeout 900 python bench.py --config c4 > gpurun_out/r4g/bench_c4.json 2> gpurun_out/r4g/bench_c4.err; echo "c4 rc $?"
timeout 900 python bench.py --config c5 > gpurun_out/r4g/bench_c5.json 2> gpurun_out/r4g/bench_c5.err; echo "c5 rc $?"
timeout 900 python bench.py > gpurun_out/r4g/bench_c1.json 2> gpurun_out/r4g/bench_c1.err; echo "c1 rc $?"
python -c "
import json
for f in ('bench_c4','bench_c5','bench_c1'):
    try:
        d=json.load(open('gpurun_out/r4g/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['config']['widths'], d['host_enqueue_ms_per_step']['mean']); print('  parity', {k:v for k,v in d.get('parity',{}).items() if k!='what'}); print('  cpu', d.get('cpu_baseline'))
    except Exception as e: print(f, 'failed', e)
"
tail -3 gpurun_out/r4g/bench_c4.err gpurun_out/r4g/bench_c5.err
