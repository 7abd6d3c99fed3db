cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4f
timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "bn or conv_bn" 2>&1 | tail -4
python scripts/bn_bench.py 2>&1 | grep -v amdgpu.ids | tail -12
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r4f/bench.json 2> gpurun_out/r4f/bench.err
python -c "
import json
for f in ('bench',):
    d=json.load(open('gpurun_out/r4f/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step',{}).get('mean'))
    print('   ', {k:v for k,v in d['ms_per_step_by_entry_point'].items() if v>0.25})
"
