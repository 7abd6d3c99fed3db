cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4e
timeout 1200 python -m pytest tests/test_configs_gpu.py -q -k "fp16 or batch_size_32 or speed_test or batch_40" 2>&1 | tail -12
python scripts/wgrad_ab.py 2 2>&1 | grep -v amdgpu.ids | tail -7
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r4e/bench.json 2> gpurun_out/r4e/bench.err
VOCR_CONV_OVERLAP=0 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r4e/bench_nooverlap.json 2> /dev/null
VOCR_CONV_OVERLAP=0 VOCR_LINEAR_DW_OVERLAP=0 VOCR_LSTM_DW_OVERLAP=0 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r4e/bench_noside.json 2> /dev/null
python -c "
import json
for f in ('bench','bench_nooverlap','bench_noside'):
    d=json.load(open('gpurun_out/r4e/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step'))
    print('   ', {k:v for k,v in d['ms_per_step_by_entry_point'].items() if v>0.3})
"
