cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4k
timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "bn or fused or conv_bn or wgrad" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_model_golden_gpu.py tests/test_round2_gpu.py -q 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r4k/bench.json 2> gpurun_out/r4k/bench.err
VOCR_BN_FUSED_FINAL=0 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r4k/bench_off.json 2> /dev/null
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r4k/bench2.json 2> /dev/null
VOCR_BN_FUSED_FINAL=0 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r4k/bench_off2.json 2> /dev/null
python -c "
import json
for f in ('bench','bench_off','bench2','bench_off2'):
    d=json.load(open('gpurun_out/r4k/%s.json'%f)); print(f, d['value'], d['ms_per_step'], {k:v for k,v in d['ms_per_step_by_entry_point'].items() if 'bn' in k or 'wgrad' in k})
"
