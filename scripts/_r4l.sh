cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4l
VOCR_CONV_WINO4=1 timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "conv3x3_fwd_dgrad_wgrad and transform" 2>&1 | tail -3
SWEEP=0 VOCR_CONV_WINO4=1 python scripts/conv_bench.py 2>&1 | grep -v amdgpu.ids
SWEEP=0 python scripts/conv_bench.py 2>&1 | grep -v amdgpu.ids | tail -1
VOCR_CONV_WINO4=1 timeout 300 python bench.py > gpurun_out/r4l/bench_w4.json 2> gpurun_out/r4l/bench_w4.err; echo rc $?
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r4l/bench.json 2>/dev/null
python -c "
import json
for f in ('bench_w4','bench'):
    d=json.load(open('gpurun_out/r4l/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d.get('parity',{}).get('label_mismatches'), d.get('parity',{}).get('loss_rel_err'))
"
VOCR_CONV_WINO4=1 timeout 900 python -m pytest tests/test_round2_gpu.py -q -k "full_size" 2>&1 | tail -3
