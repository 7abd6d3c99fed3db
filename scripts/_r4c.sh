cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4c
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "conv" > gpurun_out/r4c/pytest_conv.txt 2>&1; tail -3 gpurun_out/r4c/pytest_conv.txt
SWEEP=0 VOCR_CONV_WINO2=0 python scripts/conv_bench.py 2>&1 | grep -v amdgpu.ids
SWEEP=0 python scripts/conv_bench.py 2>&1 | grep -v amdgpu.ids
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r4c/bench.json 2> gpurun_out/r4c/bench.err
VOCR_CONV_WINO2=0 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r4c/bench_old.json 2> /dev/null
python -c "
import json
for f in ('bench','bench_old'):
    d=json.load(open('gpurun_out/r4c/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d.get('parity'))
"
