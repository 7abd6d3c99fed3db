cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4a
timeout 600 python scripts/wgrad_ab.py 1 2 > gpurun_out/r4a/wgrad_ab.txt 2>&1
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "conv3x3" > gpurun_out/r4a/pytest_conv.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err
VOCR_WGRAD_WINO_DMA=1 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r4a/bench_old.json 2> /dev/null
cat gpurun_out/r4a/wgrad_ab.txt; tail -3 gpurun_out/r4a/pytest_conv.txt; python -c "
import json
for f in ('bench','bench_old'):
    d=json.load(open('gpurun_out/r4a/%s.json'%f)); print(f, d['value'], d['ms_per_step'])
"
