cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "gemm or bilstm or persistent" 2>&1 | tail -3
python -m pytest tests/test_model_golden_gpu.py tests/test_round2_gpu.py -m gpu -x -q 2>&1 | tail -3
echo tailfill on; python scripts/gemm_bench.py 2>&1 | grep -v amdgpu
echo tailfill off; VOCR_GEMM_TAILFILL=0 python scripts/gemm_bench.py 2>&1 | grep -v amdgpu
run() { python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['h2d_inclusive']['ms_per_step'])"; }
run base
VOCR_GEMM_TAILFILL=0 run no_tailfill
VOCR_FWD_PREP=0 run no_prep
run base
