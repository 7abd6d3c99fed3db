cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
hipcc -O3 --offload-arch=gfx950 -std=c++17 -Wno-pass-failed -o /tmp/conv_stamp scripts/conv_stamp.hip 2>/dev/null && for n in 1 11 32; do /tmp/conv_stamp $n; done
python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv" 2>&1 | tail -2
SWEEP=0 python scripts/conv_bench.py 2>&1 | grep -v amdgpu
python scripts/conv_occ.py 2>&1 | grep -v amdgpu | tail -8
