cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02c
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02c/trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02c/bench_profiled.json 2> gpurun_out/r02c/err.txt
