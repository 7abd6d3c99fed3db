cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/tl
rm -rf $T; mkdir -p $T
export VOCR_PROB_DW_OVERLAP=0
rocprofv3 --kernel-trace --stats --output-format csv -d $T/trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gemm-alone > $T/bench_profiled.json 2> $T/err
python scripts/timeline.py $(find $T/trace -name "*kernel_trace.csv" | head -1) 3 > $T/timeline.txt 2>&1
grep -n "lstm_bwd_chain4w\|lstm_fwd_chain4w\|ctc_grad\|gemm" $T/timeline.txt | cut -c1-110 | sed -n 1,60p
