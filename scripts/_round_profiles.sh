# gpurun driver of a round's profile set (what profiles/rNN_README.txt lists): bash scripts/_round_profiles.sh r06f
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=${1:-r06f}
bash scripts/_prof.sh $T c1 > gpurun_out/${T}_prof.log 2>&1
bash scripts/_trace_config.sh c4 $T > gpurun_out/${T}_c4.log 2>&1
bash scripts/_trace_config.sh c5 $T > gpurun_out/${T}_c5.log 2>&1
python scripts/lstm_ab.py "" "T=576" > gpurun_out/$T/lstm_sweeps_alone.txt 2>&1
python scripts/gemm_bench.py > gpurun_out/$T/gemm_bench.txt 2>&1
python scripts/x6_bench.py > gpurun_out/$T/x6_bench.txt 2>&1
python scripts/follow_ab.py > gpurun_out/$T/follow_ab.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/mfma_rate scripts/mfma_rate.hip && /tmp/mfma_rate > gpurun_out/$T/mfma_rate.txt 2>&1
tail -3 gpurun_out/${T}_prof.log | cut -c1-600; tail -4 gpurun_out/$T/x6_bench.txt; cat gpurun_out/$T/mfma_rate.txt | head -4
# the opt-in fp16x3 split: its products alone, and the step under both splits on this box
X6_SCHEME=fp16x3 python scripts/x6_bench.py > gpurun_out/$T/x6_bench_fp16x3.txt 2>&1
for m in bf16x6 fp16x3 bf16x6 fp16x3; do VOCR_LSTM_GEMM=$m python bench.py --no-cpu-baseline --no-gemm-alone > gpurun_out/$T/ab_$m.json 2> /dev/null; python -c "
import json; d=json.loads(open('gpurun_out/$T/ab_$m.json').read().strip().splitlines()[-1]); e=d['ms_per_step_by_entry_point']; print('$m', d['value'], 'line-images/s', d['ms_per_step'], 'ms', {k:e[k] for k in e if 'x6' in k or 'h3' in k})"; done > gpurun_out/$T/split_ab.txt 2>&1
VOCR_LSTM_GEMM=fp16x3 python bench.py > gpurun_out/$T/bench_fp16x3.json 2> gpurun_out/$T/bench_fp16x3.err
cat gpurun_out/$T/split_ab.txt
