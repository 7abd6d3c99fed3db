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
