cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-r05a}
CFG=${2:-c1}      # bench.py --config of every pass (c1 headline; c4 / c5 side configurations)
mkdir -p gpurun_out/$T
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/clock_calib scripts/clock_calib.hip && /tmp/clock_calib > gpurun_out/$T/clock_calib.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/trace -- python3 bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-gemm-alone > gpurun_out/$T/bench_profiled.json 2> gpurun_out/$T/bench_profiled.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/$T/fetch -- python3 bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-gemm-alone > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/$T/write -- python3 bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-gemm-alone > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/$T/sq -- python3 scripts/one_conv.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/$T/sq2 -- python3 scripts/one_conv.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/$T/sq3 -- python3 scripts/one_conv.py > /dev/null 2>&1
F=$(find gpurun_out/$T/fetch -name "*counter_collection.csv" | head -1); W=$(find gpurun_out/$T/write -name "*counter_collection.csv" | head -1)
python scripts/pmc_summary.py traffic $F $W > gpurun_out/$T/pmc_hbm_traffic.txt 2>&1
python scripts/pmc_summary.py traffic_json $F $W gpurun_out/$T/kernel_traffic.json >> gpurun_out/$T/pmc_hbm_traffic.txt 2>&1
python scripts/pmc_summary.py busy $(find gpurun_out/$T/sq -name "*counter_collection.csv") $(find gpurun_out/$T/sq2 -name "*counter_collection.csv") $(find gpurun_out/$T/sq3 -name "*counter_collection.csv") > gpurun_out/$T/mfma_busy.txt 2>&1
python scripts/timeline.py $(find gpurun_out/$T/trace -name "*kernel_trace.csv" | head -1) 3 > gpurun_out/$T/timeline.txt 2>&1
cp $(find gpurun_out/$T/trace -name "*kernel_stats.csv" | head -1) gpurun_out/$T/kernel_stats.csv
VOCR_ROCTX=1 rocprofv3 --kernel-trace --marker-trace --output-format csv -d gpurun_out/$T/marker -- python3 bench.py --config $CFG --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python - > gpurun_out/$T/marker_ranges.txt 2>&1 <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/$T/marker/*/*marker_api_trace.csv")[0]
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    a = acc.setdefault(r["Function"], [0, 0.0])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("VOCR_ROCTX=1 rocprofv3 --kernel-trace --marker-trace -- python3 bench.py --steps 4 --warmup 2: roctx ranges (host-side spans: the host enqueues ahead of the device)")
for k, (n, us) in acc.items():
    print("  %-24s calls %4d   mean host span %9.1f us" % (k, n, us / n))
PY
# the final bench line reports the traffic measured by THIS run's passes (on the box only; copy gpurun_out/$T/kernel_traffic.json into profiles/ to keep it)
if [ "$CFG" = c1 ] && [ -s gpurun_out/$T/kernel_traffic.json ]; then cp gpurun_out/$T/kernel_traffic.json profiles/kernel_traffic.json; fi
python bench.py --config $CFG > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err; echo "bench rc $?"
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$T/smoke.txt 2>&1; tail -2 gpurun_out/$T/smoke.txt
/tmp/clock_calib >> gpurun_out/$T/clock_calib.txt 2>&1
tail -c 1500 gpurun_out/$T/bench.json; head -12 gpurun_out/$T/kernel_stats.csv; tail -8 gpurun_out/$T/pmc_hbm_traffic.txt
