cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4h
python scripts/lstm_bench.py 2>&1 | grep -v amdgpu
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r4h/w_alone -- python3 scripts/lstm_bench.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r4h/w_step -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gemm-alone > /dev/null 2>&1
for d in w_alone w_step; do python - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/r4h/$d/*/*counter_collection.csv")[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "lstm_" in k: acc[k.split("(")[0][-40:]].append(float(r["Counter_Value"]))
for k, v in acc.items(): print("$d", k, "calls", len(v), "WRITE_SIZE per call KB %.0f" % (sum(v) / len(v)))
PY
done
rm -rf gpurun_out/r4h/w_alone gpurun_out/r4h/w_step
python -m pytest tests/test_ops_gpu.py -q -k "lstm or sweep" 2>&1 | tail -3
