cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/fetch_calib scripts/fetch_calib.hip || exit 1
mkdir -p gpurun_out/calib
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/calib -- /tmp/fetch_calib > gpurun_out/calib/out.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/calib/*/*counter_collection.csv")[0]
for r in csv.DictReader(open(f)):
    print(r["Kernel_Name"][:60], r["Counter_Name"], r["Counter_Value"])
PY
