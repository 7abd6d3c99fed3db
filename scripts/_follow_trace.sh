cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/ftrace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ftrace -- python3 scripts/follow_ab.py 294 32 1 > gpurun_out/ftrace/out.txt 2>&1
F=$(find gpurun_out/ftrace -name "*kernel_trace.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
sel = [r for r in rows if "lstm_fwd_chain4w" in r["Kernel_Name"] or "xproj_follow" in r["Kernel_Name"] or "nap" in r["Kernel_Name"]]
for r in sel[:14] + sel[-30:]:
    print("%-40s start %10.1f us  dur %8.1f us  q %s" % (r["Kernel_Name"][:40], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?")))
PY
