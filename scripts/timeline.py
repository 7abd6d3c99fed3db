"""Print one training step of a rocprofv3 --kernel-trace CSV as a timeline (start us, duration us, kernel, grid)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    return n.replace('(anonymous namespace)::', '').replace('void ', '')[:64]
idx = [i for i, r in enumerate(rows) if 'clamp_adam' in r['Kernel_Name']]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
a, b = idx[-back - 1] + 1, idx[-back] + 1
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    s = int(r['Start_Timestamp']) - t0
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    print("%9.1f %8.1f  %-66s g=%s" % (s / 1e3, d / 1e3, short(r['Kernel_Name']), r.get('Grid_Size_X', '')))
print("step wall (kernel start to last kernel end): %.1f us" % ((max(int(r['End_Timestamp']) for r in rows[a:b]) - t0) / 1e3))
