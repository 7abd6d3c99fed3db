import csv, glob, sys, collections
d = sys.argv[1]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))}
acc = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    if "gemm_x6_kernel" in r["Kernel_Name"]:
        acc[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
rows = [(dur[k], v) for k, v in acc.items() if dur.get(k, 0) > 150000]
for du, v in rows[-4:]:
    clk = v["GRBM_GUI_ACTIVE"] / 8.0 / du
    print("   main launch %.0f us: clock %.2f GHz, MFMA pipe busy %.1f %% of SIMD cycles" % (du / 1e3, clk, 100.0 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8.0 * 1024)))
