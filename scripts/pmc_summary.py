"""Summarise a rocprofv3 --pmc pass (counter_collection CSV): KB per dispatch of FETCH_SIZE / WRITE_SIZE per kernel,
and for conv3x3_kernel the forward-pass launches (first n_conv of every 13) separately.
usage: python scripts/pmc_summary.py <counter_collection.csv> [<counter_collection.csv> ...]"""
import collections, csv, sys
for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    if not rows:
        continue
    cname = rows[0]["Counter_Name"]
    per = collections.OrderedDict()
    conv = []
    for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        k = k.split("(")[0]
        per.setdefault(k, []).append(float(r["Counter_Value"]))
        if k.startswith("conv3x3_kernel<"):
            conv.append(float(r["Counter_Value"]))
    print(cname)
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:14]:
        print("  %-58s calls %5d  sum_KB %14.1f  per_call_KB %12.2f" % (k[:58], len(v), sum(v), sum(v) / len(v)))
    if conv:
        fw = [v for i, v in enumerate(conv) if i % 13 < 7]
        dg = [v for i, v in enumerate(conv) if i % 13 >= 7]
        print("  conv3x3_kernel, all variants: forward launches per_call_KB %.1f (n=%d), data-gradient launches %.1f (n=%d), all %.1f"
              % (sum(fw) / len(fw), len(fw), sum(dg) / max(1, len(dg)), len(dg), sum(conv) / len(conv)))
    print()
