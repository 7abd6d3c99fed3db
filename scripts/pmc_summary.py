"""Summarise rocprofv3 --pmc passes (counter_collection CSVs).
  python scripts/pmc_summary.py traffic <FETCH csv> <WRITE csv>   -> per-kernel KB per dispatch; conv forward / data-gradient split
  python scripts/pmc_summary.py busy <csv> [<csv> ...]             -> MFMA-busy / wait shares for conv forward, wgrad, GEMM
The matching *_kernel_trace.csv (same directory) supplies the dispatch durations."""
import collections, csv, glob, os, sys


def short(k):
    return k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def load(path):
    rows = list(csv.DictReader(open(path)))
    kt = glob.glob(os.path.join(os.path.dirname(path), "*kernel_trace.csv"))
    dur = {}
    if kt:
        for r in csv.DictReader(open(kt[0])):
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return rows, dur


def traffic(paths):
    out = {}
    for path in paths:
        rows, _ = load(path)
        if not rows:
            continue
        cname = rows[0]["Counter_Name"]
        per = collections.OrderedDict()
        conv = []
        for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
            k = short(r["Kernel_Name"])
            per.setdefault(k, []).append(float(r["Counter_Value"]))
            if k.startswith("conv3x3_wino_kernel<") or k.startswith("conv3x3_dma_kernel<") or k.startswith("conv3x3_kernel<"):
                conv.append(float(r["Counter_Value"]))
        print(cname)
        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:14]:
            print("  %-58s calls %5d  sum_KB %14.1f  per_call_KB %12.2f" % (k[:58], len(v), sum(v), sum(v) / len(v)))
        if conv:
            fw = [v for i, v in enumerate(conv) if i % 13 < 7]
            dg = [v for i, v in enumerate(conv) if i % 13 >= 7]
            out[cname] = (sum(fw) / len(fw), sum(dg) / max(1, len(dg)), sum(conv) / len(conv))
            print("  conv forward/data-gradient kernel, all variants: forward launches per_call_KB %.1f (n=%d), data-gradient launches %.1f (n=%d), all %.1f"
                  % (out[cname][0], len(fw), out[cname][1], len(dg), out[cname][2]))
        print()
    return out


def busy(paths):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in paths:
        rows, dur = load(path)
        for r in rows:
            k = short(r["Kernel_Name"])
            key = "conv forward (%s)" % k.split("<")[0] if (k.startswith("conv3x3_dma_kernel") or k.startswith("conv3x3_wino_kernel")) \
                else "conv weight gradient (%s)" % k.split("<")[0] if (k.startswith("conv3x3_wgrad_kernel") or k.startswith("conv3x3_wgrad_wino")) \
                else "GEMM tile kernel (%s)" % k[:60] if k.startswith("gemm_f32_kernel") \
                else "GEMM panel kernel (%s)" % k[:60] if k.startswith("gemm_dma_kernel") else None
            if key is None:
                continue
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[key]["_dur_ns"].append(dur.get(r["Dispatch_Id"], 0))
    for key, d in acc.items():
        m = {c: sum(v[-3:]) / len(v[-3:]) for c, v in d.items()}          # last 3 launches (warm)
        print(key)
        for c in sorted(m):
            print("    %-30s %.4g" % (c, m[c]))
        if "GRBM_GUI_ACTIVE" in m and m.get("_dur_ns"):
            clk = m["GRBM_GUI_ACTIVE"] / 8.0 / m["_dur_ns"]
            print("    -> clock under the profiler %.2f GHz (GRBM_GUI_ACTIVE / 8 XCDs / duration)" % clk)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CYCLES" in m:
            # SQ_BUSY_CYCLES sums over the 8 XCDs' SQs... normalise MFMA-busy by (duration x clock x 1024 SIMDs) instead
            pass
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
            simd_cycles = m["GRBM_GUI_ACTIVE"] / 8.0 * 1024
            print("    -> MFMA pipe busy %.1f %% of SIMD-cycles (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs))"
                  % (100.0 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles))
        if "SQ_WAVE_CYCLES" in m:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                if c in m:
                    print("    -> %-18s %.1f %% of wave-cycles" % (c, 100.0 * m[c] / m["SQ_WAVE_CYCLES"]))
        print()


if __name__ == "__main__":
    if sys.argv[1] == "traffic":
        traffic(sys.argv[2:])
    else:
        busy(sys.argv[2:])
