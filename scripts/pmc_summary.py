"""Summarise rocprofv3 --pmc passes (counter_collection CSVs).
  python scripts/pmc_summary.py traffic <FETCH csv> <WRITE csv>   -> per-kernel KB per dispatch; conv forward / data-gradient split
  python scripts/pmc_summary.py busy <csv> [<csv> ...]             -> MFMA-busy / wait shares for conv forward, wgrad, GEMM
The matching *_kernel_trace.csv (same directory) supplies the dispatch durations."""
import collections, csv, glob, os, re, sys


def short(k):
    # names with _Float16 parameters stay mangled in rocprofv3's CSVs: _ZN12_GLOBAL__N_1<len><name>E...
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", k)
    if m:
        n = int(m.group(1))
        return k[m.end():m.end() + n]
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
            if k.startswith("conv3x3_wino_kernel<") or k.startswith("conv3x3_wino2_kernel") or k.startswith("conv3x3_wino4") or k.startswith("conv3x3_dma_kernel<") or k.startswith("conv3x3_kernel<"):
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
            key = "conv forward (%s)" % k.split("<")[0] if (k.startswith("conv3x3_dma_kernel") or k.startswith("conv3x3_wino_kernel") or k.startswith("conv3x3_wino2_kernel") or k.startswith("conv3x3_wino4")) \
                else "conv weight gradient (%s)" % k.split("<")[0] if (k.startswith("conv3x3_wgrad_kernel") or k.startswith("conv3x3_wgrad_wino")) \
                else "fp16-operand conv forward (%s)" % k.split("<")[0] if k.startswith("conv3x3_h16_kernel") \
                else "fp16-operand conv weight gradient (%s)" % k.split("<")[0] if k.startswith("conv3x3_wgrad_h16_kernel") \
                else "LSTM sweep (%s)" % k.split("<")[0] if (k.startswith("lstm_fwd_chain4w") or k.startswith("lstm_bwd_chain4w")) \
                else "bf16x6 GEMM (%s)" % k[:40] if k.startswith("gemm_x6_kernel") \
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


FAMILIES = (      # bench.py's FAMILY names -> kernel-name prefixes
    ("conv3x3 forward + data gradient (conv3x3_wino4 / conv3x3_wino2 kernels: F(4,3) / F(2,3) along the row)", ("conv3x3_wino_kernel", "conv3x3_wino2_kernel", "conv3x3_wino4", "conv3x3_wino8_kernel", "conv3x3_dma_kernel", "conv3x3_kernel", "conv3x3_tail")),
    ("conv3x3 weight gradient (conv3x3_wgrad_wino2d_kernel: F(3,2) along the row and across row pairs, piece stream)", ("conv3x3_wgrad_wino", "conv3x3_wgrad_kernel", "conv3x3_wgrad_smallcin")),
    ("dense GEMMs (gemm_dma_kernel / gemm_f32_kernel behind vocr_gemm and vocr_gemm_pair: bridge, LSTM projections, prob, their dX / dW)", ("gemm_dma_kernel", "gemm_f32_kernel")),
    ("LSTM sweeps (lstm_fwd_chain4w / lstm_bwd_chain4w)", ("lstm_fwd_", "lstm_bwd_")),
    ("conv3x3 with fp16 operands, fp32 accumulate (conv3x3_h16_kernel on NHWC fp16 activations: forward + data gradient; conv3x3_wgrad_h16_kernel on "
     "channel-major fp16 copies: weight gradient; the register-staged kernels where the channel counts do not fit; v_mfma_f32_32x32x16_f16)",
     ("conv3x3_h16_kernel", "conv3x3_wgrad_h16_kernel", "conv3x3_f16_kernel", "conv3x3_wgrad_f16_kernel")),
    ("dense fp32 GEMMs on the bf16 matrix pipe (gemm_x6_kernel behind vocr_gemm_x6: every fp32 operand split EXACTLY into three bf16 planes, six "
     "v_mfma_f32_32x32x16_bf16 per fp32 product, fp32 accumulate, error below the f32-MFMA kernels' against fp64 - the LSTM projections and their "
     "dX / dW; FLOPs counted as the fp32 products they replace, peak = the bf16 MFMA peak / 6 = 416.7 TFLOP/s)", ("gemm_x6_kernel",)),
)


def traffic_json(fetch_csv, write_csv, out_path):
    """profiles/kernel_traffic.json: HBM-side KB per launch of every MFMA kernel family (FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950
    correction, WRITE_SIZE as reported), averaged over the launches of the profiled steps."""
    import json
    res = {}
    for cname, path in (("fetch", fetch_csv), ("write", write_csv)):
        rows, _ = load(path)
        for fam, prefixes in FAMILIES:
            vals = [float(r["Counter_Value"]) for r in rows if short(r["Kernel_Name"]).startswith(prefixes)]
            if vals:
                e = res.setdefault(fam, {})
                e[cname + "_KB_per_launch_reported"] = round(sum(vals) / len(vals), 1)
                e["launches_profiled"] = len(vals)
    for fam, e in res.items():
        e["fetch_KB_per_launch"] = round(2.0 * e.get("fetch_KB_per_launch_reported", 0.0), 1)
        e["write_KB_per_launch"] = e.get("write_KB_per_launch_reported", 0.0)
    res["_note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py --steps 2 --warmup 1; fetch_KB_per_launch = 2 x FETCH_SIZE "
                    "(gfx950 reports 1/2 of streamed read bytes: MI355X_MICROARCH.md, calibrated here with scripts/fetch_calib.hip), WRITE_SIZE exact; "
                    "per launch, averaged over all launches of the family in the profiled steps")
    # tie the numbers to the kernels they were measured on: bench.py reports traffic: null + a traffic_stale note when the library it
    # runs was built from another csrc tree
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from vistaocr_amd import build as _build
    res["csrc_tree_hash"] = _build.tree_hash()
    json.dump(res, open(out_path, "w"), indent=1)
    for fam, e in res.items():
        if isinstance(e, dict):
            print("%-60s fetch %9.1f KB (2 x %9.1f)  write %9.1f KB  per launch, n=%d" % (fam[:60], e["fetch_KB_per_launch"], e.get("fetch_KB_per_launch_reported", 0),
                                                                                          e["write_KB_per_launch"], e["launches_profiled"]))


if __name__ == "__main__":
    if sys.argv[1] == "traffic":
        traffic(sys.argv[2:])
    elif sys.argv[1] == "traffic_json":
        traffic_json(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        busy(sys.argv[2:])
