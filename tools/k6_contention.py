"""Reads the rocprofv3 passes written by tools/k6_contention.sh (counter_collection + kernel_trace CSVs per pass) and prints,
per (K6 build, streams in the launch), the counters per launch and what they say about what co-resident streams share:
SIMD-level vector issue, wave-time shares, LDS pipe utilisation, LDS / instruction-fetch latency, effective clock.
  python tools/k6_contention.py <dir> > summary.json"""
import csv, glob, json, os, sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))      # (kernel, streams) -> counter -> [per dispatch]
durs = defaultdict(list)
for p in sorted(glob.glob(os.path.join(root, "*"))):
    if not os.path.isdir(p):
        continue
    per = defaultdict(float); meta = {}
    for path in glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            n = r["Kernel_Name"].split("(")[0]
            if not n.startswith("k_alloc"):
                continue
            key = (r["Dispatch_Id"], n, int(r["Grid_Size"]) // 128)
            per[key + (r["Counter_Name"],)] += float(r["Counter_Value"])
            if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                meta[key] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    # per pass: the dispatches of one (kernel, streams) in launch order; the first two are warm-up calls
    ids = defaultdict(set)
    for (d, n, s, c) in per:
        ids[(n, s)].add(int(d))
    keep = {ns: set(sorted(v)[2:] if len(v) > 2 else sorted(v)) for ns, v in ids.items()}
    for (d, n, s, c), v in per.items():
        if int(d) in keep[(n, s)]:
            acc[(n, s)][c].append(v)
    for (d, n, s), v in meta.items():
        if int(d) in keep[(n, s)]:
            durs[(n, s)].append(v)
    for path in ([] if meta else glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True)):
        for r in csv.DictReader(open(path)):
            n = r["Kernel_Name"].split("(")[0]
            if n.startswith("k_alloc"):
                g = int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)
                durs[(n, g // 128)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
out = {}
for (n, s), cs in sorted(acc.items()):
    # (the first two dispatches of a size are warm-up calls: drop them where more than two exist)
    k = {c: sum(v) / len(v) for c, v in cs.items()}
    dl = durs.get((n, s), [])
    ms = (sum(dl) / len(dl)) if dl else None
    k["kernel_ms_profiled"] = ms
    d = {}
    wc = k.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS",
                  "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC", "SQ_INST_CYCLES_SALU"):
            if c in k:
                d[c + "/WAVE_CYCLES"] = round(k[c] / wc, 4)
    if ms and k.get("GRBM_GUI_ACTIVE"):
        d["effective_clock_GHz"] = round(k["GRBM_GUI_ACTIVE"] / 8.0 / (ms * 1e-3) / 1e9, 3)
    clk = d.get("effective_clock_GHz", 2.4)
    if ms and "SQ_INSTS_VALU" in k:
        # SIMD-level vector issue: a wave64 vector instruction occupies its SIMD's issue for 2 cycles (MI355X_MICROARCH.md)
        simds = 1024.0
        d["valu_issue_share_of_all_SIMDs"] = round(k["SQ_INSTS_VALU"] * 2.0 / (simds * ms * 1e-3 * clk * 1e9), 4)
        used = min(simds, max(1.0, s / 256.0) * 256.0 * 1.0 if s < 1024 else simds)
        d["note_simds"] = "1024 SIMDs; a launch of S < 1024 streams leaves SIMDs empty (2 waves per stream)"
    if k.get("SQ_INSTS_LDS") and k.get("SQ_INST_LEVEL_LDS"):
        d["lds_mean_latency_units"] = round(k["SQ_INST_LEVEL_LDS"] / k["SQ_INSTS_LDS"], 2)
    if k.get("SQ_IFETCH") and k.get("SQ_IFETCH_LEVEL"):
        d["ifetch_mean_latency_units"] = round(k["SQ_IFETCH_LEVEL"] / k["SQ_IFETCH"], 2)
    if k.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_share"] = round(k.get("SQ_LDS_BANK_CONFLICT", 0) / k["SQ_LDS_IDX_ACTIVE"], 4)
        if ms:
            # LDS_IDX_ACTIVE summed over the CUs' LDS pipes, in cycles: share of the launch a CU's LDS is busy
            d["lds_pipe_busy_share_per_CU"] = round(k["SQ_LDS_IDX_ACTIVE"] / (256.0 * ms * 1e-3 * clk * 1e9), 4)
            d["lds_pipe_busy_share_per_CU_if_quad_cycles"] = round(4.0 * k["SQ_LDS_IDX_ACTIVE"] / (256.0 * ms * 1e-3 * clk * 1e9), 4)
    if k.get("SQ_WAVES") and wc:
        d["wave_lifetime_Mcycles"] = round(4.0 * wc / k["SQ_WAVES"] / 1e6, 3)
    out["%s S=%d (%.1f per CU)" % (n, s, s / 256.0)] = {"derived": d, "counters": {c: round(v, 1) if v is not None else None for c, v in sorted(k.items())}}
print(json.dumps(out, indent=1))
