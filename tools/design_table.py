"""DESIGN.md section 4's table of current numbers, generated from profiles/ and the kernels' own resource notes:
   python tools/design_table.py r04 [--listings DIR]
Per kernel and BASELINE config (2 = 1024 x 256, 3 = 4096 x 256): mean duration in the bench run (rocprofv3 --kernel-trace
--stats; under the overlap, so a front-end kernel's duration includes its wait for the allocator's LDS), HBM bytes per launch
(2 x FETCH_SIZE + WRITE_SIZE), vector-issue share per wave (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES: a wave64 instruction holds its
wave 4 cycles) and at the SIMD (2 x SQ_INSTS_VALU / (1024 SIMDs x busy cycles): it holds the SIMD's issue port 2 cycles; busy cycles =
SQ_BUSY_CYCLES / 32, the counter being the sum over the chip's 32 shader engines), waiting share (SQ_WAIT_ANY / SQ_WAVE_CYCLES), LDS bank-conflict share (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE); VGPRs, LDS bytes and scratch bytes from the
code objects' metadata (listings compiled with the product's flags by tools/check_lds_flat.py --keep)."""
import csv
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r04"
lst = sys.argv[sys.argv.index("--listings") + 1] if "--listings" in sys.argv else None
if lst is None:
    lst = tempfile.mkdtemp(prefix="hxlst.")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "check_lds_flat.py"), "--keep", lst], stdout=subprocess.DEVNULL)
res = {}
for f in os.listdir(lst):
    if not f.endswith(".s"):
        continue
    txt = open(os.path.join(lst, f)).read()
    for m in re.finditer(r"\.group_segment_fixed_size: (\d+).*?\.name:\s+(\w+).*?\.private_segment_fixed_size: (\d+).*?\.vgpr_count:\s+(\d+)", txt, re.S):
        name = subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.split("(")[0].strip()
        res[name] = dict(lds=int(m.group(1)), scratch=int(m.group(3)), vgpr=int(m.group(4)))


def stats(cfg):
    p = os.path.join(ROOT, "profiles", "%s_kernel_stats_bench_config%d.csv" % (R, cfg))
    return {r["Name"].split("(")[0]: float(r["AverageNs"]) / 1e6 for r in csv.DictReader(open(p))} if os.path.exists(p) else {}


def pmc(cfg, shape):
    p = os.path.join(ROOT, "profiles", "%s_pmc_counters_config%d_%s.json" % (R, cfg, shape))
    return json.load(open(p))["kernels"] if os.path.exists(p) else {}


s2, s3, p2, p3 = stats(2), stats(3), pmc(2, "1024x256"), pmc(3, "4096x256")
order = ["k_polyphase", "k_spec", "k_spec_direct", "k_prep", "k_alloc", "k_alloc_slim", "k_pack", "k_msscan", "k_detect", "k_pack_carry", "k_pack_pre", "k_order"]
print("| kernel | VGPRs | LDS B | scratch B | ms cfg 2 | ms cfg 3 | GB cfg 2 | GB cfg 3 | vector issue per wave | at the SIMD | waiting | LDS conflicts |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
tot2 = tot3 = 0.0
for k in order:
    r = res.get(k, {})
    c = p2.get(k) or p3.get(k) or {}
    g2 = p2.get(k, {}).get("hbm_bytes_corrected")
    g3 = p3.get(k, {}).get("hbm_bytes_corrected")
    tot2 += g2 or 0
    tot3 += g3 or 0

    def pct(a, b):
        return "%.0f %%" % (100.0 * c[a] / c[b]) if c.get(b) else "-"
    simd = "%.0f %%" % (100.0 * 2.0 * c["SQ_INSTS_VALU"] * 32.0 / (1024.0 * c["SQ_BUSY_CYCLES"])) if c.get("SQ_BUSY_CYCLES") and c.get("SQ_INSTS_VALU") else "-"
    print("| `%s` | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s |" % (
        k, r.get("vgpr", "-"), r.get("lds", "-"), r.get("scratch", "-"),
        "%.3f" % s2[k] if k in s2 else "-", "%.3f" % s3[k] if k in s3 else "-",
        "%.2f" % (g2 / 1e9) if g2 else "-", "%.2f" % (g3 / 1e9) if g3 else "-",
        pct("SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES"), simd, pct("SQ_WAIT_ANY", "SQ_WAVE_CYCLES"), pct("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")))
print("| **step total** | | | | | | **%.2f** | **%.2f** | | | | |" % (tot2 / 1e9, tot3 / 1e9))


# ---- section 5: the bench lines of the four BASELINE configurations ----
print()
print("| BASELINE config (one GPU's share) | frames/s | × real-time | ms / step | K6 kernel, ms | step − K6 | roofline (HBM) | verified |")
print("|---|---|---|---|---|---|---|---|")
for cfg in (2, 3, 4, 5):
    p = os.path.join(ROOT, "profiles", "%s_bench_line_config%d.json" % (R, cfg))
    if not os.path.exists(p):
        continue
    d = json.load(open(p))
    r = d["roofline"]
    print("| %d: %s | **%.2f M** | %.0f k | %.2f | %s %.2f | %.2f | %.2f %% | %s / %s |" % (
        cfg, d["config"]["workload"].split(" per step")[0], d["value"] / 1e6, d["x_realtime_per_gpu"] / 1e3, d["ms_per_step"], r["kernel"], r["kernel_ms"],
        d["ms_per_step"] - r["kernel_ms"], 100.0 * r["frac"], d.get("verified_streams"), d.get("verify", {}).get("checked")))
    if cfg == 2:
        w, h, c = d.get("worst_case", {}), d.get("host_fed", {}), d.get("cpu_baseline", {})
        extra = ["| 2, correlation cycled over {0.7, 0, 1, 0.3} (`worst_case_value`) | %.2f M | | %.2f | %.2f | | | |" % (d.get("worst_case_value", 0) / 1e6, w.get("ms_per_step", 0), w.get("kernel_ms", 0)),
                 "| 2, fed from page-locked host memory (`host_fed`, PCIe both ways in the timed region) | %.2f M | | %.2f | | | H2D %.1f GB/s | |" % (h.get("value", 0) / 1e6, h.get("ms_per_step", 0), h.get("h2d_GBps", 0)),
                 "| the reference itself on the box's usable host CPUs (`cpu_baseline`, kind %s, %s cores) | %.3f M | | | | | | |" % (c.get("kind"), c.get("cores"), (c.get("value") or 0) / 1e6)]
        print("\n".join(extra))
        for o in d.get("other_configs", []):
            print("| %d, inside the config-2 line (`other_configs`, 4 timed steps) | %.2f M | | %.2f | %s %.2f | %.2f | %.2f %% | %s / %s |" % (
                o["baseline_config"], o["value"] / 1e6, o["ms_per_step"], o["kernel_build"], o["kernel_ms"], o["ms_per_step"] - o["kernel_ms"], 100.0 * (o["roofline_frac"] or 0), o["verified_streams"], o["verify_checked"]))
