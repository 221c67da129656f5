"""Per-kernel means of the rocprofv3 --pmc passes collected by tools/collect_pmc.sh -> JSON on stdout
(the format bench.py's pmc_profile() reads from profiles/rNN_pmc_counters_*.json)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
cfg = 2
for i, a in enumerate(sys.argv):
    if a == "--config":
        cfg = int(sys.argv[i + 1])
S, F = {2: (1024, 256), 3: (4096, 256), 4: (4096, 128), 5: (4096, 256)}[cfg]
acc = defaultdict(lambda: defaultdict(list))
disp = defaultdict(dict)
for path in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    per = defaultdict(float)        # (dispatch, kernel, counter) -> sum over the dimension rows
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0]
        if not name.startswith("k_"):
            continue
        per[(r["Dispatch_Id"], name, r["Counter_Name"])] += float(r["Counter_Value"])
        # what the dispatch packet says the kernel occupies (same for every dispatch of a kernel): registers, LDS, scratch
        for col, key in (("VGPR_Count", "vgpr_count"), ("Accum_VGPR_Count", "agpr_count"), ("SGPR_Count", "sgpr_count"), ("LDS_Block_Size", "lds_block_size"),
                         ("Scratch_Size", "scratch_size"), ("Workgroup_Size", "workgroup_size"), ("Grid_Size", "grid_size")):
            if col in r and r[col] not in ("", None):
                try:
                    disp[name][key] = int(float(r[col]))
                except ValueError:
                    pass
    for (d, name, c), v in per.items():
        acc[name][c].append(v)
# the build the counters belong to: bench.py quotes them only next to the same hx_build_id
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    from hmp3_amd import api
    build_id = api.build_id()
except Exception:
    build_id = None
kern = {}
for name, cs in acc.items():
    k = {c + ("_KB" if c in ("FETCH_SIZE", "WRITE_SIZE") else ""): sum(v) / len(v) for c, v in cs.items()}
    if "FETCH_SIZE_KB" in k and "WRITE_SIZE_KB" in k:
        # gfx950: FETCH_SIZE tallies the 128-byte requests of 16-byte-per-lane streaming loads at 64 bytes (guide, HBM section)
        k["hbm_bytes_corrected"] = (2.0 * k["FETCH_SIZE_KB"] + k["WRITE_SIZE_KB"]) * 1024.0
    if k.get("SQ_INSTS_VALU") and k.get("SQ_BUSY_CYCLES"):
        # vector issue at the SIMD: a wave64 vector instruction holds its SIMD's issue port 2 cycles (MI355X_MICROARCH.md; its own wave
        # 4, which is what SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES measures); SQ_BUSY_CYCLES is the sum over the chip's 32 shader engines
        k["valu_issue_share_at_simd"] = 2.0 * k["SQ_INSTS_VALU"] * 32.0 / (1024.0 * k["SQ_BUSY_CYCLES"])
    if disp.get(name):
        d = dict(disp[name])
        # (as rocprofv3 prints them: LDS rounded up to 512 bytes, VGPR_Count in its own allocation unit - half the
        # per-lane registers of a wave64 kernel; the code objects' exact figures are in DESIGN.md section 4)
        k["dispatch"] = d
    kern[name] = k
print(json.dumps({
    "command": "tools/collect_pmc.sh: rocprofv3 --pmc <group> --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-worst-case --host-fed 0 --verify 0"
               + (" --config %d" % cfg if cfg != 2 else "") + " (one pass per group: sq1, sq2, fetch, write)",
    "build_id": build_id,
    "workload": {"config": cfg, "streams": S, "frames_per_step": F},
    "units": "FETCH_SIZE / WRITE_SIZE in KB per launch as reported; SQ_* raw counts per launch (cycle counters in quad-cycles); means over the launches of the run",
    "note": "gfx950: hbm_bytes_corrected = 2 * FETCH_SIZE + WRITE_SIZE (16-byte-per-lane streaming loads are tallied at half their bytes; MI355X_MICROARCH.md, HBM section)",
    "kernels": kern}, indent=1))
