"""Timeline of one bench run from a rocprofv3 --kernel-trace CSV: per dispatch, start / end in ms relative to the
first dispatch.  python tools/trace_timeline.py <kernel_trace.csv> [first_row] [rows]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if r["Kernel_Name"].startswith("k_")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
a = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 80
for r in rows[a:a + n]:
    name = r["Kernel_Name"].split("(")[0][:28]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    print("%-28s q%-3s start %9.3f  end %9.3f  dur %8.3f" % (name, r.get("Queue_Id", "?"), s, e, e - s))
