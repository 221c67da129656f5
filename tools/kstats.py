"""Print the k_* rows of a rocprofv3 kernel_stats.csv: name, calls, average microseconds."""
import csv, sys
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        if r["Name"].startswith("k_"):
            print("  %-14s calls %3s  avg %9.1f us" % (r["Name"].split("(")[0], r["Calls"], float(r["AverageNs"]) / 1e3))
