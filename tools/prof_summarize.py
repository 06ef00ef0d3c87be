"""Summarises rocprofv3 csv output (kernel stats + per-kernel counter averages) into a small text table."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
for f in stats:
    print("== kernel stats (%s)" % os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        print("  %-48s calls=%-5s avg_ns=%-12s total_ns=%-14s pct=%s" % (row.get("Name", "")[:48], row.get("Calls"), row.get("AverageNs"), row.get("TotalDurationNs"), row.get("Percentage")))
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:48]][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("== counters (average per dispatch)")
for k, cs in agg.items():
    print("  " + k)
    for c, v in sorted(cs.items()):
        print("      %-28s %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
