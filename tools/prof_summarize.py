"""Summarises rocprofv3 csv output (kernel stats + per-kernel counter averages) into a small text table."""
import csv
import glob
import os
import sys
from collections import defaultdict



def short(name):
    """Kernel name without its argument list (template arguments kept: they tell the sweep variants apart)."""
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name


out = sys.argv[1]
stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
for f in stats:
    print("== kernel stats (%s)" % os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        print("  %-64s calls=%-5s avg_ns=%-12s total_ns=%-14s pct=%s" % (short(row.get("Name", ""))[:64], row.get("Calls"), row.get("AverageNs"), row.get("TotalDurationNs"), row.get("Percentage")))
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        agg[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("== counters (average per dispatch)")
for k, cs in agg.items():
    print("  " + k)
    for c, v in sorted(cs.items()):
        print("      %-28s %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
