#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_trace.csv: per kernel, the average duration over the LAST n launches (steady state)."""
import csv, sys, collections
path, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 100
d = collections.defaultdict(list)
with open(path) as f:
    for r in csv.DictReader(f):
        d[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("%-40s %8s %12s %12s %12s" % ("kernel", "calls", "avg_all_us", "avg_last_us", "max_last_us"))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1][-n:])):
    t = v[-n:]
    print("%-40s %8d %12.1f %12.1f %12.1f" % (k[:40], len(v), sum(v) / len(v) / 1e3, sum(t) / len(t) / 1e3, max(t) / 1e3))
