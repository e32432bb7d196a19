#!/usr/bin/env python3
"""Condense rocprofv3 output dirs (stats + pmc passes) into a small text summary for profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
print(f"# rocprofv3 summary of {os.path.basename(out)}")
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    print(f"\n## kernel stats ({os.path.relpath(f, out)})")
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print({k: row[k] for k in row if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = defaultdict(lambda: defaultdict(float))
        calls = defaultdict(set)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "?")
                if "vrt_" not in k:
                    continue
                agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                calls[k].add(row.get("Dispatch_Id"))
        print(f"\n## {os.path.relpath(f, out)}")
        for k, cs in agg.items():
            n = max(1, len(calls[k]))
            print(f"kernel {k[:90]}  dispatches={n}")
            for c, v in sorted(cs.items()):
                print(f"   {c:32s} total={v:.6g}  per_dispatch={v / n:.6g}")
