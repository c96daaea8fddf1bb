#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel, the dispatch with the largest
grid, all counters.  usage: pmc_summary.py <dir> [kernel-substring]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
by = defaultdict(dict)
meta = {}
for r in rows:
    name = r["Kernel_Name"].split("(")[0]
    if sub and sub not in name:
        continue
    key = (name, r["Dispatch_Id"])
    by[key][r["Counter_Name"]] = by[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    meta[key] = int(r["Grid_Size"])
best = {}
for (name, did), c in by.items():
    if name not in best or meta[(name, did)] > meta[(name, best[name])]:
        best[name] = did
for name, did in best.items():
    print(name, "dispatch", did, "grid", meta[(name, did)])
    for k, v in sorted(by[(name, did)].items()):
        print(f"   {k:28s} {v:16.0f}")
