#!/usr/bin/env python3
"""Per-kernel sums of a rocprofv3 --pmc pass: pmc_kernel.py <dir> <kernel name substring>"""
import csv
import glob
import sys
from collections import defaultdict

acc, n = defaultdict(float), set()
for f in glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            n.add(r["Dispatch_Id"])
print(sys.argv[2], "launches", len(n))
for k, v in sorted(acc.items()):
    print(f"  {k:28s} {v:16.0f}  per launch {v / max(len(n), 1):14.0f}")
