#!/usr/bin/env python3
"""Text timeline of a window of a rocprofv3 --kernel-trace CSV: all kernels between the n-th launch of kernel A (substring)
and the next launch of kernel B (substring).   usage: stage_gantt.py <dir> <A> <B> [n]"""
import csv
import glob
import sys

rows = []
for f in glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
A, B = sys.argv[2], sys.argv[3]
n = int(sys.argv[4]) if len(sys.argv) > 4 else 0
starts = [i for i, r in enumerate(rows) if A in r["Kernel_Name"]]
i0 = starts[n]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cvhip::", "")
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:9.1f} {e:9.1f} {e - s:8.1f}  q{r.get('Queue_Id', '?'):>3s}  {name[:70]}")
    if B in name and s > 0:
        break
